// Greedy decoding: launched decision kernel, resident per-stream decoder (greedy_stream), cooperative experiment (greedy_flow), control helpers.
// Part of rnnt_kernels.hip.h (include that umbrella, not this file).
#pragma once

// ------------------------------------------------------------------------------------------------
// greedy_decide: one thread per stream applies the argmax of the previous evaluation (packed key written by the
// EPI_ARGMAX epilogue of joint.ffn_out) to the per-stream RNN-T greedy state machine of
// _decode_chunk_streaming_logic (online_rnnt_model.py:193-220):
//   blank      -> next frame, symbol counter reset
//   non-blank  -> emit, token <- k, the candidate LSTM state becomes the committed one (sel ^= 1: the two
//                 state buffers swap roles, no copy); after n_steps symbols on one frame move to the next frame.
// key == 0 means "no evaluation pending" (idle stream, or already applied).
// ------------------------------------------------------------------------------------------------
struct GreedyState {
    int* tok;        // [B] predictor input token
    int* fidx;       // [B] current frame index (relative to frame-buffer start)
    int* nsym;       // [B] symbols emitted on the current frame
    int* count;      // [B] tokens emitted so far
    int* tokens;     // [B][max_tokens]
    int* sel;        // [B] which LSTM state buffer is committed
    unsigned long long* key;   // [B]
    int* misc;       // [0] streams with frames left (greedy_decide with count != 0), [1] beam rows active, [2] decodable frames
    int* host_backlog;   // host-mapped pinned int: max over streams of (decodable frames - current frame), written every call
};

__global__ __launch_bounds__(64) void greedy_decide(int B, int blank, int n_steps, int max_tokens, int do_count, GreedyState st) {
    const int n_frames = st.misc[2];
    int act = 0, behind = 0;
    for (int b = threadIdx.x; b < B; b += 64) {
        const unsigned long long k64 = st.key[b];
        int f = st.fidx[b];
        if (k64 != 0ull) {
            st.key[b] = 0ull;
            const int k = (int)(0xFFFFFFFFu - (unsigned)(k64 & 0xFFFFFFFFull));
            if (k == blank) {
                f += 1;
                st.fidx[b] = f;
                st.nsym[b] = 0;
            } else {
                const int cnt = st.count[b];
                if (cnt < max_tokens) st.tokens[(long long)b * max_tokens + cnt] = k;
                st.count[b] = cnt + 1;
                st.tok[b] = k;
                st.sel[b] ^= 1;
                const int ns = st.nsym[b] + 1;
                if (ns >= n_steps) {
                    st.nsym[b] = 0;
                    f += 1;
                    st.fidx[b] = f;
                } else {
                    st.nsym[b] = ns;
                }
            }
        }
        act += f < n_frames ? 1 : 0;
        behind = max(behind, n_frames - f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) behind = max(behind, __shfl_xor(behind, o, 64));
    if (threadIdx.x == 0 && st.host_backlog) *st.host_backlog = behind;   // stale-tolerant feedback for the host's step budgets
    if (do_count) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) act += __shfl_xor(act, o, 64);
        if (threadIdx.x == 0) st.misc[0] = act;
    }
}

// ------------------------------------------------------------------------------------------------
// Resident greedy decoder (kernel greedy_stream below): the whole greedy decode of an utterance batch as ONE kernel.
// One workgroup owns one stream for the whole call and runs their RNN-T greedy state machine
// (_decode_chunk_streaming_logic, online_rnnt_model.py:193-220) without any exchange with other workgroups:
//   LSTM cell      gates = E[tok] + W_hh h      (predictor.py:200-204; gate rows interleaved i,f,g,o per unit)
//   joint          z = tanh(enc_proj[t] + W_c h' + b_c), W_c = W_pf W_pr folded (joint.py:54-66)
//   vocabulary     logits = W_out z + b_out, argmax (first index on ties, online_rnnt_model.py:212)
//   decision       blank -> next frame; else emit, commit (h', c'), <= n_steps symbols per frame.
// Streams are independent, so there is no lock step between workgroups: a "runaway" stream (n_steps symbols on many
// frames) only delays itself.  The lock-stepped launch-per-evaluation path needed 4 dependent kernels (~21 us, ~30 us
// when the encoder's grids fill the dispatcher) per evaluation of the SLOWEST stream; here an evaluation is ~1.7 MB of
// weight rows streamed from L2 by one CU plus ~0.4 MFLOP of VALU dot products.
// Matrix-vector layout: 16 lanes per weight row (16 x float4 = 256 contiguous bytes per load instruction and row,
// 4 loads cover K = 256), 16 rows per pass of the 256 threads, partial sums reduced with 4 in-row shuffles.
// Frames arrive while the kernel runs: the encoder stream publishes `frames_ready` after each chunk's joint.enc_ffn
// projection (kernel boundary = release); thread 0 polls it with relaxed agent-scope loads and, when it grows, issues
// ONE agent-scope acquire fence before anyone reads the new enc_proj rows.  Every wait is bounded (wall clock).
// ------------------------------------------------------------------------------------------------
struct DecP {
    const float* whh;     // [1024][256] gate-interleaved
    const float* egate;   // [vocab][1024] gate-interleaved input table
    const float* wjc;     // [256][256] folded pred_ffn o projection
    const float* bjc;     // [256]
    const float* wout;    // [vocab][256]
    const float* bout;    // [vocab]
    const float* encp;    // [B][fstride][256] projected encoder frames
    float* h;             // [2][bstride] state buffers (committed one selected by sel[b])
    float* c;
    int* sel;
    int* tok;
    int* fidx;
    int* nsym;
    int* count;
    int* tokens;          // [B][max_tokens]
    int* ctrl;            // [0] frames_ready (published by the encoder stream), [1] error flag, [2] evaluations (stats)
    long long fstride_f;  // floats between streams in encp
    long long bstride;    // floats between the two state buffers
    int B, vocab, blank, n_steps, max_tokens, n_total;
    long long timeout_ticks;   // s_memrealtime ticks (100 MHz)
    const int* nlim;           // optional per-stream frame count (offline search over padded batches); null = n_total for all
};

template <int SPW, int NTH, int U = 2, typename Epi>
__device__ __forceinline__ void dec_matvec(const float* __restrict__ W, int nrows, const float (*x)[RNNT_D], Epi epi) {
    const int tid = threadIdx.x, g = tid >> 4, l = tid & 15;
    float4 xv[SPW][4];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[s][j] = *reinterpret_cast<const float4*>(&x[s][4 * l + 64 * j]);
    // U weight rows in flight per lane group: U x NTH/16 KB per CU
    constexpr int RP = NTH / 16;    // rows per pass of the workgroup
    for (int r0 = 0; r0 < nrows; r0 += RP * U) {
        float4 w[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = min(r0 + RP * u + g, nrows - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[u][j] = ldg4(W + (long long)n * RNNT_D + 4 * l + 64 * j);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = r0 + RP * u + g;
            float acc[SPW];
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a = fmaf(w[u][j].x, xv[s][j].x, a);
                    a = fmaf(w[u][j].y, xv[s][j].y, a);
                    a = fmaf(w[u][j].z, xv[s][j].z, a);
                    a = fmaf(w[u][j].w, xv[s][j].w, a);
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o, 16);
                acc[s] = a;
            }
            if (l == 0 && n < nrows) epi(n, acc);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// greedy_stream<KF>: resident greedy decoder, one workgroup (512 threads) per stream, exploiting two facts of the
// reference's loop (online_rnnt_model.py:193-220) that make most of its evaluations redundant:
//   (1) a blank leaves (token, h, c) unchanged, so the predictor output -- and with it W_c h' + b_c, the predictor half
//       of the joint -- only changes when a symbol is emitted: the LSTM (1 MB of W_hh) and the folded projection (256 KB)
//       are recomputed only then ("dirty");
//   (2) while the predictor half is fixed, frames t, t+1, ... are independent of each other: KF frames go through the
//       vocabulary projection in ONE pass over W_out (412 KB), and the decisions are scanned in order -- blanks advance
//       the frame, the first non-blank emits, commits (h', c') and ends the scan (later frames' logits are discarded).
// The results are those of the sequential loop (same operands and summation order per logit).  The dependent chain is
// (#symbols) x (L + Jc + O) + (#blank runs / KF) x O instead of (#symbols + #frames) x (L + Jc + O).
// ------------------------------------------------------------------------------------------------
template <int KF, int UL = 2, int UO = 2>
__global__ __launch_bounds__(512) void greedy_stream(DecP p) {
    constexpr int NTH = 512;
    __shared__ __attribute__((aligned(16))) float hs[1][RNNT_D], cs[RNNT_D], h2[1][RNNT_D], c2[RNNT_D], pp[RNNT_D];
    __shared__ __attribute__((aligned(16))) float zs[KF][RNNT_D];
    __shared__ __attribute__((aligned(16))) float gates[4 * RNNT_D];
    __shared__ float redv[NTH / 16][KF];
    __shared__ int redi[NTH / 16][KF];
    __shared__ int s_ctl[4];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    if (b >= p.B) return;
    {
        const long long off = (long long)(ldgi(p.sel + b) & 1) * p.bstride + (long long)b * RNNT_D;
        if (tid < RNNT_D) { hs[0][tid] = ldg1(p.h + off + tid); cs[tid] = ldg1(p.c + off + tid); }
    }
    int tok = ldgi(p.tok + b), fidx = ldgi(p.fidx + b), nsym = ldgi(p.nsym + b), count = ldgi(p.count + b);   // uniform
    const int n_total = p.nlim ? min(p.n_total, ldgi(p.nlim + b)) : p.n_total;
    int evals = 0, seen_ready = 0;
    bool dirty = true;
    const float* encp = p.encp + (long long)b * p.fstride_f;
    __syncthreads();
    while (fidx < n_total) {
        // ---- frames available to this stream (bounded wait) ------------------------------------------------------------
        if (tid == 0) {
            int nf = __hip_atomic_load(p.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            int err = 0;
            while (nf <= fidx) {
                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks) {
                    __hip_atomic_store(p.ctrl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    err = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
                nf = __hip_atomic_load(p.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (nf > seen_ready) {   // ONE acquire per publication: nobody reads stale enc_proj lines
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                seen_ready = nf;
            }
            s_ctl[0] = err;
            s_ctl[1] = nf;
        }
        __syncthreads();
        if (s_ctl[0]) break;
        // frames evaluated together: right after a symbol only the current frame (more symbols are likely on it and the
        // single-frame pass is cheaper), otherwise up to KF
        const int kf = dirty ? 1 : min(KF, min(s_ctl[1], n_total) - fidx);
        if (dirty) {
            // ---- predictor step: gates = E[tok] + W_hh h; candidate (h', c'); pp = W_c h' + b_c ---------------------------
            dec_matvec<1, NTH, UL>(p.whh, 4 * RNNT_D, hs, [&](int n, const float* acc) {
                gates[n] = acc[0] + ldg1(p.egate + (long long)tok * (4 * RNNT_D) + n);
            });
            __syncthreads();
            if (tid < RNNT_D) {
                const float4 gt = *reinterpret_cast<const float4*>(&gates[4 * tid]);
                const float cc = sigmoidf_(gt.y) * cs[tid] + sigmoidf_(gt.x) * tanhf(gt.z);
                c2[tid] = cc;
                h2[0][tid] = sigmoidf_(gt.w) * tanhf(cc);
            }
            __syncthreads();
            dec_matvec<1, NTH, UL>(p.wjc, RNNT_D, h2, [&](int n, const float* acc) { pp[n] = acc[0] + ldg1(p.bjc + n); });
            dirty = false;
            __syncthreads();
        }
        // ---- joint activations of kf frames --------------------------------------------------------------------------------
        for (int e = tid; e < (kf == 1 ? 1 : KF) * RNNT_D; e += NTH) {
            const int k = e >> 8, n = e & 255;
            zs[k][n] = k < kf ? tanhf(pp[n] + ldg1(encp + (long long)(fidx + k) * RNNT_D + n)) : 0.f;
        }
        __syncthreads();
        // ---- vocabulary projection of the kf frames + per-frame argmax (first index on ties) ----------------------------------
        float bv[KF];
        int bi[KF];
#pragma unroll
        for (int k = 0; k < KF; ++k) { bv[k] = -INFINITY; bi[k] = 0x7fffffff; }
        if (KF > 1 && kf == 1) {
            dec_matvec<1, NTH, UL>(p.wout, p.vocab, zs, [&](int n, const float* acc) {
                const float v = acc[0] + ldg1(p.bout + n);
                if (v > bv[0]) { bv[0] = v; bi[0] = n; }
            });
        } else {
            dec_matvec<KF, NTH, UO>(p.wout, p.vocab, zs, [&](int n, const float* acc) {
                const float bo = ldg1(p.bout + n);
#pragma unroll
                for (int k = 0; k < KF; ++k) {
                    const float v = acc[k] + bo;
                    if (v > bv[k]) { bv[k] = v; bi[k] = n; }
                }
            });
        }
        if ((tid & 15) == 0) {
#pragma unroll
            for (int k = 0; k < KF; ++k) { redv[tid >> 4][k] = bv[k]; redi[tid >> 4][k] = bi[k]; }
        }
        __syncthreads();
        // ---- decisions, in frame order (every thread computes the same uniform result) -----------------------------------------
        {
            const int k = tid >> 6 < KF ? tid >> 6 : 0;     // wave k reduces frame k (waves >= KF idle)
            const int ln = tid & 63;
            float best = -INFINITY;
            int ix = 0x7fffffff;
            if (ln < NTH / 16) { best = redv[ln][k]; ix = redi[ln][k]; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(ix, o, 64);
                if (ov > best || (ov == best && oi < ix)) { best = ov; ix = oi; }
            }
            __syncthreads();                                  // redi fully read before it is reused for the winners
            if (ln == 0 && (tid >> 6) < KF) redi[0][tid >> 6] = ix;
        }
        __syncthreads();
        bool commit = false;
        for (int k = 0; k < kf; ++k) {
            const int w = redi[0][k];
            if (w == p.blank) { fidx += 1; nsym = 0; continue; }
            if (tid == 0 && count < p.max_tokens) p.tokens[(long long)b * p.max_tokens + count] = w;
            count += 1;
            tok = w;
            nsym += 1;
            if (nsym >= p.n_steps) { nsym = 0; fidx += 1; }
            commit = true;
            break;
        }
        if (commit) {
            if (tid < RNNT_D) { hs[0][tid] = h2[0][tid]; cs[tid] = c2[tid]; }
            dirty = true;
        }
        ++evals;
        __syncthreads();
    }
    // ---- write the state back (buffer 0 becomes the committed one) ----------------------------------------------------
    if (tid < RNNT_D) {
        stg1(p.h + (long long)b * RNNT_D + tid, hs[0][tid]);
        stg1(p.c + (long long)b * RNNT_D + tid, cs[tid]);
    }
    if (tid == 0) {
        p.sel[b] = 0; p.tok[b] = tok; p.fidx[b] = fidx; p.nsym[b] = nsym; p.count[b] = count;
        atomicAdd(p.ctrl + 2, evals);
    }
}

// packed argmax keys (EPI_ARGMAX) -> int32 indices (CTC head)
__global__ void unpack_keys(const unsigned long long* __restrict__ key, int* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = (int)(0xFFFFFFFFu - (unsigned)(key[i] & 0xFFFFFFFFull));
}

// ------------------------------------------------------------------------------------------------
// greedy_flow: cooperative, weights-STATIONARY greedy decoder for B <= 64 streams (experiment, RNNT_COOP=1).
// Workgroup g = sg * 16 + cg owns streams [16 sg, 16 sg + 16) and column group cg of every weight matrix, resident in
// LDS for the whole call: W_hh rows [64 cg, +64) (16 hidden units x 4 gates), W_c rows [16 cg, +16), W_out rows
// [26 cg, +26).  An evaluation is three exchanges among the 16 workgroups of a stream group
//     h' slices  ->  z slices  ->  per-workgroup argmax partials (+ frames_ready from cg 0)
// and every exchanged 32-bit value travels as ONE 8-byte word (payload | tag << 32, tag = evaluation number), written
// with a single write-through store and read with an L1-bypassing load: a word is valid iff its tag matches, so there
// is no counter, no store drain and no fence on the exchange path -- a consumer's cost is the round trips it needs to
// see all its words (the barrier-based predecessor paid ~10 us per exchange for drain + atomic + poll + load).  Buffers
// alternate by evaluation parity; a workgroup can only be one exchange ahead of the slowest of its group, so a slot is
// never rewritten before every reader has passed it.  Every workgroup derives the same decisions from the same words
// and keeps the stream state (token, frame, counts) privately; cell states live in registers of the lanes that own them.
// The predictor is re-evaluated only for streams that emitted (dirty), as in greedy_stream.  Spins are wall-clock bounded.
// ------------------------------------------------------------------------------------------------
struct FlowP {
    const float* whh; const float* egate; const float* wjc; const float* bjc; const float* wout; const float* bout;
    const float* encp;
    float* h; float* c;                 // [2][bstride] state buffers (committed one by sel[]; written back to buffer 0)
    int* sel; int* tok; int* fidx; int* nsym; int* count; int* tokens;
    unsigned long long* xh;             // [2][64][256] tagged h' words
    unsigned long long* xz;             // [2][64][256] tagged z words
    unsigned long long* xa;             // [2][4][16][16][4] tagged (ordered max, index, frames_ready, -) per (group, workgroup, stream)
    int* ctrl;                          // [0] frames_ready, [1] error, [2] evaluations, [4] abort
    long long fstride_f, bstride;
    int B, vocab, blank, n_steps, max_tokens, n_total;
    long long timeout_ticks;
    long long* dbg;                     // optional [16]: phase timers (100 MHz ticks) and poll iterations of workgroup 0
};

__device__ __forceinline__ float ld_sc1f(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int ld_sc1i(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1i(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_tag(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_tag(unsigned long long* p, unsigned payload, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define FLOW_G 64      // workgroups: 4 stream groups x 16 column groups
#define FLOW_CG 16
#define FLOW_LD 260
#define FLOW_NONE 0x7fffffff

// wait until all NW words of this thread carry `tag`; false = abort (timeout or another workgroup gave up)
template <int NW>
__device__ __forceinline__ bool flow_wait(const FlowP& p, const unsigned long long* src, unsigned tag, unsigned* out, int* s_flag, long long* polls) {
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while (true) {
        ++*polls;
        unsigned long long v[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) v[j] = ld_tag(src + j);
        int ok = 1;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            ok &= (unsigned)(v[j] >> 32) == tag ? 1 : 0;
            out[j] = (unsigned)v[j];
        }
        if (__syncthreads_and(ok)) return true;
        if (threadIdx.x == 0) {
            int bad = ld_sc1i(p.ctrl + 4) != 0 ? 1 : 0;
            if (!bad && (long long)__builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks) {
                st_sc1i(p.ctrl + 1, 2);
                st_sc1i(p.ctrl + 4, 1);
                bad = 1;
            }
            *s_flag = bad;
        }
        __syncthreads();
        if (*s_flag) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

__global__ __launch_bounds__(256) void greedy_flow(FlowP p) {
    __shared__ __attribute__((aligned(16))) float Wl[64 * FLOW_LD], Wj[16 * FLOW_LD], Wo[32 * FLOW_LD];
    __shared__ __attribute__((aligned(16))) float Hn[16 * FLOW_LD];     // h' of every stream (= committed h of the streams that emitted)
    __shared__ __attribute__((aligned(16))) float X[16 * FLOW_LD];      // z of every stream
    __shared__ __attribute__((aligned(16))) float red[4 * 256];
    __shared__ int s_tok[16], s_fidx[16], s_nsym[16], s_count[16], s_act[16], s_had[16], s_dirty[16], s_emit[16];
    __shared__ unsigned s_pv[16][16];
    __shared__ int s_pi[16][16];
    __shared__ unsigned s_bv[2][16];
    __shared__ int s_bi[2][16];
    __shared__ int s_flag, s_nf, s_done, s_any;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sg = blockIdx.x / FLOW_CG, cg = blockIdx.x % FLOW_CG;
    const int i = lane & 15, kq = lane >> 4;
    const int b0 = 16 * sg;
    const int nb = min(16, p.B - b0);
    if (nb <= 0) return;                                       // the whole stream group is absent
    // ---- resident weight slices ---------------------------------------------------------------------------------------
    for (int e = tid; e < 64 * 64; e += 256) {
        const int r = e >> 6, c4 = (e & 63) * 4;
        *reinterpret_cast<float4*>(&Wl[r * FLOW_LD + c4]) = ldg4(p.whh + (long long)(64 * cg + r) * RNNT_D + c4);
        if (r < 16) *reinterpret_cast<float4*>(&Wj[r * FLOW_LD + c4]) = ldg4(p.wjc + (long long)(16 * cg + r) * RNNT_D + c4);
        if (r < 32) *reinterpret_cast<float4*>(&Wo[r * FLOW_LD + c4]) = ldg4(p.wout + (long long)min(26 * cg + min(r, 25), p.vocab - 1) * RNNT_D + c4);
    }
    // ---- private copy of the streams' state --------------------------------------------------------------------------------
    {
        const int m = tid >> 4, c16 = (tid & 15) * 16;
        const int bb = b0 + min(m, nb - 1);
        const float* hp = p.h + (long long)(ldgi(p.sel + bb) & 1) * p.bstride + (long long)bb * RNNT_D + c16;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&Hn[m * FLOW_LD + c16 + 4 * j]) = ldg4(hp + 4 * j);
    }
    if (tid < 16) {
        const bool v = tid < nb;
        const int bb = b0 + tid;
        s_tok[tid] = v ? ldgi(p.tok + bb) : p.blank;
        s_fidx[tid] = v ? ldgi(p.fidx + bb) : p.n_total;
        s_nsym[tid] = v ? ldgi(p.nsym + bb) : 0;
        s_count[tid] = v ? ldgi(p.count + bb) : 0;
        s_act[tid] = 0; s_had[tid] = 0; s_dirty[tid] = 1; s_emit[tid] = 0;
    }
    // cell state of my (stream 4 kq + r, unit 16 cg + 4 wave + i / 4), held by the lanes with i % 4 == 0
    const int unit = 16 * cg + 4 * wave + (i >> 2);
    float cc[4], hc[4], cc2[4], hh2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int bb = b0 + min(4 * kq + r, nb - 1);
        const long long off = (long long)(ldgi(p.sel + bb) & 1) * p.bstride + (long long)bb * RNNT_D + unit;
        cc[r] = ldg1(p.c + off);
        hc[r] = ldg1(p.h + off);
        cc2[r] = cc[r];
        hh2[r] = hc[r];
    }
    unsigned e = 1;                                            // evaluation number = tag
    int seen_nf = 0, evals = 0;
    long long polls[3] = {0, 0, 0}, tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = (long long)__builtin_amdgcn_s_memrealtime();
#define FLOW_T(k) { if (p.dbg && blockIdx.x == 0 && tid == 0) { const long long t_ = (long long)__builtin_amdgcn_s_memrealtime(); tacc[k] += t_ - tl; tl = t_; } }
    // round 0: no argmax yet, only frames_ready from cg 0
    if (tid < 16) {
        unsigned long long* q = p.xa + ((((size_t)(e & 1) * 4 + sg) * 16 + cg) * 16 + tid) * 4;
        st_tag(q + 0, 0u, e);
        st_tag(q + 1, (unsigned)FLOW_NONE, e);
        st_tag(q + 2, cg == 0 ? (unsigned)ld_sc1i(p.ctrl) : 0u, e);
    }
    __syncthreads();
    while (true) {
        const unsigned par = e & 1;
        FLOW_T(5)
        // ---- D: gather the 16 partials of every stream, decide -----------------------------------------------------------
        {
            unsigned w3[3];
            const int wg = tid >> 4, m = tid & 15;
            if (!flow_wait<3>(p, p.xa + ((((size_t)par * 4 + sg) * 16 + wg) * 16 + m) * 4, e, w3, &s_flag, &polls[0])) return;
            FLOW_T(0)
            s_pv[wg][m] = w3[0];
            s_pi[wg][m] = (int)w3[1];
            if (tid == 0) s_nf = (int)w3[2];
        }
        __syncthreads();
        if (tid < 16) {
            const int m = tid;
            unsigned bv = 0u;
            int bi = FLOW_NONE;
            for (int g = 0; g < 16; ++g) {
                const unsigned v = s_pv[g][m];
                const int ix = s_pi[g][m];
                if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
            }
            int emit = 0;
            if (s_had[m] && bi != FLOW_NONE) {
                if (bi == p.blank) { s_fidx[m] += 1; s_nsym[m] = 0; }
                else {
                    const int cnt = s_count[m];
                    if (cg == 0 && cnt < p.max_tokens) p.tokens[(long long)(b0 + m) * p.max_tokens + cnt] = bi;
                    s_count[m] = cnt + 1;
                    s_tok[m] = bi;
                    const int ns = s_nsym[m] + 1;
                    if (ns >= p.n_steps) { s_nsym[m] = 0; s_fidx[m] += 1; } else { s_nsym[m] = ns; }
                    s_dirty[m] = 1;
                    emit = 1;
                }
            }
            s_emit[m] = emit;
            const int f = s_fidx[m];
            const int act = (m < nb && f < p.n_total && f < s_nf) ? 1 : 0;
            s_act[m] = act;
            const unsigned long long m16 = 0xFFFFull;
            const unsigned long long anyact = __ballot(act != 0) & m16, notdone = __ballot(m < nb && f < p.n_total) & m16;
            if (m == 0) { s_done = notdone == 0ull ? 1 : 0; s_any = anyact != 0ull ? 1 : 0; }
        }
        __syncthreads();
        // commit the cell / hidden state of the streams that emitted (their Hn row already is the new h)
        if ((i & 3) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (s_emit[4 * kq + r]) { cc[r] = cc2[r]; hc[r] = hh2[r]; }
        }
        if (s_done) break;
        if (s_nf > seen_nf) {   // new encoder frames were published: one agent-scope acquire before reading enc_proj rows
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            seen_nf = s_nf;
            __syncthreads();
        }
        const bool anyact = s_any != 0;
        if (anyact) {
            // ---- L: gates of my 16 units for the 16 streams; new candidate (h', c') only where the predictor input changed ----
            {
                f32x4_ acc = (f32x4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                for (int u = 0; u < 16; ++u) {
                    const float4 a = *reinterpret_cast<const float4*>(&Hn[i * FLOW_LD + 16 * u + 4 * kq]);
                    const float4 w = *reinterpret_cast<const float4*>(&Wl[(16 * wave + i) * FLOW_LD + 16 * u + 4 * kq]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
                }
                const int n = 64 * cg + 16 * wave + i;             // gate column (interleaved i,f,g,o)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * kq + r;
                    const float v = acc[r] + ldg1(p.egate + (long long)s_tok[m] * (4 * RNNT_D) + n);
                    const float gf = __shfl_down(v, 1, 64), gg = __shfl_down(v, 2, 64), go = __shfl_down(v, 3, 64);
                    if ((i & 3) == 0) {
                        if (s_dirty[m] && m < nb) {
                            const float c2v = sigmoidf_(gf) * cc[r] + sigmoidf_(v) * tanhf(gg);
                            cc2[r] = c2v;
                            hh2[r] = sigmoidf_(go) * tanhf(c2v);
                        }
                        st_tag(p.xh + ((size_t)par * 64 + b0 + m) * RNNT_D + unit, __float_as_uint(hh2[r]), e);
                    }
                }
            }
            __syncthreads();                                       // everybody has read Hn and s_dirty
            if (tid < 16) s_dirty[tid] = 0;
            // ---- J: all h' of my streams -> z = tanh(enc_proj[t] + h' W_c^T + b_c), my 16 columns ------------------------------
            {
                unsigned w16[16];
                const int m = tid >> 4, c16 = (tid & 15) * 16;
                FLOW_T(1)
                if (!flow_wait<16>(p, p.xh + ((size_t)par * 64 + b0 + m) * RNNT_D + c16, e, w16, &s_flag, &polls[1])) return;
                FLOW_T(2)
#pragma unroll
                for (int j = 0; j < 16; ++j) Hn[m * FLOW_LD + c16 + j] = __uint_as_float(w16[j]);
            }
            __syncthreads();
            {
                f32x4_ acc = (f32x4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 4 * wave; u < 4 * wave + 4; ++u) {
                    const float4 a = *reinterpret_cast<const float4*>(&Hn[i * FLOW_LD + 16 * u + 4 * kq]);
                    const float4 w = *reinterpret_cast<const float4*>(&Wj[i * FLOW_LD + 16 * u + 4 * kq]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * 256 + r * 64 + lane] = acc[r];
            }
            __syncthreads();
            {   // 256 outputs (16 streams x 16 columns), one per thread
                const int r = tid >> 6, ln = tid & 63;
                const float sum = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
                const int m = 4 * (ln >> 4) + r, n = 16 * cg + (ln & 15);
                float ev = 0.f;
                if (s_act[m]) ev = ldg1(p.encp + (long long)(b0 + m) * p.fstride_f + (long long)s_fidx[m] * RNNT_D + n);
                st_tag(p.xz + ((size_t)par * 64 + b0 + m) * RNNT_D + n, __float_as_uint(tanhf(sum + ldg1(p.bjc + n) + ev)), e);
            }
            // ---- O: all z of my streams -> logits of my 26 vocabulary rows -> argmax partial ----------------------------------------
            {
                unsigned w16[16];
                const int m = tid >> 4, c16 = (tid & 15) * 16;
                FLOW_T(3)
                if (!flow_wait<16>(p, p.xz + ((size_t)par * 64 + b0 + m) * RNNT_D + c16, e, w16, &s_flag, &polls[2])) return;
                FLOW_T(4)
#pragma unroll
                for (int j = 0; j < 16; ++j) X[m * FLOW_LD + c16 + j] = __uint_as_float(w16[j]);
            }
            __syncthreads();
            {
                const int tile = wave >> 1, kh = wave & 1;
                f32x4_ acc = (f32x4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 8 * kh; u < 8 * kh + 8; ++u) {
                    const float4 a = *reinterpret_cast<const float4*>(&X[i * FLOW_LD + 16 * u + 4 * kq]);
                    const float4 w = *reinterpret_cast<const float4*>(&Wo[(16 * tile + i) * FLOW_LD + 16 * u + 4 * kq]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * 256 + r * 64 + lane] = acc[r];
            }
            __syncthreads();
            if (tid < 128) {   // 2 tiles x (4 regs x 64 lanes): thread = (tile, lane), loops the 4 regs
                const int t2 = tid >> 6, ln = tid & 63;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sum = red[(2 * t2) * 256 + r * 64 + ln] + red[(2 * t2 + 1) * 256 + r * 64 + ln];
                    const int m = 4 * (ln >> 4) + r;
                    const int jr = 16 * t2 + (ln & 15);            // local vocabulary row 0..31 (26 valid)
                    const int n = 26 * cg + jr;
                    const bool nin = jr < 26 && n < p.vocab;
                    float v = nin ? sum + ldg1(p.bout + min(n, p.vocab - 1)) : -INFINITY;
                    int bi = nin ? n : FLOW_NONE;
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) {
                        const float ov = __shfl_xor(v, o, 16);
                        const int oi = __shfl_xor(bi, o, 16);
                        if (ov > v || (ov == v && oi < bi)) { v = ov; bi = oi; }
                    }
                    if ((ln & 15) == 0) {
                        unsigned uu = 0u;
                        if (bi != FLOW_NONE) {
                            uu = __float_as_uint(v);
                            uu = (uu & 0x80000000u) ? ~uu : (uu | 0x80000000u);   // order-preserving; > 0 for every real value
                        }
                        s_bv[t2][m] = uu;
                        s_bi[t2][m] = bi;
                    }
                }
            }
            __syncthreads();
            ++evals;
        } else {
            __builtin_amdgcn_s_sleep(64);                          // nothing decodable: wait for the encoder
            if (tid < 16) { s_bv[0][tid] = 0u; s_bv[1][tid] = 0u; s_bi[0][tid] = FLOW_NONE; s_bi[1][tid] = FLOW_NONE; }
            __syncthreads();
        }
        // ---- partial argmax of my rows + frames_ready (cg 0) for the next evaluation -------------------------------------------
        if (tid < 16) {
            const int m = tid;
            unsigned bv = s_bv[0][m];
            int bi = s_bi[0][m];
            if (s_bv[1][m] > bv || (s_bv[1][m] == bv && s_bi[1][m] < bi)) { bv = s_bv[1][m]; bi = s_bi[1][m]; }
            if (!s_act[m]) { bv = 0u; bi = FLOW_NONE; }
            s_had[m] = s_act[m];
            unsigned long long* q = p.xa + ((((size_t)((e + 1) & 1) * 4 + sg) * 16 + cg) * 16 + m) * 4;
            st_tag(q + 0, bv, e + 1);
            st_tag(q + 1, (unsigned)bi, e + 1);
            st_tag(q + 2, cg == 0 ? (unsigned)ld_sc1i(p.ctrl) : 0u, e + 1);
        }
        __syncthreads();
        ++e;
    }
    // ---- canonical state for the host / the next call (buffer 0 becomes the committed one) ---------------------------------
    if ((i & 3) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * kq + r;
            if (m < nb) {
                stg1(p.h + (long long)(b0 + m) * RNNT_D + unit, hc[r]);
                stg1(p.c + (long long)(b0 + m) * RNNT_D + unit, cc[r]);
            }
        }
    }
    if (cg == 0 && tid < nb) {
        const int b = b0 + tid;
        p.tok[b] = s_tok[tid]; p.fidx[b] = s_fidx[tid]; p.nsym[b] = s_nsym[tid]; p.sel[b] = 0; p.count[b] = s_count[tid];
    }
    if (cg == 0 && tid == 0) atomicAdd(p.ctrl + 2, evals);
    if (p.dbg && blockIdx.x == 0 && tid == 0) {
        for (int k = 0; k < 8; ++k) p.dbg[k] = tacc[k];
        for (int k = 0; k < 3; ++k) p.dbg[8 + k] = polls[k];
        p.dbg[11] = evals;
    }
#undef FLOW_T
}

// frames_ready <- n (one thread; the kernel boundary before it released the encoder's writes)
__global__ void publish_frames(int* ctrl, int n) {
    __hip_atomic_store(ctrl, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Do kernels of two streams really run at the same time?  ctrl[1] <- 1 if ctrl[0] becomes non-zero within `ticks`
// (100 MHz) while this kernel is resident.  A profiler that serialises dispatches, or two streams folded onto one
// hardware queue, make it time out; the resident decoder is then not used.
__global__ void probe_overlap_wait(int* ctrl, long long ticks) {
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    int seen = 0;
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (__hip_atomic_load(ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { seen = 1; break; }
        __builtin_amdgcn_s_sleep(32);
    }
    ctrl[1] = seen;
}

// ------------------------------------------------------------------------------------------------
// greedy_multi<KF>: the resident greedy decoder with FOUR workgroups (CUs) per stream and the weights STATIONARY on chip.
// greedy_stream streams 1.7 MB of weight rows through one CU per symbol (~23 us at the per-CU L2 fetch rate, ~2x that next to
// the encoder), and the utterance batch ends with its slowest stream's dependent chain.  Here part p of stream b keeps for
// the whole call
//   W_hh rows of hidden units [64 p, 64 p + 64)  (256 gate rows x 256)      in registers (128 per thread)
//   W_c  columns              [64 p, 64 p + 64)  (256 rows x 64, the folded pred_ffn o projection)  in registers (32 per thread)
//   W_out rows                [ceil(V/4) p, ...) (103 x 256 for V = 412)                             in LDS
// and a symbol costs two exchanges among the four parts instead of a weight stream:
//   X1  h' slice (64) + the part's partial sums of W_c h' over its 64 columns (256)  ->  every part has h' and
//       pp = b_c + sum of the four partials (added in part order, so all four hold the same bits)
//   XA  per frame of the pass: the part's (max logit, argmax) over its vocabulary rows  ->  every part takes the same decision
// (a run of blank frames costs one XA only: as in greedy_stream the predictor is re-evaluated only after a symbol and up to KF
// frames share a pass).  Every exchanged 32-bit value travels as one 8-byte word (payload | tag << 32) written with one
// write-through store and polled with L1-bypassing loads: valid iff the tag matches the exchange's sequence number; buffers
// alternate by that number's parity (a part can only be one exchange ahead of the slowest of its group).  Same state machine and
// per-logit operand order as the reference loop (online_rnnt_model.py:193-220); h' and pp are summed in a different (fixed)
// order than greedy_stream's, i.e. equal to float32 rounding.  Needs 4 * B workgroups resident at once (B <= 64 on 256 CUs);
// every wait is wall-clock bounded and an abort word releases the whole grid.
// ------------------------------------------------------------------------------------------------
#define GM_PARTS 4
#define GM_X1 320            // words per part: 64 h' + 256 pp partials
#define GM_WLD 260           // LDS row stride of the W_out slice (floats)
struct DecMP {
    const float* whh; const float* egate; const float* wjc; const float* bjc; const float* wout; const float* bout; const float* encp;
    float* h; float* c;
    int* sel; int* tok; int* fidx; int* nsym; int* count; int* tokens; int* ctrl;
    unsigned long long* x1;      // [2][B][4][GM_X1]
    unsigned long long* xa;      // [2][B][4][2 * KF]
    long long fstride_f, bstride;
    int B, vocab, blank, n_steps, max_tokens, n_total;
    long long timeout_ticks;
    const int* nlim;
    int rows_per;                // ceil(vocab / 4): vocabulary rows of a part (<= 128), resident in LDS
    long long* dbg;              // optional [16]: phase timers of workgroup (stream 0, part 0), 100 MHz ticks (RNNT_GM_DBG=1)
};

// spin until the word carries `tag`; false on abort / timeout (sets the error and abort words)
__device__ __forceinline__ bool gm_poll(const DecMP& p, const unsigned long long* src, unsigned tag, unsigned& out) {
    unsigned long long v = ld_tag(src);
    if ((unsigned)(v >> 32) == tag) { out = (unsigned)v; return true; }
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    int it = 0;
    while (true) {
        v = ld_tag(src);
        if ((unsigned)(v >> 32) == tag) { out = (unsigned)v; return true; }
        if ((++it & 63) == 0) {
            if (ld_sc1i(p.ctrl + 4) != 0) return false;
            if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks) {
                st_sc1i(p.ctrl + 1, 2);
                st_sc1i(p.ctrl + 4, 1);
                return false;
            }
        }
    }
}

// 512 threads.  Where the 423 KB of a part's weights live: the W_out rows in LDS (107 KB for V = 412), the W_hh slice in
// registers (128 per thread: row tid / 2, half of K) and the W_c columns in registers too (32 per thread): nothing is streamed
// after the launch.  (GM_WHH_REGS=0 streams the W_hh slice from L2 per symbol instead, 4 us.)  (Keeping W_hh in registers was tried: as
// plain arrays or parked in AGPRs through inline asm, hipcc spills the 128 weights of a thread to scratch at load time, which is
// the same L2 stream with worse coalescing.)  So a symbol costs a quarter of greedy_stream's weight stream plus two exchanges,
// and a run of blank frames one exchange and 15 KB.
#ifndef GM_WHH_REGS
#define GM_WHH_REGS 1
#endif
template <int KF>
__global__ __launch_bounds__(512) void greedy_multi(DecMP p) {
    extern __shared__ __attribute__((aligned(16))) float gm_smem[];
    float* Wo = gm_smem;                                  // [rows_per][GM_WLD]  this part's vocabulary rows of W_out
    float* hs = Wo + p.rows_per * GM_WLD;                 // [256] committed h
    float* h2 = hs + RNNT_D;                              // [256] candidate h'
    float* pp = h2 + RNNT_D;                              // [256] W_c h' + b_c
    float* mypp = pp + RNNT_D;                            // [256] this part's partial
    float* zs = mypp + RNNT_D;                            // [KF][256]
    float* redv = zs + KF * RNNT_D;                       // [8 waves][KF]
    int* redi = reinterpret_cast<int*>(redv + 8 * KF);    // [8][KF]
    unsigned* xav = reinterpret_cast<unsigned*>(redi + 8 * KF);   // [4 parts][2 KF] gathered partials
    int* s_bad = reinterpret_cast<int*>(xav + GM_PARTS * 2 * KF);
    float* lpart = reinterpret_cast<float*>(s_bad + 4);      // [KF][4 quarters][128 rows] partial logits
    static_assert(KF == 4, "thread (frame, row) mapping of the logit finish assumes KF * 128 == 512");
    const int tid0 = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot >> 2) * 8 + xcd, pw = slot & 3;   // the four parts of a stream share blockIdx % 8 (one XCD: speed hint only)
    if (b >= p.B) return;
    // ---- resident weights ------------------------------------------------------------------------------------------------
    const int cell_unit = 64 * pw + (tid0 >> 3);          // threads 0, 8, ... own the 64 hidden units of this part
    const bool cell = (tid0 & 7) == 0;
    const int v0 = p.rows_per * pw;                        // first vocabulary row of this part
    float cc = 0.f, cc2 = 0.f;
    float* gates = zs;                                     // [256] gate pre-activations of this part (zs is free during the predictor step)
#if GM_WHH_REGS
    float whh_r[128];                                      // W_hh[256 pw + tid / 2][128 (tid & 1) .. +128)
    {
        const float* src = p.whh + (long long)(256 * pw + (tid0 >> 1)) * RNNT_D + 128 * (tid0 & 1);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float4 v = ldg4(src + 4 * j);
            whh_r[4 * j] = v.x; whh_r[4 * j + 1] = v.y; whh_r[4 * j + 2] = v.z; whh_r[4 * j + 3] = v.w;
        }
    }
#endif
    // this part's columns of W_c: thread (row tid / 2, half of the 64 columns) holds 32 weights
    float4 wcv[8];
    {
        const float* wc = p.wjc + (long long)(tid0 >> 1) * RNNT_D + 64 * pw + 32 * (tid0 & 1);
#pragma unroll
        for (int u = 0; u < 8; ++u) wcv[u] = ldg4(wc + 4 * u);
    }
    {
    const int tid = tid0;
    for (int e = tid; e < p.rows_per * 64; e += 512) {
        const int r = e >> 6, c4 = (e & 63) * 4;
        const int vr = v0 + r;
        *reinterpret_cast<float4*>(&Wo[r * GM_WLD + c4]) = vr < p.vocab ? ldg4(p.wout + (long long)vr * RNNT_D + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---- state ---------------------------------------------------------------------------------------------------------------
    const long long soff = (long long)(ldgi(p.sel + b) & 1) * p.bstride + (long long)b * RNNT_D;
    if (tid < RNNT_D) hs[tid] = ldg1(p.h + soff + tid);
    cc = cell ? ldg1(p.c + soff + cell_unit) : 0.f;
    cc2 = cc;
    if (tid == 0) *s_bad = 0;
    }
    int tok = ldgi(p.tok + b), fidx = ldgi(p.fidx + b), nsym = ldgi(p.nsym + b), count = ldgi(p.count + b);
    const int n_total = p.nlim ? min(p.n_total, ldgi(p.nlim + b)) : p.n_total;
    const float* encp = p.encp + (long long)b * p.fstride_f;
    unsigned long long* x1b = p.x1 + (long long)b * GM_PARTS * GM_X1;
    unsigned long long* xab = p.xa + (long long)b * GM_PARTS * 2 * KF;
    const long long x1par = (long long)p.B * GM_PARTS * GM_X1, xapar = (long long)p.B * GM_PARTS * 2 * KF;
    unsigned ev1 = 0, ev3 = 0;
    int evals = 0, seen_ready = 0;
    bool dirty = true, bad = false;
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = (long long)__builtin_amdgcn_s_memrealtime();
    const bool tdbg = p.dbg != nullptr && b == 0 && pw == 0 && tid0 == 0;
#define GM_T(k) { if (tdbg) { const long long t_ = (long long)__builtin_amdgcn_s_memrealtime(); tacc[k] += t_ - tl; tl = t_; } }
    int nsymev = 0;
    __syncthreads();
    while (fidx < n_total) {
        // Every index below is derived from an OPAQUE copy of the thread id made inside the iteration: with plain loop-invariant
        // indices LLVM hoists a few hundred LDS / global address computations out of this loop and then spills them (and the
        // resident weights) to scratch.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6, row2 = tid >> 1, kh = tid & 1;
        (void)lane;
        // ---- frames available (bounded wait; the sequential schedule publishes everything before the launch) -----------------
        int avail = seen_ready;
        if (avail <= fidx) {
            if (tid == 0) {
                int nf = ld_sc1i(p.ctrl);
                const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                int err = 0;
                while (nf <= fidx) {
                    if (ld_sc1i(p.ctrl + 4) != 0 || (long long)__builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks) {
                        st_sc1i(p.ctrl + 1, 1);
                        st_sc1i(p.ctrl + 4, 1);
                        err = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                    nf = ld_sc1i(p.ctrl);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // one acquire per publication seen
                redi[0] = err ? -1 : nf;
            }
            __syncthreads();
            avail = redi[0];
            __syncthreads();
            if (avail < 0) { bad = true; break; }
            seen_ready = avail;
        }
        const int kf = dirty ? 1 : min(KF, min(avail, n_total) - fidx);
        // the encoder-projection rows of the pass are fetched now and used after the predictor step
        float er[KF * RNNT_D / 512];
#pragma unroll
        for (int i2 = 0; i2 < KF * RNNT_D / 512; ++i2) {
            const int e = tid + 512 * i2, k = e >> 8;
            er[i2] = k < kf ? ldg1(encp + (long long)(fidx + k) * RNNT_D + (e & 255)) : 0.f;
        }
        GM_T(0)
        if (dirty) {
            ++nsymev;
            // ---- predictor step on this part's 64 units (predictor.py:200-204), then X1 ---------------------------------------
            ++ev1;
            unsigned long long* xw = x1b + (ev1 & 1) * x1par;
#if GM_WHH_REGS
            {
                const float eg = ldg1(p.egate + (long long)tok * (4 * RNNT_D) + 256 * pw + row2);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const float4 hv = *reinterpret_cast<const float4*>(&hs[128 * kh + 4 * j]);
                    a0 = fmaf(whh_r[4 * j], hv.x, a0);
                    a1 = fmaf(whh_r[4 * j + 1], hv.y, a1);
                    a2 = fmaf(whh_r[4 * j + 2], hv.z, a2);
                    a3 = fmaf(whh_r[4 * j + 3], hv.w, a3);
                    if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
                float g = (a0 + a1) + (a2 + a3);
                g += __shfl_xor(g, 1, 64);
                if (!kh) gates[row2] = g + eg;
            }
#else
            {
                const float* eg = p.egate + (long long)tok * (4 * RNNT_D) + 256 * pw;
                dec_matvec<1, 512, 4>(p.whh + (long long)(256 * pw) * RNNT_D, RNNT_D, reinterpret_cast<const float (*)[RNNT_D]>(hs),
                                      [&](int n, const float* acc) { gates[n] = acc[0] + ldg1(eg + n); });
            }
#endif
            __syncthreads();
            GM_T(1)
            if (cell) {
                const int unit = 64 * pw + (tid >> 3);
                const float4 gt = *reinterpret_cast<const float4*>(&gates[4 * (tid >> 3)]);     // i, f, g, o of this unit
                cc2 = sigmoidf_(gt.y) * cc + sigmoidf_(gt.x) * tanhf(gt.z);
                const float hn = sigmoidf_(gt.w) * tanhf(cc2);
                h2[unit] = hn;
                st_tag(xw + pw * GM_X1 + (unit - 64 * pw), __float_as_uint(hn), ev1);
            }
            __syncthreads();
            {
                float q0 = 0.f, q1 = 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 hv = *reinterpret_cast<const float4*>(&h2[64 * pw + 32 * kh + 4 * u]);   // two addresses per wave: broadcast
                    q0 = fmaf(wcv[u].x, hv.x, q0);
                    q1 = fmaf(wcv[u].y, hv.y, q1);
                    q0 = fmaf(wcv[u].z, hv.z, q0);
                    q1 = fmaf(wcv[u].w, hv.w, q1);
                }
                float part = q0 + q1;
                part += __shfl_xor(part, 1, 64);
                if (!kh) {
                    mypp[row2] = part;
                    st_tag(xw + pw * GM_X1 + 64 + row2, __float_as_uint(part), ev1);
                }
            }
            __syncthreads();
            GM_T(2)
            // gather: threads 0..255 sum the four partials of pp[n] in part order; threads 256..447 fetch the other parts' h'
            if (tid < RNNT_D) {
                float sum = 0.f;
#pragma unroll
                for (int q = 0; q < GM_PARTS; ++q) {
                    float v = mypp[tid];
                    if (q != pw) {
                        unsigned w = 0;
                        if (!gm_poll(p, xw + q * GM_X1 + 64 + tid, ev1, w)) bad = true;
                        v = __uint_as_float(w);
                    }
                    sum = q == 0 ? v : sum + v;
                }
                pp[tid] = sum + ldg1(p.bjc + tid);
            } else if (tid < RNNT_D + 192) {
                const int idx = tid - RNNT_D;
                int q = idx >> 6;
                if (q >= pw) ++q;                              // the idx/64-th OTHER part
                unsigned w = 0;
                if (!gm_poll(p, xw + q * GM_X1 + (idx & 63), ev1, w)) bad = true;
                h2[64 * q + (idx & 63)] = __uint_as_float(w);
            }
            if (bad) *s_bad = 1;
            __syncthreads();
            if (*s_bad) { bad = true; break; }
            dirty = false;
            GM_T(3)
        }
        // ---- joint activations of kf frames (every part computes all 256) ------------------------------------------------------
#pragma unroll
        for (int i2 = 0; i2 < KF * RNNT_D / 512; ++i2) {
            const int e = tid + 512 * i2, k = e >> 8;
            zs[e] = k < kf ? tanhf(pp[e & 255] + er[i2]) : 0.f;
        }
        __syncthreads();
        // ---- logits of this part's vocabulary rows.  Thread (row = tid & 127, quarter of K = tid >> 7): a wave reads 64 consecutive
        // rows at one quarter, so the 16 lanes of a ds_read_b128 cycle hit 16 distinct slots (row stride 260 floats) and z is a
        // broadcast; the four quarter sums meet in LDS, then thread (frame, row) finishes the logit and the argmax runs per frame.
        {
            const int r = tid & 127, qk = tid >> 7;
            if (r < p.rows_per) {
                const float* wl = &Wo[r * GM_WLD + 64 * qk];
#pragma unroll
                for (int k = 0; k < KF; ++k) {
                    if (k < kf) {   // one frame at a time: with the frames innermost hipcc SLP-packs across frames and spills ~90 registers
                        const float* zk = &zs[k * RNNT_D + 64 * qk];
                        float a0 = 0.f, a1 = 0.f;
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const float4 w = *reinterpret_cast<const float4*>(wl + 4 * j);
                            const float4 z = *reinterpret_cast<const float4*>(zk + 4 * j);
                            a0 = fmaf(w.x, z.x, a0);
                            a1 = fmaf(w.y, z.y, a1);
                            a0 = fmaf(w.z, z.z, a0);
                            a1 = fmaf(w.w, z.w, a1);
                            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                        }
                        lpart[(k * 4 + qk) * 128 + r] = a0 + a1;
                    }
                }
            }
        }
        __syncthreads();
        {
            const int k = tid >> 7, r = tid & 127;         // KF * 128 = 512 threads
            const int vr = v0 + r;
            const bool vin = k < kf && r < p.rows_per && vr < p.vocab;
            float v = -INFINITY;
            int ix = 0x7fffffff;
            if (vin) {
                v = ((lpart[(k * 4) * 128 + r] + lpart[(k * 4 + 1) * 128 + r]) + (lpart[(k * 4 + 2) * 128 + r] + lpart[(k * 4 + 3) * 128 + r])) + ldg1(p.bout + vr);
                ix = vr;
            }
            // ascending butterfly; DPP partners for 1..8 (xor_partner: the (value, index) pair is uniform in quads / 8-groups by then)
#define GM_STEP(O_) { const float ov = xor_partner<O_>(v); const int oi = __builtin_bit_cast(int, xor_partner<O_>(__builtin_bit_cast(float, ix))); \
                      if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; } }
            GM_STEP(1) GM_STEP(2) GM_STEP(4) GM_STEP(8) GM_STEP(16) GM_STEP(32)
#undef GM_STEP
            if (lane == 0) { redv[wave] = v; redi[wave] = ix; }   // wave 2 k, 2 k + 1 = frame k
        }
        __syncthreads();
        GM_T(4)
        ++ev3;
        unsigned long long* xq = xab + (ev3 & 1) * xapar;
        if (tid < KF) {
            float v = redv[2 * tid];
            int ix = redi[2 * tid];
            {
                const float ov = redv[2 * tid + 1];
                const int oi = redi[2 * tid + 1];
                if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
            }
            unsigned u = 0u;
            if (ix != 0x7fffffff) {
                u = __float_as_uint(v);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // order-preserving; > 0 for every real value
            }
            st_tag(xq + pw * 2 * KF + 2 * tid, u, ev3);
            st_tag(xq + pw * 2 * KF + 2 * tid + 1, (unsigned)ix, ev3);
            xav[pw * 2 * KF + 2 * tid] = u;
            xav[pw * 2 * KF + 2 * tid + 1] = (unsigned)ix;
        } else if (tid >= 64 && tid < 64 + (GM_PARTS - 1) * 2 * KF) {
            const int idx = tid - 64;
            int q = idx / (2 * KF);
            const int w = idx - q * 2 * KF;
            if (q >= pw) ++q;
            unsigned val = 0;
            if (!gm_poll(p, xq + q * 2 * KF + w, ev3, val)) *s_bad = 1;
            xav[q * 2 * KF + w] = val;
        }
        __syncthreads();
        GM_T(5)
        if (*s_bad) { bad = true; break; }
        // ---- decisions, in frame order (identical in every thread of every part) ---------------------------------------------------
        bool commit = false;
        for (int k = 0; k < kf; ++k) {
            unsigned bv = 0u;
            int w = 0x7fffffff;
#pragma unroll
            for (int q = 0; q < GM_PARTS; ++q) {
                const unsigned v = xav[q * 2 * KF + 2 * k];
                const int ix = (int)xav[q * 2 * KF + 2 * k + 1];
                if (v > bv || (v == bv && ix < w)) { bv = v; w = ix; }
            }
            if (w == p.blank) { fidx += 1; nsym = 0; continue; }
            if (pw == 0 && tid == 0 && count < p.max_tokens) p.tokens[(long long)b * p.max_tokens + count] = w;
            count += 1;
            tok = w;
            nsym += 1;
            if (nsym >= p.n_steps) { nsym = 0; fidx += 1; }
            commit = true;
            break;
        }
        __syncthreads();                                     // xav / redv fully consumed before the next pass rewrites them
        if (commit) {
            if (tid < RNNT_D) hs[tid] = h2[tid];
            cc = cc2;
            dirty = true;
            __syncthreads();
        }
        ++evals;
        GM_T(6)
    }
    if (tdbg) {
        for (int k = 0; k < 8; ++k) p.dbg[k] = tacc[k];
        p.dbg[8] = evals; p.dbg[9] = nsymev;
    }
#undef GM_T
    // ---- canonical state for the host / the next call (buffer 0 becomes the committed one) ---------------------------------
    if (pw == 0 && tid0 < RNNT_D) stg1(p.h + (long long)b * RNNT_D + tid0, hs[tid0]);
    if (cell) stg1(p.c + (long long)b * RNNT_D + cell_unit, cc);
    if (pw == 0 && tid0 == 0) {
        p.sel[b] = 0; p.tok[b] = tok; p.fidx[b] = fidx; p.nsym[b] = nsym; p.count[b] = count;
        atomicAdd(p.ctrl + 2, evals);
    }
    (void)bad;
}
