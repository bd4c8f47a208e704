// Beam search kernels: per-step reduction (launched path), per-hypothesis extension chain, state-pool gather, lattice log-softmax.
// Part of rnnt_kernels.hip.h (include that umbrella, not this file).
#pragma once

// ------------------------------------------------------------------------------------------------
// beam_reduce: one wave per hypothesis row, one step of the extension chain of
// _decode_chunk_beam_search (online_rnnt_model.py:446-499): log_softmax statistics, blank log-prob,
// top-k non-blank (value desc, index asc), stop test `blank >= max - 1e-6` in double (:486), else the
// row's next predictor input is its best non-blank token.
// ------------------------------------------------------------------------------------------------
struct BeamOut {
    int* active;      // [R]
    int* tok;         // [R] predictor input token (updated when the chain continues)
    int* steps;       // [R] steps evaluated so far
    float* blank_lp;  // [R][n_steps]
    float* top_lp;    // [R][n_steps][k]
    int* top_tok;     // [R][n_steps][k]
    int* n_active;    // [1]
};

// Any vocabulary size: strided passes over the row (the first version kept the row in 8 registers per lane and silently
// ignored tokens >= 512); k <= 64 winners are excluded through a small LDS list.
__global__ __launch_bounds__(64) void beam_reduce(const float* __restrict__ logits, int ldl, int vocab, int blank, int k, int step,
                                                int n_steps, BeamOut o) {
    __shared__ int chosen[64];
    const int r = blockIdx.x, lane = threadIdx.x;
    if (!o.active[r]) return;
    const float* x = logits + (long long)r * ldl;
    float mx = -INFINITY;
    for (int idx = lane; idx < vocab; idx += 64) mx = fmaxf(mx, x[idx]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int idx = lane; idx < vocab; idx += 64) se += expf(x[idx] - mx);
    const float lse = logf(wave_sum(se));
    const float blank_lp = (x[blank] - mx) - lse;
    const float max_lp = (mx - mx) - lse;
    int best_tok = 0;
    for (int t = 0; t < k; ++t) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int idx = lane; idx < vocab; idx += 64) {
            if (idx == blank) continue;
            bool taken = false;
            for (int c = 0; c < t; ++c) taken = taken || chosen[c] == idx;
            if (taken) continue;
            const float lp = (x[idx] - mx) - lse;
            if (lp > bv) { bv = lp; bi = idx; }          // ascending scan: ties keep the lower index
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) {
            chosen[t] = bi;
            o.top_lp[((long long)r * n_steps + step) * k + t] = bv;
            o.top_tok[((long long)r * n_steps + step) * k + t] = bi;
        }
        __syncthreads();
        if (t == 0) best_tok = bi;
    }
    if (lane == 0) {
        o.blank_lp[(long long)r * n_steps + step] = blank_lp;
        o.steps[r] = step + 1;
        const bool stop = ((double)blank_lp >= (double)max_lp - 1e-6) || (step + 1 >= n_steps);
        if (stop) {
            o.active[r] = 0;
            atomicSub(o.n_active, 1);
        } else {
            o.tok[r] = best_tok;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// beam_chain: the whole extension chain of ONE hypothesis row for one encoder frame as one workgroup (512 threads) --
// the loop of _decode_chunk_beam_search (online_rnnt_model.py:446-499) that the launched path runs as 5 kernels and one
// host synchronisation per step: predictor step (table row + W_hh product + cell), projection, joint.pred_ffn,
// tanh(enc_ffn(enc)[t] + .), vocabulary projection, log-softmax statistics, blank log-prob, top-k non-blank, stop test,
// next input token = best non-blank.  Rows are independent, so 64 streams x 4 hypotheses fill the 256 CUs; every
// intermediate LSTM state goes to the row's pool slots exactly as in the launched path.
// ------------------------------------------------------------------------------------------------
struct BeamChainP {
    const float* whh; const float* egate; const float* wpr; const float* bpr; const float* wpf; const float* bpf;
    const float* wout; const float* bout; const float* encp;
    float* pool;                 // [R][slots][512] (h | c); slot 0 = state before the first evaluation
    const int* frame;            // [R] row of encp
    const int* tok_in;           // [R] predictor input token of the first evaluation
    int* steps; float* blank_lp; float* top_lp; int* top_tok;
    int vocab, blank, k, n_steps, slots;
};

__global__ __launch_bounds__(512) void beam_chain(BeamChainP p) {
    constexpr int NTH = 512;
    __shared__ __attribute__((aligned(16))) float hs[1][RNNT_D], cs[RNNT_D], h2[1][RNNT_D], pr[1][RNNT_D], zs[1][RNNT_D];
    __shared__ __attribute__((aligned(16))) float gates[4 * RNNT_D];
    __shared__ float lg[512];
    __shared__ int s_ctl[2];
    const int tid = threadIdx.x, r = blockIdx.x;
    float* pool = p.pool + (long long)r * p.slots * 512;
    if (tid < RNNT_D) { hs[0][tid] = ldg1(pool + tid); cs[tid] = ldg1(pool + RNNT_D + tid); }
    int tok = ldgi(p.tok_in + r);
    const float* enc = p.encp + (long long)ldgi(p.frame + r) * RNNT_D;
    __syncthreads();
    int st = 0;
    for (; st < p.n_steps; ++st) {
        // predictor.forward_step (predictor.py:185-210): LSTM cell on (embed[tok], state slot st) -> slot st + 1
        dec_matvec<1, NTH>(p.whh, 4 * RNNT_D, hs, [&](int n, const float* acc) {
            gates[n] = acc[0] + ldg1(p.egate + (long long)tok * (4 * RNNT_D) + n);
        });
        __syncthreads();
        if (tid < RNNT_D) {
            const float4 gt = *reinterpret_cast<const float4*>(&gates[4 * tid]);
            const float cc = sigmoidf_(gt.y) * cs[tid] + sigmoidf_(gt.x) * tanhf(gt.z);
            const float hh = sigmoidf_(gt.w) * tanhf(cc);
            cs[tid] = cc;                                          // the chain continues from the new state
            h2[0][tid] = hh;
            stg1(pool + (long long)(st + 1) * 512 + tid, hh);
            stg1(pool + (long long)(st + 1) * 512 + RNNT_D + tid, cc);
        }
        __syncthreads();
        if (tid < RNNT_D) hs[0][tid] = h2[0][tid];
        dec_matvec<1, NTH>(p.wpr, RNNT_D, h2, [&](int n, const float* acc) { pr[0][n] = acc[0] + ldg1(p.bpr + n); });   // predictor.projection
        __syncthreads();
        dec_matvec<1, NTH>(p.wpf, RNNT_D, pr, [&](int n, const float* acc) {                                           // joint (joint.py:54-66)
            zs[0][n] = tanhf(acc[0] + ldg1(p.bpf + n) + ldg1(enc + n));
        });
        __syncthreads();
        dec_matvec<1, NTH>(p.wout, p.vocab, zs, [&](int n, const float* acc) { lg[n] = acc[0] + ldg1(p.bout + n); });
        __syncthreads();
        if (tid < 64) {   // log_softmax statistics, blank log-prob, top-k non-blank (value desc, index asc), stop test (:468,:486)
            const int lane = tid;
            float v[8];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = lane + 64 * j;
                v[j] = idx < p.vocab ? lg[idx] : -INFINITY;
                mx = fmaxf(mx, v[j]);
            }
            mx = wave_max(mx);
            float se = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) se += (lane + 64 * j) < p.vocab ? expf(v[j] - mx) : 0.f;
            const float lse = logf(wave_sum(se));
            const float blank_lp = (lg[p.blank] - mx) - lse;
            const float max_lp = (mx - mx) - lse;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = lane + 64 * j;
                v[j] = (idx < p.vocab && idx != p.blank) ? (v[j] - mx) - lse : -INFINITY;
            }
            int best_tok = 0;
            for (int t = 0; t < p.k; ++t) {
                float bv = -INFINITY;
                int bi = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (v[j] > bv) { bv = v[j]; bi = lane + 64 * j; }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(bv, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                if ((bi & 63) == lane) v[bi >> 6] = -INFINITY;   // remove the winner
                if (lane == 0) {
                    p.top_lp[((long long)r * p.n_steps + st) * p.k + t] = bv;
                    p.top_tok[((long long)r * p.n_steps + st) * p.k + t] = bi;
                }
                if (t == 0) best_tok = bi;
            }
            if (lane == 0) {
                p.blank_lp[(long long)r * p.n_steps + st] = blank_lp;
                s_ctl[0] = ((double)blank_lp >= (double)max_lp - 1e-6) || (st + 1 >= p.n_steps) ? 1 : 0;
                s_ctl[1] = best_tok;
            }
        }
        __syncthreads();
        if (s_ctl[0]) { ++st; break; }
        tok = s_ctl[1];
        __syncthreads();
    }
    if (tid == 0) p.steps[r] = st;
}

// new_pool[r][0] <- old_pool[src_row[r]][src_step[r]]  (state = [h(256) | c(256)])
__global__ void beam_gather(const float* __restrict__ old_pool, float* __restrict__ new_pool, const int* __restrict__ src_row,
                            const int* __restrict__ src_step, int n_new, int slots) {
    const int r = blockIdx.x;
    if (r >= n_new) return;
    const float* s = old_pool + ((long long)src_row[r] * slots + src_step[r]) * 512;
    float* d = new_pool + (long long)r * slots * 512;
    for (int i = threadIdx.x; i < 512; i += blockDim.x) d[i] = s[i];
}

// log_softmax over the last dimension, in place, one wave per row (joint lattice mode 1).  A row (n <= 512 floats) is read
// ONCE into registers (8 values per lane), reduced, and written once: the pass is a pure HBM stream of 2 x rows x n x 4 B.
// Rows longer than 512 take the three-pass loop.
__global__ __launch_bounds__(256) void log_softmax_rows(float* __restrict__ x, long long rows, int n) {
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = x + row * n;
    if (n <= 512) {
        float v[8];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = lane + 64 * j;
            v[j] = idx < n ? ldg1(p + idx) : -INFINITY;
            mx = fmaxf(mx, v[j]);
        }
        mx = wave_max(mx);
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) se += (lane + 64 * j) < n ? expf(v[j] - mx) : 0.f;
        const float lse = logf(wave_sum(se));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = lane + 64 * j;
            if (idx < n) stg1(p + idx, v[j] - mx - lse);
        }
        return;
    }
    float mx = -INFINITY;
    for (int v = lane; v < n; v += 64) mx = fmaxf(mx, p[v]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int v = lane; v < n; v += 64) s += expf(p[v] - mx);
    s = logf(wave_sum(s));
    for (int v = lane; v < n; v += 64) p[v] = p[v] - mx - s;
}
