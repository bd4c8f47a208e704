// Layer-major whole-utterance encoder kernels (rnnt_encoder_chunks when every chunk of the call is already in HBM).
// Part of rnnt_kernels.hip.h (include that umbrella, not this file).
//
// Layer l of chunk c needs layer l-1 of the SAME chunk (per frame) and, from layer l's own earlier chunks, only their K/V rows
// and post-GLU conv rows -- both functions of layer l's INPUT, not of its attention output (encoder.py:274-288,
// convolution.py:122-130).  So a whole layer can run over all chunks of the call at once: every per-frame contraction becomes
// one GEMM over M = streams x frames rows, and what remains chunk-aware is (1) attention, where a query of chunk c sees exactly
// the keys the chunk-by-chunk loop would have cached for it, at that chunk's positional window, and (2) the causal depthwise
// conv, which is simply a causal conv over the utterance with the stream's 30-row left context.
#pragma once

// ------------------------------------------------------------------------------------------------
// Attention of ALL chunks of a call (RelPositionMultiHeadedAttention scores / softmax / PV, attention.py:400-418,170-177).
// Every query frame is a ROW with its own key window: the absolute cache rows [ks, ke) of its stream that the chunk-by-chunk
// loop would have cached for its chunk, and the positional row of cache row a is a + pshift (pshift = pos_start - kv_start:
// encoder.py:257 rebuilt per chunk, no rel_shift).  A workgroup takes LM_ROWS consecutive rows (about ten 3-frame chunks) and one
// (stream, head): K / V / positional rows are staged ONCE per 64-key tile for all of them (the chunk-by-chunk kernels re-read
// the whole cache for every chunk: 9.3 GB per 64 x 10 s batch).  The rows' positional windows differ by a few rows: the
// positional tile carries LM_PEXT extra rows.  Kernel: rel_attention_lm_mfma below.
//   q, out  [B*F][256] stream-major rows (b*F + f);  kc, vc [B][kv_stride][256];  ptab [5000][256] of this layer
// ------------------------------------------------------------------------------------------------
#define LM_ROWS 32
#define LM_PEXT 16
struct LmRow { int f, ks, ke, pshift; };   // f < 0: unused slot
struct LmBlock {
    int n_rows, amin, amax, pmin;          // key rows [amin, amax) cover every row's window; pmin = smallest pshift
    LmRow r[LM_ROWS];
};
struct LmAttnP {
    const float* q;
    const float* kc;
    const float* vc;
    const float* ptab;
    const float* bias_u;
    const float* bias_v;
    float* out;
    const LmBlock* blocks;
    int F;
    long long kv_stride;
    const int* klen;     // per-stream number of valid keys counted from a row's ks (full-context padding mask); null = no limit
    int per_stream;      // 0: one block table for every stream; 1: `blocks` is [B][gridDim.y] (ragged batch: every stream has its own chunk plan)
};

// ------------------------------------------------------------------------------------------------
// Ragged batches (rnnt_decode_ragged): every stream's LAST chunk has its own start and length (online_rnnt_decode.py:88-91 merges
// a short remainder into it), so the tails are subsampled per tail-length class: their fbank frames are gathered into a compact
// [n][len][80] buffer, run through the ordinary subsampling of one chunk, and the resulting encoder-input rows scattered to the
// stream's rows of the layer-major activation buffer.  ent = (stream, first fbank frame / first destination frame).
// ------------------------------------------------------------------------------------------------
__global__ void lm_gather_tails(const float* __restrict__ fbank, float* __restrict__ dst, const int2* __restrict__ ent, int n, int len, int Tstride) {
    const long long tot = (long long)n * len * RNNT_IDIM;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < tot; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % RNNT_IDIM);
        const long long r = id / RNNT_IDIM;
        const int t = (int)(r % len), v = (int)(r / len);
        const int2 e = ent[v];
        dst[id] = ldg1(fbank + ((long long)e.x * Tstride + e.y + t) * RNNT_IDIM + c);
    }
}
__global__ void lm_scatter_rows(const float* __restrict__ src, float* __restrict__ x, const int2* __restrict__ ent, int n, int tq, int F) {
    const long long tot = (long long)n * tq * RNNT_D;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < tot; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id & 255);
        const long long r = id >> 8;
        const int t = (int)(r % tq), v = (int)(r / tq);
        const int2 e = ent[v];
        x[((long long)e.x * F + e.y + t) * RNNT_D + c] = src[id];
    }
}

// ------------------------------------------------------------------------------------------------
// lm_ctx_in: the 30-row left context of every layer's linear post-GLU buffer <- the stream's ring rows before `pos`
// (a fresh stream's ring holds GLU(b_pw1): the zero pad goes through the biased pointwise conv, convolution.py:122-124,138).
// g_lin [L][B][gs][256], ring [L][Bmax][cap][256].
// ------------------------------------------------------------------------------------------------
__global__ void lm_ctx_in(const float* __restrict__ ring, float* __restrict__ g_lin, int B, int Bmax, int cap, int gs, int pos) {
    const long long n = (long long)RNNT_L * B * RNNT_LORDER * RNNT_D;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id & 255);
        long long r = id >> 8;
        const int i = (int)(r % RNNT_LORDER);
        r /= RNNT_LORDER;
        const int b = (int)(r % B), l = (int)(r / B);
        const int row = (pos - RNNT_LORDER + i + cap * 64) % cap;
        g_lin[(((long long)l * B + b) * gs + i) * RNNT_D + c] = ldg1(ring + (((long long)l * Bmax + b) * cap + row) * RNNT_D + c);
    }
}

// ------------------------------------------------------------------------------------------------
// dwconv_lm: causal depthwise conv k=31 + BatchNorm(eval) + SiLU (convolution.py:142-145) over the linear post-GLU rows of one
// layer, g [B][gs][256] with 30 left-context rows first; frame f of stream b = row 30 + f.  One thread = one channel and
// LM_FB consecutive frames (sliding window in registers: 38 loads for 8 outputs); taps summed in the order of dwconv_bn_silu.
// Also leaves the stream state the chunk-by-chunk path would leave: the last `cap` frames' post-GLU rows and conv-module
// input rows in the two rings at (pos + f) % cap.
// ------------------------------------------------------------------------------------------------
#define LM_FB 8
struct DwLmP {
    const float* g;
    const float* wdw_t;
    const float* bdw;
    const float* bn_s;
    const float* bn_t;
    float* out;            // [B*F][256]
    const float* xres;     // [B*F][256] conv-module input (before norm_conv)
    float* gring;          // [Bmax][cap][256] of this layer
    float* xring;
    int B, F, gs, cap, pos;
};
__global__ void dwconv_lm(DwLmP P) {
    const int nfb = (P.F + LM_FB - 1) / LM_FB;
    const long long n = (long long)P.B * nfb * RNNT_D;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id & 255);
        const int r = (int)(id >> 8);
        const int b = r / nfb, f0 = (r - b * nfb) * LM_FB;
        const float* gb = P.g + ((long long)b * P.gs + f0) * RNNT_D + c;
        float win[LM_FB + RNNT_LORDER];
#pragma unroll
        for (int i = 0; i < LM_FB + RNNT_LORDER; ++i) win[i] = (f0 + i < P.F + RNNT_LORDER) ? ldg1(gb + (long long)i * RNNT_D) : 0.f;
        float w[RNNT_KDW];
#pragma unroll
        for (int k = 0; k < RNNT_KDW; ++k) w[k] = ldg1(P.wdw_t + k * RNNT_D + c);
        const float bd = ldg1(P.bdw + c), bs = ldg1(P.bn_s + c), bt = ldg1(P.bn_t + c);
#pragma unroll
        for (int j = 0; j < LM_FB; ++j) {
            const int f = f0 + j;
            if (f >= P.F) break;
            float acc = bd;
#pragma unroll
            for (int k = 0; k < RNNT_KDW; ++k) acc = fmaf(w[k], win[j + k], acc);
            float v = acc * bs + bt;
            v = v * sigmoidf_(v);
            const long long m = (long long)b * P.F + f;
            stg1(P.out + m * RNNT_D + c, v);
            if (f >= P.F - P.cap) {
                const long long rr = ((long long)b * P.cap + (P.pos + f) % P.cap) * RNNT_D + c;
                stg1(P.gring + rr, win[j + RNNT_LORDER]);
                stg1(P.xring + rr, ldg1(P.xres + m * RNNT_D + c));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rel_attention_lm_mfma: the same attention with the three contractions on the exact-f32 matrix instruction
// (v_mfma_f32_16x16x4_f32: bit-for-bit an fmaf chain, 2x the scalar FMA rate, and -- what matters here -- operands read from
// LDS once per 16 x 16 output tile instead of once per output).  rel_attention_lm is bound by its LDS reads (every lane
// re-reads the query rows for every key: 185 us per layer at 64 x 10 s); this form is bound by the K / V / positional tiles it
// stages (53 KB per 64 keys and workgroup).
//   A  scores: S_ac[32 q x 64 keys] = (Q + u) K^T and G[32 q x 80 rows] = (Q + v) P^T for the 80 positional rows the block's
//      units can address in this key tile, both to LDS; a unit's matrix_bd is G shifted by its window offset
//      (encoder.py:257: the positional window is rebuilt per chunk; attention.py:406-409: no rel_shift)
//   B  softmax: lane = key, one wave per 8 query slots, online maximum / sum; probabilities overwrite S_ac
//   C  O[32 q x 64] = alpha O + P V on the accumulators
// 4 waves: wave w owns query tile w & 1 (16 slots = 4 units) and half w >> 1 of the key tiles (A) / d tiles (C).
// grid = (B*H, n_blocks), block = 256, dynamic LDS = LM2_LDS bytes.  Same unit / block tables as rel_attention_lm.
// ------------------------------------------------------------------------------------------------
#define LM2_LD 68
#define LM2_GLD 84
#define LM2_LDS ((64 * LM2_LD + (64 + LM_PEXT) * LM2_LD + 64 * LM2_LD + 32 * LM2_LD + 32 * LM2_GLD + 64) * 4)
__global__ __launch_bounds__(256) void rel_attention_lm_mfma(LmAttnP P) {
    extern __shared__ __attribute__((aligned(16))) float lm2_smem[];
    float* Ks = lm2_smem;                               // [64][68]
    float* Ps = Ks + 64 * LM2_LD;                       // [80][68]
    float* Vs = Ps + (64 + LM_PEXT) * LM2_LD;           // [64][68]
    float* Sx = Vs + 64 * LM2_LD;                       // [32][68]  matrix_ac scores, then probabilities
    float* Gx = Sx + 32 * LM2_LD;                       // [32][84]  (Q + v) P^T over the tile's 80 positional rows
    float* al = Gx + 32 * LM2_GLD;                      // [32] rescale factor of the tile, then 1 / sum
    typedef float f32x4m __attribute__((ext_vector_type(4)));
    const int b = blockIdx.x / RNNT_H, h = blockIdx.x % RNNT_H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int qt = wave & 1, hf = wave >> 1;
    const LmBlock* __restrict__ blk = P.blocks + (P.per_stream ? (long long)b * gridDim.y : 0) + blockIdx.y;
    const int amin = ldgi(&blk->amin), amax = ldgi(&blk->amax), pmin = ldgi(&blk->pmin);
    // softmax role: rows 8*wave .. +7, each with its own key window
    int s_ks[8], s_ke[8], s_prel[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const LmRow* rw = &blk->r[8 * wave + j];
        s_ks[j] = 0; s_ke[j] = 0; s_prel[j] = 0;
        if (ldgi(&rw->f) >= 0) {
            s_ks[j] = ldgi(&rw->ks); s_ke[j] = ldgi(&rw->ke);
            if (P.klen) s_ke[j] = min(s_ke[j], s_ks[j] + ldgi(P.klen + b));
            s_prel[j] = ldgi(&rw->pshift) - pmin;
        }
    }
    // contraction role: this lane's query row 16*qt + i
    const int my_f = ldgi(&blk->r[16 * qt + i].f);
    float4 qu[4], qv[4];
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
        const int d = 16 * sp + 4 * kq;
        float4 qq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (my_f >= 0) qq = ldg4(P.q + ((long long)b * P.F + my_f) * RNNT_D + h * RNNT_DK + d);
        const float4 bu = ldg4(P.bias_u + h * RNNT_DK + d), bv = ldg4(P.bias_v + h * RNNT_DK + d);
        qu[sp] = make_float4(qq.x + bu.x, qq.y + bu.y, qq.z + bu.z, qq.w + bu.w);
        qv[sp] = make_float4(qq.x + bv.x, qq.y + bv.y, qq.z + bv.z, qq.w + bv.w);
    }
    f32x4m o[2];
    o[0] = (f32x4m){0.f, 0.f, 0.f, 0.f};
    o[1] = (f32x4m){0.f, 0.f, 0.f, 0.f};
    float mrun[8], lrun[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mrun[j] = -INFINITY; lrun[j] = 0.f; }
    const float* kbase = P.kc + (long long)b * P.kv_stride * RNNT_D + h * RNNT_DK;
    const float* vbase = P.vc + (long long)b * P.kv_stride * RNNT_D + h * RNNT_DK;
    const float* pbase = P.ptab + h * RNNT_DK;
    // K / V / positional rows of a tile travel global -> registers -> LDS; the NEXT tile's loads are issued as soon as this tile's
    // rows are in LDS, so their L2 latency hides behind the three phases (13 float4 per thread in flight)
    float4 rk[4], rv[4], rp[5];
    auto tile_load = [&](int a0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j, r = e >> 4, c4 = e & 15;
            rk[j] = make_float4(0.f, 0.f, 0.f, 0.f); rv[j] = rk[j];
            if (a0 + r < amax) {
                rk[j] = ldg4(kbase + (long long)(a0 + r) * RNNT_D + c4 * 4);
                rv[j] = ldg4(vbase + (long long)(a0 + r) * RNNT_D + c4 * 4);
            }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int e = tid + 256 * j, r = e >> 4, c4 = e & 15;
            const int pr = a0 + pmin + r;
            rp[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pr >= 0 && pr < RNNT_PE_LEN) rp[j] = ldg4(pbase + (long long)pr * RNNT_D + c4 * 4);
        }
    };
    tile_load(amin);
    for (int a0 = amin; a0 < amax; a0 += 64) {
        __syncthreads();                                            // the previous tile's PV is done with Vs / Sx / al
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j, r = e >> 4, c4 = e & 15;
            *reinterpret_cast<float4*>(&Ks[r * LM2_LD + c4 * 4]) = rk[j];
            *reinterpret_cast<float4*>(&Vs[r * LM2_LD + c4 * 4]) = rv[j];
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int e = tid + 256 * j, r = e >> 4, c4 = e & 15;
            *reinterpret_cast<float4*>(&Ps[r * LM2_LD + c4 * 4]) = rp[j];
        }
        __syncthreads();
        if (a0 + 64 < amax) tile_load(a0 + 64);
        // ---- A: matrix_ac tiles and G tiles of this wave's query tile ------------------------------------------------------------
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kt = 2 * hf + kk;
            f32x4m acc = (f32x4m){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                const float4 kf = *reinterpret_cast<const float4*>(&Ks[(16 * kt + i) * LM2_LD + 16 * sp + 4 * kq]);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qu[sp].x, kf.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qu[sp].y, kf.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qu[sp].z, kf.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qu[sp].w, kf.w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Sx[(16 * qt + 4 * kq + r) * LM2_LD + 16 * kt + i] = acc[r];
        }
        for (int gt = (hf ? 3 : 0); gt < (hf ? 5 : 3); ++gt) {
            f32x4m acc = (f32x4m){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                const float4 pf = *reinterpret_cast<const float4*>(&Ps[(16 * gt + i) * LM2_LD + 16 * sp + 4 * kq]);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[sp].x, pf.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[sp].y, pf.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[sp].z, pf.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[sp].w, pf.w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Gx[(16 * qt + 4 * kq + r) * LM2_GLD + 16 * gt + i] = acc[r];
        }
        __syncthreads();
        // ---- B: softmax of query slots 8*wave .. +7, lane = key.  The eight rows' reductions advance in lock step (step outer,
        //      row inner): eight independent cross-lane exchanges per step instead of 96 dependent ones ---------------------------------
        {
            float sc[8], mx[8], pe_[8], sm[8];
            bool live[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) live[j] = a0 < s_ke[j] && a0 + 64 > s_ks[j];   // wave-uniform
            const int a = a0 + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int qs = 8 * wave + j;
                const bool valid = live[j] && a >= s_ks[j] && a < s_ke[j];
                sc[j] = valid ? (Sx[qs * LM2_LD + lane] + Gx[qs * LM2_GLD + lane + s_prel[j]]) * 0.125f : -INFINITY;
                mx[j] = sc[j];
            }
#pragma unroll
            for (int o_ = 32; o_ > 0; o_ >>= 1)
#pragma unroll
                for (int j = 0; j < 8; ++j) mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], o_, 64));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float mnew = fmaxf(mrun[j], mx[j]);
                pe_[j] = sc[j] > -INFINITY ? expf(sc[j] - mnew) : 0.f;
                sm[j] = pe_[j];
                mx[j] = live[j] ? expf(mrun[j] - mnew) : 1.0f;               // alpha (first live tile: exp(-inf) = 0)
                if (live[j]) mrun[j] = mnew;
            }
#pragma unroll
            for (int o_ = 32; o_ > 0; o_ >>= 1)
#pragma unroll
                for (int j = 0; j < 8; ++j) sm[j] += __shfl_xor(sm[j], o_, 64);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int qs = 8 * wave + j;
                lrun[j] = lrun[j] * mx[j] + sm[j];
                Sx[qs * LM2_LD + lane] = pe_[j];
                if (lane == 0) al[qs] = mx[j];
            }
        }
        __syncthreads();
        // ---- C: O = alpha O + P V for d tiles 2*hf, 2*hf + 1 ---------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a_ = al[16 * qt + 4 * kq + r];
            o[0][r] *= a_;
            o[1][r] *= a_;
        }
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            const float4 pf = *reinterpret_cast<const float4*>(&Sx[(16 * qt + i) * LM2_LD + 16 * sp + 4 * kq]);
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) {
                const float* vp = Vs + (16 * sp + 4 * kq) * LM2_LD + 16 * (2 * hf + dd) + i;
                o[dd] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.x, vp[0], o[dd], 0, 0, 0);
                o[dd] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.y, vp[LM2_LD], o[dd], 0, 0, 0);
                o[dd] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.z, vp[2 * LM2_LD], o[dd], 0, 0, 0);
                o[dd] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.w, vp[3 * LM2_LD], o[dd], 0, 0, 0);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (lane == 0) al[8 * wave + j] = lrun[j] > 0.f ? 1.0f / lrun[j] : 0.f;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * qt + 4 * kq + r;                       // query row of accumulator register r
        const int fr = ldgi(&blk->r[row].f);
        if (fr >= 0) {
            const long long m = (long long)b * P.F + fr;
            const float li = al[row];
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) stg1(P.out + m * RNNT_D + h * RNNT_DK + 16 * (2 * hf + dd) + i, o[dd][r] * li);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rel_attention_lm_bf<NSPLIT,F16>: rel_attention_lm_mfma with the three contractions on v_mfma_f32_16x16x32_{bf16,f16} (round 3).
// The f32 matrix instruction runs at 1/16 of the 16-bit rate: K = 64 of one 16 x 16 score tile costs 16 x 32 = 512 cycles there
// and 2 k-steps x 3 products x 16 = 96 cycles here (x = hi + lo, a_lo b_hi + a_hi b_lo + a_hi b_hi, f32 accumulate: the split
// modes' error model, ~1e-5 relative on a score).  Same block / row tables, same phases (A scores, B softmax in f32, C PV), same
// wave roles as the f32 kernel; what changes is the operand path:
//   * K and the positional rows are staged as 16-bit plane images [row][64 d] (128-byte rows, 16-byte chunk c of row r at chunk
//     c ^ ((r >> 1) & 7): the 16 lanes one ds_read_b128 lane group serves hit 16 distinct slots), split while they are staged;
//   * V is staged TRANSPOSED, [d][64 keys], because the PV product sums over keys and an MFMA operand is k-contiguous per lane:
//     a thread loads one 4-float piece of four consecutive keys and writes, per d, the four keys' halves as one 8-byte store;
//   * (Q + u), (Q + v) are split once per workgroup into operand registers; the probabilities are split when phase C reads them.
// Softmax statistics, rescaling and the output stay f32.  grid = (B*H, n_blocks), block = 256, dynamic LDS = LMB_LDS bytes.
// ------------------------------------------------------------------------------------------------
#define LMB_LDS (4 * 8192 + 2 * 10240 + (32 * LM2_LD + 32 * LM2_GLD + 64) * 4)
__device__ __forceinline__ int lmb_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
template <int NSPLIT, bool F16>
__global__ __launch_bounds__(256) void rel_attention_lm_bf(LmAttnP P) {
    constexpr bool LO = NSPLIT == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lmb_smem[];
    unsigned char* Kh = lmb_smem;                       // [64 keys][64 d] 16-bit, hi plane
    unsigned char* Kl = Kh + 8192;
    unsigned char* Ph = Kl + 8192;                      // [80 positional rows][64 d]
    unsigned char* Pl = Ph + 10240;
    unsigned char* Vh = Pl + 10240;                     // [64 d][64 keys] (transposed)
    unsigned char* Vl = Vh + 8192;
    float* Sx = reinterpret_cast<float*>(Vl + 8192);    // [32][68]  matrix_ac scores, then probabilities
    float* Gx = Sx + 32 * LM2_LD;                       // [32][84]  (Q + v) P^T over the tile's 80 positional rows
    float* al = Gx + 32 * LM2_GLD;                      // [32] rescale factor of the tile, then 1 / sum
    const int b = blockIdx.x / RNNT_H, h = blockIdx.x % RNNT_H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int qt = wave & 1, hf = wave >> 1;
    const LmBlock* __restrict__ blk = P.blocks + (P.per_stream ? (long long)b * gridDim.y : 0) + blockIdx.y;
    const int amin = ldgi(&blk->amin), amax = ldgi(&blk->amax), pmin = ldgi(&blk->pmin);
    int s_ks[8], s_ke[8], s_prel[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const LmRow* rw = &blk->r[8 * wave + j];
        s_ks[j] = 0; s_ke[j] = 0; s_prel[j] = 0;
        if (ldgi(&rw->f) >= 0) {
            s_ks[j] = ldgi(&rw->ks); s_ke[j] = ldgi(&rw->ke);
            if (P.klen) s_ke[j] = min(s_ke[j], s_ks[j] + ldgi(P.klen + b));
            s_prel[j] = ldgi(&rw->pshift) - pmin;
        }
    }
    // contraction role: this lane's query row 16*qt + i, d = 32 s + 8 kq .. + 8 of k-step s: (Q + u) and (Q + v) as operand planes
    const int my_f = ldgi(&blk->r[16 * qt + i].f);
    uint4 quh[2], qul[2], qvh[2], qvl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int d = 32 * s + 8 * kq;
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
        if (my_f >= 0) {
            const float* qp = P.q + ((long long)b * P.F + my_f) * RNNT_D + h * RNNT_DK + d;
            q0 = ldg4(qp); q1 = ldg4(qp + 4);
        }
        const float4 u0 = ldg4(P.bias_u + h * RNNT_DK + d), u1 = ldg4(P.bias_u + h * RNNT_DK + d + 4);
        const float4 v0 = ldg4(P.bias_v + h * RNNT_DK + d), v1 = ldg4(P.bias_v + h * RNNT_DK + d + 4);
        split8_16<F16, LO>(make_float4(q0.x + u0.x, q0.y + u0.y, q0.z + u0.z, q0.w + u0.w), make_float4(q1.x + u1.x, q1.y + u1.y, q1.z + u1.z, q1.w + u1.w), quh[s], qul[s]);
        split8_16<F16, LO>(make_float4(q0.x + v0.x, q0.y + v0.y, q0.z + v0.z, q0.w + v0.w), make_float4(q1.x + v1.x, q1.y + v1.y, q1.z + v1.z, q1.w + v1.w), qvh[s], qvl[s]);
    }
    f32x4_ o[2];
    o[0] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    o[1] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    float mrun[8], lrun[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mrun[j] = -INFINITY; lrun[j] = 0.f; }
    const float* kbase = P.kc + (long long)b * P.kv_stride * RNNT_D + h * RNNT_DK;
    const float* vbase = P.vc + (long long)b * P.kv_stride * RNNT_D + h * RNNT_DK;
    const float* pbase = P.ptab + h * RNNT_DK;
    // staging roles.  K / positional rows: 8-float chunk c8 of row r, e = tid + 256 j -> (r = e >> 3, c8 = e & 7); V: the 4-float piece
    // c4 of the four keys 4 kg .. 4 kg + 3.  The NEXT tile's loads are issued as soon as this tile's rows are in LDS.
    const int kg = tid >> 4, c4 = tid & 15;
    float4 rk[2][2], rp[3][2], rv[4];
    auto tile_load = [&](int a0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + 256 * j, r = e >> 3, c8 = e & 7;
            rk[j][0] = make_float4(0.f, 0.f, 0.f, 0.f); rk[j][1] = rk[j][0];
            if (a0 + r < amax) { const float* p_ = kbase + (long long)(a0 + r) * RNNT_D + 8 * c8; rk[j][0] = ldg4(p_); rk[j][1] = ldg4(p_ + 4); }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e = tid + 256 * j, r = e >> 3, c8 = e & 7;
            const int pr = a0 + pmin + r;
            rp[j][0] = make_float4(0.f, 0.f, 0.f, 0.f); rp[j][1] = rp[j][0];
            if (e < 640 && pr >= 0 && pr < RNNT_PE_LEN) { const float* p_ = pbase + (long long)pr * RNNT_D + 8 * c8; rp[j][0] = ldg4(p_); rp[j][1] = ldg4(p_ + 4); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a0 + 4 * kg + j < amax) rv[j] = ldg4(vbase + (long long)(a0 + 4 * kg + j) * RNNT_D + 4 * c4);
        }
    };
    auto put4 = [&](unsigned char* hi, unsigned char* lo, int off, float x0, float x1, float x2, float x3) {   // four values -> 8 bytes per plane
        float r0, r1, r2, r3, d0, d1;
        uint2 hv, lv;
        hv.x = pack2_16<F16>(x0, x1, r0, r1);
        hv.y = pack2_16<F16>(x2, x3, r2, r3);
        *reinterpret_cast<uint2*>(hi + off) = hv;
        if constexpr (LO) {
            lv.x = pack2_16<F16>(r0, r1, d0, d1);
            lv.y = pack2_16<F16>(r2, r3, d0, d1);
            *reinterpret_cast<uint2*>(lo + off) = lv;
        }
    };
    tile_load(amin);
    for (int a0 = amin; a0 < amax; a0 += 64) {
        __syncthreads();                                            // the previous tile's PV is done with the V image / Sx / al
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + 256 * j, r = e >> 3, c8 = e & 7;
            uint4 hv, lv;
            split8_16<F16, LO>(rk[j][0], rk[j][1], hv, lv);
            *reinterpret_cast<uint4*>(Kh + lmb_off(r, c8)) = hv;
            if constexpr (LO) *reinterpret_cast<uint4*>(Kl + lmb_off(r, c8)) = lv;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e = tid + 256 * j, r = e >> 3, c8 = e & 7;
            if (e < 640) {
                uint4 hv, lv;
                split8_16<F16, LO>(rp[j][0], rp[j][1], hv, lv);
                *reinterpret_cast<uint4*>(Ph + lmb_off(r, c8)) = hv;
                if constexpr (LO) *reinterpret_cast<uint4*>(Pl + lmb_off(r, c8)) = lv;
            }
        }
        {   // V transposed: row d = 4 c4 + dd, keys 4 kg .. 4 kg + 3 = half (kg & 1) of 16-byte chunk kg >> 1
            const int vo = (kg & 1) * 8;
            put4(Vh, Vl, lmb_off(4 * c4 + 0, kg >> 1) + vo, rv[0].x, rv[1].x, rv[2].x, rv[3].x);
            put4(Vh, Vl, lmb_off(4 * c4 + 1, kg >> 1) + vo, rv[0].y, rv[1].y, rv[2].y, rv[3].y);
            put4(Vh, Vl, lmb_off(4 * c4 + 2, kg >> 1) + vo, rv[0].z, rv[1].z, rv[2].z, rv[3].z);
            put4(Vh, Vl, lmb_off(4 * c4 + 3, kg >> 1) + vo, rv[0].w, rv[1].w, rv[2].w, rv[3].w);
        }
        __syncthreads();
        if (a0 + 64 < amax) tile_load(a0 + 64);
        // ---- A: matrix_ac tiles and G tiles of this wave's query tile ------------------------------------------------------------
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kt = 2 * hf + kk;
            f32x4_ acc = (f32x4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int off = lmb_off(16 * kt + i, kq + 4 * s);
                const uint4 kh = *reinterpret_cast<const uint4*>(Kh + off);
                if constexpr (LO) {
                    const uint4 kl = *reinterpret_cast<const uint4*>(Kl + off);
                    acc = mfma16_<F16>(kh, qul[s], acc);
                    acc = mfma16_<F16>(kl, quh[s], acc);
                }
                acc = mfma16_<F16>(kh, quh[s], acc);
            }
            // operands swapped (bitwise the same sums): the lane holds query row 16 qt + i and keys 16 kt + 4 kq + 0..3 -> one 16-byte write
            *reinterpret_cast<f32x4_*>(&Sx[(16 * qt + i) * LM2_LD + 16 * kt + 4 * kq]) = acc;
        }
        for (int gt = (hf ? 3 : 0); gt < (hf ? 5 : 3); ++gt) {
            f32x4_ acc = (f32x4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int off = lmb_off(16 * gt + i, kq + 4 * s);
                const uint4 ph = *reinterpret_cast<const uint4*>(Ph + off);
                if constexpr (LO) {
                    const uint4 pl = *reinterpret_cast<const uint4*>(Pl + off);
                    acc = mfma16_<F16>(ph, qvl[s], acc);
                    acc = mfma16_<F16>(pl, qvh[s], acc);
                }
                acc = mfma16_<F16>(ph, qvh[s], acc);
            }
            *reinterpret_cast<f32x4_*>(&Gx[(16 * qt + i) * LM2_GLD + 16 * gt + 4 * kq]) = acc;
        }
        __syncthreads();
        // ---- B: softmax of query slots 8*wave .. +7, lane = key (f32, as in rel_attention_lm_mfma) -----------------------------------
        {
            float sc[8], mx[8], pe_[8], sm[8];
            bool live[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) live[j] = a0 < s_ke[j] && a0 + 64 > s_ks[j];   // wave-uniform
            const int a = a0 + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int qs = 8 * wave + j;
                const bool valid = live[j] && a >= s_ks[j] && a < s_ke[j];
                sc[j] = valid ? (Sx[qs * LM2_LD + lane] + Gx[qs * LM2_GLD + lane + s_prel[j]]) * 0.125f : -INFINITY;
                mx[j] = sc[j];
            }
            // all-reduce over the 64 keys, ascending butterfly (DPP partners for 1..8: rnnt_common.hip.h), the 8 rows interleaved
#define LMB_RED(O_, EXPR_) _Pragma("unroll") for (int j = 0; j < 8; ++j) { const float o_ = xor_partner<O_>(RV_[j]); RV_[j] = EXPR_; }
#define RV_ mx
            LMB_RED(1, fmaxf(RV_[j], o_)) LMB_RED(2, fmaxf(RV_[j], o_)) LMB_RED(4, fmaxf(RV_[j], o_)) LMB_RED(8, fmaxf(RV_[j], o_)) LMB_RED(16, fmaxf(RV_[j], o_)) LMB_RED(32, fmaxf(RV_[j], o_))
#undef RV_
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float mnew = fmaxf(mrun[j], mx[j]);
                pe_[j] = sc[j] > -INFINITY ? __builtin_amdgcn_exp2f((sc[j] - mnew) * 1.4426950408889634f) : 0.f;   // v_exp_f32 (1 ulp); the exact-f32 kernel keeps expf
                sm[j] = pe_[j];
                mx[j] = live[j] ? __builtin_amdgcn_exp2f((mrun[j] - mnew) * 1.4426950408889634f) : 1.0f;   // alpha (first live tile: exp2(-inf) = 0)
                if (live[j]) mrun[j] = mnew;
            }
#define RV_ sm
            LMB_RED(1, RV_[j] + o_) LMB_RED(2, RV_[j] + o_) LMB_RED(4, RV_[j] + o_) LMB_RED(8, RV_[j] + o_) LMB_RED(16, RV_[j] + o_) LMB_RED(32, RV_[j] + o_)
#undef RV_
#undef LMB_RED
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int qs = 8 * wave + j;
                lrun[j] = lrun[j] * mx[j] + sm[j];
                Sx[qs * LM2_LD + lane] = pe_[j];
                if (lane == 0) al[qs] = mx[j];
            }
        }
        __syncthreads();
        // ---- C: O = alpha O + P V for d tiles 2*hf, 2*hf + 1; the probabilities of row 16 qt + i, keys 32 s + 8 kq .. + 8, split here ----
        {
            const float a_ = al[16 * qt + i];                      // (transposed accumulators: one query row per lane)
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[0][r] *= a_; o[1][r] *= a_; }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float* pp = &Sx[(16 * qt + i) * LM2_LD + 32 * s + 8 * kq];
            uint4 ph, pl;
            split8_16<F16, LO>(*reinterpret_cast<const float4*>(pp), *reinterpret_cast<const float4*>(pp + 4), ph, pl);
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) {
                const int off = lmb_off(16 * (2 * hf + dd) + i, kq + 4 * s);
                const uint4 vh = *reinterpret_cast<const uint4*>(Vh + off);
                if constexpr (LO) {
                    const uint4 vl = *reinterpret_cast<const uint4*>(Vl + off);
                    o[dd] = mfma16_<F16>(vh, pl, o[dd]);
                    o[dd] = mfma16_<F16>(vl, ph, o[dd]);
                }
                o[dd] = mfma16_<F16>(vh, ph, o[dd]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (lane == 0) al[8 * wave + j] = lrun[j] > 0.f ? 1.0f / lrun[j] : 0.f;
    __syncthreads();
    if (my_f >= 0) {                                                // query row 16 qt + i: d = 16 (2 hf + dd) + 4 kq + 0..3 -> 16-byte stores
        const long long m = (long long)b * P.F + my_f;
        const float li = al[16 * qt + i];
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
            stg4(P.out + m * RNNT_D + h * RNNT_DK + 16 * (2 * hf + dd) + 4 * kq, make_float4(o[dd][0] * li, o[dd][1] * li, o[dd][2] * li, o[dd][3] * li));
    }
}
