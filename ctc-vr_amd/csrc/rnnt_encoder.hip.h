// Encoder kernels other than the GEMMs: conv1, LayerNorm, the two attention kernels, depthwise conv, ring initialisation.
// Part of rnnt_kernels.hip.h (include that umbrella, not this file).
#pragma once

// ------------------------------------------------------------------------------------------------
// conv1_relu: y1[b][t][f][c] = relu(b1[c] + sum_{kh,kw} x[b][2t+kh][2f+kw] * w1[c][kh][kw])
// (Conv2d(1,256,3,2)+ReLU, wenet/transformer/subsampling.py:189-190).  Channels-last so that the
// conv2 implicit GEMM reads 768 contiguous floats per kernel row.  One thread per (b,t,f,c).
// ------------------------------------------------------------------------------------------------
// Virtual streams: v = c*B + b reads fbank[b][starts[c] .. ) (starts == null -> c = 0, start 0): the
// wavefront path subsamples several equal-length chunks of every stream in one launch.
__global__ void conv1_relu(const float* __restrict__ x, const float* __restrict__ w1t /*[9][256]*/,
                           const float* __restrict__ b1, float* __restrict__ y1, int B, int T, int t1,
                           const int* __restrict__ starts, int n_chunks, int b_major) {
    const long long n = (long long)n_chunks * B * t1 * RNNT_F1 * RNNT_D;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id & 255);
        long long r = id >> 8;
        const int f = (int)(r % RNNT_F1);
        r /= RNNT_F1;
        const int t = (int)(r % t1);
        const int v = (int)(r / t1);
        // virtual stream order: chunk-major (v = c*B + b) or stream-major (v = b*n_chunks + c: a stream's chunks are consecutive,
        // so its frames come out of the embed Linear in time order -- the layer-major encoder's row layout)
        const int cidx = b_major ? v % n_chunks : v / B, b = b_major ? v / n_chunks : v - cidx * B;
        const int st0 = starts ? starts[cidx] : 0;
        const float* xp = x + ((long long)b * T + st0 + 2 * t) * RNNT_IDIM + 2 * f;
        float acc = b1[c];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc = fmaf(xp[kh * RNNT_IDIM + kw], w1t[(kh * 3 + kw) * RNNT_D + c], acc);
        if (n_chunks > 1) stg1_nt(y1 + id, fmaxf(acc, 0.f));   // whole slab (~180 MB): written once, read once by conv2: keep it out of L2
        else y1[id] = fmaxf(acc, 0.f);
    }
}

// conv1_relu_rows: the same convolution with one workgroup per (virtual stream, block of C1_TB output rows) and one thread per
// channel: the 2*C1_TB+1 input rows sit in LDS (every tap is a broadcast read), the nine weights of the thread's channel in
// registers, and each output row of 39 x 256 floats leaves as 39 fully coalesced 1 KiB stores.  conv1_relu above spends 18
// vector loads per 4-byte store (0.9-1.2 TB/s); this one is bound by the 1.1 GB of output it writes.
#define C1_TB 8
__global__ __launch_bounds__(256) void conv1_relu_rows(const float* __restrict__ x, const float* __restrict__ w1t /*[9][256]*/,
                                                       const float* __restrict__ b1, float* __restrict__ y1, int B, int T, int t1,
                                                       const int* __restrict__ starts, int n_chunks, int b_major) {
    __shared__ float xs[(2 * C1_TB + 1) * RNNT_IDIM];
    const int v = blockIdx.x, tb = blockIdx.y * C1_TB;
    const int cidx = b_major ? v % n_chunks : v / B, b = b_major ? v / n_chunks : v - cidx * B;
    const int st0 = starts ? starts[cidx] : 0;
    const int c = threadIdx.x;
    const int nt = min(C1_TB, t1 - tb), nrows = 2 * nt + 1;
    const float* xp = x + ((long long)b * T + st0 + 2 * tb) * RNNT_IDIM;
    for (int e = threadIdx.x; e < nrows * RNNT_IDIM; e += 256) xs[e] = ldg1(xp + e);
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = ldg1(w1t + k * RNNT_D + c);
    const float bias = ldg1(b1 + c);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const float* r0 = xs + 2 * t * RNNT_IDIM;
        float* yp = y1 + (((long long)v * t1 + tb + t) * RNNT_F1) * RNNT_D + c;
#pragma unroll 3
        for (int f = 0; f < RNNT_F1; ++f) {
            float acc = bias;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc = fmaf(r0[kh * RNNT_IDIM + 2 * f + kw], w[kh * 3 + kw], acc);
            stg1_nt(yp + (long long)f * RNNT_D, fmaxf(acc, 0.f));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// layer_norm: y[row] = LN(x[row]) over 256 columns, one wave per row, output row map like gemm C.
// ------------------------------------------------------------------------------------------------
struct LnP {
    const float* x;
    const float* g;
    const float* b;
    float* y;
    int M, c_n, c_r0;
    long long c_s0, c_s1;
};
__device__ __forceinline__ void layer_norm_body(const LnP& p) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= p.M) return;
    const float4 v = ldg4(p.x + (long long)row * RNNT_D + lane * 4);
    const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.0f);
    const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
    const float rstd = 1.0f / sqrtf(wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / 256.0f) + 1e-5f);
    const float4 gg = ldg4(p.g + lane * 4);
    const float4 bb = ldg4(p.b + lane * 4);
    float4 o;
    o.x = dx * rstd * gg.x + bb.x;
    o.y = dy * rstd * gg.y + bb.y;
    o.z = dz * rstd * gg.z + bb.z;
    o.w = dw * rstd * gg.w + bb.w;
    long long off;
    if (p.c_s0 == 0) off = (long long)(row + p.c_r0) * p.c_s1;   // plain rows
    else off = (long long)(row / p.c_n) * p.c_s0 + (long long)((row % p.c_n) + p.c_r0) * p.c_s1;
    stg4(p.y + off + lane * 4, o);
}
__global__ void layer_norm(LnP p) { layer_norm_body(p); }
__global__ void layer_norm_tab(const LnP* __restrict__ tab) {
    const LnP p = tab[blockIdx.z];
    layer_norm_body(p);
}

// ------------------------------------------------------------------------------------------------
// rel_attention: RelPositionMultiHeadedAttention score/softmax/PV (attention.py:400-418,170-177)
// for streaming chunks and full context.  grid = (B*H, ceil(tq/16)), block = 256 (4 waves).
//   q      [B*tq, 256]            query projections (bias included)
//   kc, vc [B][kv_stride rows][256] K / V caches; keys j = 0..T2-1 live at rows kv_start + j
//   ptab   [5000][256]            pe * W_pos^T for this layer; key j uses row pos_start + j
//   klen   per-stream number of valid keys (null -> T2 for all; full-context padding mask)
// (ATT_QB below is the largest query block, NQ = 4.)
// score(i,j) = ((q_i+u).k_j + (q_i+v).p_j) / 8, softmax over j, out_i = sum_j a_ij v_j.
// Per 64-key tile: K/P/V rows staged in LDS with coalesced float4 loads; scores with lane = key;
// online softmax per query row (wave w owns queries w, w+4, w+8, w+12); PV with lane = d.
// ------------------------------------------------------------------------------------------------
#define ATT_QB 16
#define ATT_TK 64
#define ATT_LD 68
struct AttnP {
    const float* q;
    const float* kc;
    const float* vc;
    const float* ptab;
    const float* bias_u;
    const float* bias_v;
    const int* klen;
    float* out;
    int tq, T2, kv_start, pos_start;
    long long kv_stride;
};
// NQ = query slots per wave: a workgroup covers 4*NQ queries (wave w owns queries w, w+4, ...).  Streaming chunks
// have t' = 3..5 new frames, so NQ = 1 or 2 avoids computing 16 query slots for 3 queries.
template <int NQ>
__device__ __forceinline__ void rel_attention_body(const AttnP& P) {
    const float* __restrict__ q = P.q;
    const float* __restrict__ kc = P.kc;
    const float* __restrict__ vc = P.vc;
    const float* __restrict__ ptab = P.ptab;
    const float* __restrict__ bias_u = P.bias_u;
    const float* __restrict__ bias_v = P.bias_v;
    const int* __restrict__ klen = P.klen;
    float* __restrict__ out = P.out;
    const int tq = P.tq, T2 = P.T2, kv_start = P.kv_start, pos_start = P.pos_start;
    const long long kv_stride = P.kv_stride;
    constexpr int QB = 4 * NQ;
    if ((int)blockIdx.y * QB >= tq) return;
    __shared__ __attribute__((aligned(16))) float Ks[ATT_TK * ATT_LD];
    __shared__ __attribute__((aligned(16))) float Ps[ATT_TK * ATT_LD];
    __shared__ __attribute__((aligned(16))) float Vs[ATT_TK * RNNT_DK];
    __shared__ __attribute__((aligned(16))) float Qu[QB * RNNT_DK];
    __shared__ __attribute__((aligned(16))) float Qv[QB * RNNT_DK];
    __shared__ float Pm[QB * ATT_TK];
    const int b = blockIdx.x / RNNT_H, h = blockIdx.x % RNNT_H;
    const int q0 = blockIdx.y * QB;
    const int nq = min(QB, tq - q0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nk = klen ? min(ldgi(klen + b), T2) : T2;

    for (int e = tid; e < QB * RNNT_DK; e += 256) {
        const int iq = e >> 6, d = e & 63;
        float qq = 0.f;
        if (iq < nq) qq = ldg1(q + ((long long)b * tq + q0 + iq) * RNNT_D + h * RNNT_DK + d);
        Qu[e] = qq + ldg1(bias_u + h * RNNT_DK + d);
        Qv[e] = qq + ldg1(bias_v + h * RNNT_DK + d);
    }
    float mrun[NQ], lrun[NQ], o[NQ];
#pragma unroll
    for (int s = 0; s < NQ; ++s) {
        mrun[s] = -INFINITY;
        lrun[s] = 0.f;
        o[s] = 0.f;
    }
    const float* kbase = kc + ((long long)b * kv_stride + kv_start) * RNNT_D + h * RNNT_DK;
    const float* vbase = vc + ((long long)b * kv_stride + kv_start) * RNNT_D + h * RNNT_DK;
    const float* pbase = ptab + (long long)pos_start * RNNT_D + h * RNNT_DK;

    for (int j0 = 0; j0 < nk; j0 += ATT_TK) {
        __syncthreads();
#pragma unroll
        for (int mIt = 0; mIt < 4; ++mIt) {
            const int r = (tid >> 4) + 16 * mIt, c4 = tid & 15;
            const int j = j0 + r;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), pv = kv, vv = kv;
            if (j < nk) {
                kv = ldg4(kbase + (long long)j * RNNT_D + c4 * 4);
                pv = ldg4(pbase + (long long)j * RNNT_D + c4 * 4);
                vv = ldg4(vbase + (long long)j * RNNT_D + c4 * 4);
            }
            *reinterpret_cast<float4*>(&Ks[r * ATT_LD + c4 * 4]) = kv;
            *reinterpret_cast<float4*>(&Ps[r * ATT_LD + c4 * 4]) = pv;
            *reinterpret_cast<float4*>(&Vs[r * RNNT_DK + c4 * 4]) = vv;
        }
        __syncthreads();
        // scores: lane = key
        float s[NQ];
#pragma unroll
        for (int sI = 0; sI < NQ; ++sI) s[sI] = 0.f;
#pragma unroll 4
        for (int dc = 0; dc < 16; ++dc) {
            const float4 k4 = *reinterpret_cast<const float4*>(&Ks[lane * ATT_LD + dc * 4]);
            const float4 p4 = *reinterpret_cast<const float4*>(&Ps[lane * ATT_LD + dc * 4]);
#pragma unroll
            for (int sI = 0; sI < NQ; ++sI) {
                const int iq = wave + 4 * sI;
                const float4 u4 = *reinterpret_cast<const float4*>(&Qu[iq * RNNT_DK + dc * 4]);
                const float4 v4 = *reinterpret_cast<const float4*>(&Qv[iq * RNNT_DK + dc * 4]);
                float t = s[sI];
                t = fmaf(u4.x, k4.x, t);
                t = fmaf(u4.y, k4.y, t);
                t = fmaf(u4.z, k4.z, t);
                t = fmaf(u4.w, k4.w, t);
                t = fmaf(v4.x, p4.x, t);
                t = fmaf(v4.y, p4.y, t);
                t = fmaf(v4.z, p4.z, t);
                t = fmaf(v4.w, p4.w, t);
                s[sI] = t;
            }
        }
        const bool valid = (j0 + lane) < nk;
        float alpha[NQ];
#pragma unroll
        for (int sI = 0; sI < NQ; ++sI) {
            const float sc = valid ? s[sI] * 0.125f : -INFINITY;
            const float mnew = fmaxf(mrun[sI], wave_max(sc));
            const float pe_ = valid ? expf(sc - mnew) : 0.f;
            alpha[sI] = expf(mrun[sI] - mnew);   // first tile: exp(-inf) = 0
            lrun[sI] = lrun[sI] * alpha[sI] + wave_sum(pe_);
            mrun[sI] = mnew;
            Pm[(wave + 4 * sI) * ATT_TK + lane] = pe_;
        }
        __syncthreads();   // Pm visible (uniform trip count: nk is the same for the whole workgroup)
        // PV: lane = d
#pragma unroll
        for (int sI = 0; sI < NQ; ++sI) o[sI] *= alpha[sI];
        const int jn = min(ATT_TK, nk - j0);
        for (int j = 0; j < jn; ++j) {
            const float vj = Vs[j * RNNT_DK + lane];
#pragma unroll
            for (int sI = 0; sI < NQ; ++sI) o[sI] = fmaf(Pm[(wave + 4 * sI) * ATT_TK + j], vj, o[sI]);
        }
    }
#pragma unroll
    for (int sI = 0; sI < NQ; ++sI) {
        const int iq = wave + 4 * sI;
        if (iq < nq) stg1(out + ((long long)b * tq + q0 + iq) * RNNT_D + h * RNNT_DK + lane, o[sI] / lrun[sI]);
    }
}
template <int NQ>
__global__ __launch_bounds__(256) void rel_attention(AttnP p) { rel_attention_body<NQ>(p); }
template <int NQ>
__global__ __launch_bounds__(256) void rel_attention_tab(const AttnP* __restrict__ tab) {
    const AttnP p = tab[blockIdx.z];
    rel_attention_body<NQ>(p);
}

// ------------------------------------------------------------------------------------------------
// rel_attention_stream: the same attention for a STREAMING chunk (tq <= 4 new frames against a long cache).  With so
// few queries there is nothing to reuse a staged K tile for, and the kernel is a pure stream over the cache
// (K, V: 512 B per key and head from HBM / Infinity Cache; the positional rows come from L2).  So nothing is staged:
//   scores   16 lanes per key read the key's 64-float K row and P row as one float4 each (256 contiguous bytes per
//            row and instruction), multiply against the 4 queries' (q+u), (q+v) slices held in registers, and reduce
//            over the 16 lanes; 4 keys per lane group are in flight per iteration (8 independent 16-B loads per lane);
//   softmax  wave w owns query w: max / exp / sum over the score row in LDS;
//   PV       16 lanes per key again (float4 of V per lane), per-group partial sums, one LDS reduction over 16 groups.
// One workgroup per (stream, head); LDS = 4 score rows + 16 KB of partial sums, so 8 workgroups fit a CU and their
// phases interleave.  Dynamic LDS: (4 * t2cap + 16 * 4 * 64) floats.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rel_attention_stream_body(const AttnP& P, float* smem, int t2cap, int bh) {
    const float* __restrict__ kc = P.kc;
    const float* __restrict__ vc = P.vc;
    const int tq = P.tq, T2 = P.T2;
    float* S = smem;                       // [4][t2cap] scores, then probabilities
    float* red = smem + 4 * t2cap;         // [16 groups][4 queries][64]
    __shared__ float linv[4];
    const int b = bh / RNNT_H, h = bh % RNNT_H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = tid >> 4, l16 = tid & 15;
    const int nk = P.klen ? min(ldgi(P.klen + b), T2) : T2;
    const float* kbase = kc + ((long long)b * P.kv_stride + P.kv_start) * RNNT_D + h * RNNT_DK + 4 * l16;
    const float* vbase = vc + ((long long)b * P.kv_stride + P.kv_start) * RNNT_D + h * RNNT_DK + 4 * l16;
    const float* pbase = P.ptab + (long long)P.pos_start * RNNT_D + h * RNNT_DK + 4 * l16;
    // The first K/P rows travel together with the query rows.  Register double buffering of K/P/V and V rows fetched
    // across the softmax were measured slower: they cost the fifth wave per SIMD (> 96 VGPRs).
    float4 ka[4], pa[4];
#define ATS_LOAD(kk_, pp_, j0_)                                                                       \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                   \
        const int j_ = min((j0_) + g + 16 * u, nk - 1);                                               \
        kk_[u] = ldg4_nt(kbase + (long long)j_ * RNNT_D);                                             \
        pp_[u] = ldg4(pbase + (long long)j_ * RNNT_D);                                                \
    }
#define ATS_SCORE(kk_, pp_, j0_)                                                                      \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                   \
        const int j_ = (j0_) + g + 16 * u;                                                            \
        _Pragma("unroll") for (int iq = 0; iq < 4; ++iq) {                                            \
            float t = 0.f;                                                                            \
            t = fmaf(qu[iq].x, kk_[u].x, t);                                                          \
            t = fmaf(qu[iq].y, kk_[u].y, t);                                                          \
            t = fmaf(qu[iq].z, kk_[u].z, t);                                                          \
            t = fmaf(qu[iq].w, kk_[u].w, t);                                                          \
            t = fmaf(qv[iq].x, pp_[u].x, t);                                                          \
            t = fmaf(qv[iq].y, pp_[u].y, t);                                                          \
            t = fmaf(qv[iq].z, pp_[u].z, t);                                                          \
            t = fmaf(qv[iq].w, pp_[u].w, t);                                                          \
            _Pragma("unroll") for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 16);              \
            if (l16 == iq && j_ < nk) S[iq * t2cap + j_] = t * 0.125f;                                \
        }                                                                                             \
    }
    ATS_LOAD(ka, pa, 0)
    // this lane's 4-float slice of every query, with the two biases
    float4 qu[4], qv[4];
    {
        const float4 bu = ldg4(P.bias_u + h * RNNT_DK + 4 * l16), bv = ldg4(P.bias_v + h * RNNT_DK + 4 * l16);
#pragma unroll
        for (int iq = 0; iq < 4; ++iq) {
            float4 qq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iq < tq) qq = ldg4(P.q + ((long long)b * tq + iq) * RNNT_D + h * RNNT_DK + 4 * l16);
            qu[iq] = make_float4(qq.x + bu.x, qq.y + bu.y, qq.z + bu.z, qq.w + bu.w);
            qv[iq] = make_float4(qq.x + bv.x, qq.y + bv.y, qq.z + bv.z, qq.w + bv.w);
        }
    }
    // ---- scores --------------------------------------------------------------------------------------------------------
    for (int j0 = 0; j0 < nk; j0 += 64) {
        if (j0 > 0) ATS_LOAD(ka, pa, j0)
        ATS_SCORE(ka, pa, j0)
    }
#undef ATS_LOAD
#undef ATS_SCORE
    __syncthreads();
    // ---- softmax: wave w = query w ---------------------------------------------------------------------------------------
    {
        float* row = S + wave * t2cap;
        float m = -INFINITY;
        for (int j = lane; j < nk; j += 64) m = fmaxf(m, row[j]);
        m = wave_max(m);
        float sum = 0.f;
        for (int j = lane; j < nk; j += 64) {
            const float e = expf(row[j] - m);
            row[j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        if (lane == 0) linv[wave] = 1.0f / sum;
    }
    __syncthreads();
    // ---- PV: group g takes keys g, g+16, ... -------------------------------------------------------------------------------
    float4 acc[4];
#pragma unroll
    for (int iq = 0; iq < 4; ++iq) acc[iq] = make_float4(0.f, 0.f, 0.f, 0.f);
#define ATS_PV(vv_, j0_)                                                                              \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                   \
        const int j_ = (j0_) + g + 16 * u;                                                            \
        if (j_ < nk) {                                                                                \
            _Pragma("unroll") for (int iq = 0; iq < 4; ++iq) {                                        \
                const float pj = S[iq * t2cap + j_];                                                  \
                acc[iq].x = fmaf(pj, vv_[u].x, acc[iq].x);                                            \
                acc[iq].y = fmaf(pj, vv_[u].y, acc[iq].y);                                            \
                acc[iq].z = fmaf(pj, vv_[u].z, acc[iq].z);                                            \
                acc[iq].w = fmaf(pj, vv_[u].w, acc[iq].w);                                            \
            }                                                                                         \
        }                                                                                             \
    }
    for (int j0 = 0; j0 < nk; j0 += 64) {
        float4 va[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) va[u] = ldg4_nt(vbase + (long long)min(j0 + g + 16 * u, nk - 1) * RNNT_D);
        ATS_PV(va, j0)
    }
#undef ATS_PV
#pragma unroll
    for (int iq = 0; iq < 4; ++iq) *reinterpret_cast<float4*>(&red[(g * 4 + iq) * RNNT_DK + 4 * l16]) = acc[iq];
    __syncthreads();
    {   // thread = (query, d): sum the 16 groups in a fixed order
        const int iq = tid >> 6, d = tid & 63;
        float o = 0.f;
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) o += red[(gg * 4 + iq) * RNNT_DK + d];
        if (iq < tq) stg1(P.out + ((long long)b * tq + iq) * RNNT_D + h * RNNT_DK + d, o * linv[iq]);
    }
}
__global__ __launch_bounds__(256) void rel_attention_stream(AttnP p, int t2cap) {
    extern __shared__ __attribute__((aligned(16))) float att_smem[];
    rel_attention_stream_body(p, att_smem, t2cap, blockIdx.x);
}
// Table form, 1-D grid of ceil(bh_total / 8) * 8 * n_desc workgroups dealt round-robin over the 8 XCDs: XCD x takes the
// (stream, head) pairs x, x + 8, ... and runs ALL descriptors of a pair back to back.  With two chunks of a layer per
// stage the second chunk reads the K/V rows the first one has just pulled into that XCD's L2 (its own three rows more),
// so the cache is streamed from HBM once per stage instead of once per chunk.  Placement is a speed hint only.
__global__ __launch_bounds__(256) void rel_attention_stream_tab(const AttnP* __restrict__ tab, int t2cap, int n_desc, int bh_total, int pair_major) {
    extern __shared__ __attribute__((aligned(16))) float att_smem[];
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int per = (bh_total + 7) / 8;                        // (stream, head) pairs per XCD
    const int d = pair_major ? slot % n_desc : slot / per;
    const int bh = (pair_major ? slot / n_desc : slot % per) * 8 + xcd;
    if (bh >= bh_total || d >= n_desc) return;
    const AttnP p = tab[d];
    rel_attention_stream_body(p, att_smem, t2cap, bh);
}

// ------------------------------------------------------------------------------------------------
// dwconv_bn_silu: causal depthwise conv k=31 + BatchNorm(eval) + SiLU over the post-GLU ring
// (convolution.py:142-145).  Padded frames of a full-context batch are not masked: the conv is causal and
// padded keys are masked in attention, so they can never reach a valid frame.  ring g [B][cap][256]; frame (pos+r) of stream b lives at row
// (pos + r + cap*K) % cap; the 30 frames before pos are the left context.  Also records the
// pre-LayerNorm conv-module input rows into the xin ring (for the reference's cnn_cache view).
//   out[m][c] = silu((bdw[c] + sum_k wdw[k][c] * g[frame pos+r-30+k][c]) * bn_s[c] + bn_t[c])
// ------------------------------------------------------------------------------------------------
struct DwP {
    const float* g;
    const float* wdw_t;
    const float* bdw;
    const float* bn_s;
    const float* bn_t;
    float* out;
    const float* xres;
    float* xring;
    int B, tq, cap, pos;
};
__device__ __forceinline__ void dwconv_body(const DwP& P) {
    const float* __restrict__ g = P.g;
    const float* __restrict__ wdw_t = P.wdw_t;
    const float* __restrict__ bdw = P.bdw;
    const float* __restrict__ bn_s = P.bn_s;
    const float* __restrict__ bn_t = P.bn_t;
    float* __restrict__ out = P.out;
    const float* __restrict__ xres = P.xres;
    float* __restrict__ xring = P.xring;
    const int B = P.B, tq = P.tq, cap = P.cap, pos = P.pos;
    const long long n = (long long)B * tq * RNNT_D;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id & 255);
        const int m = (int)(id >> 8);
        const int b = m / tq, r = m % tq;
        const float* gb = g + (long long)b * cap * RNNT_D + c;
        float acc = ldg1(bdw + c);
        int ridx = (pos + r - RNNT_LORDER + cap * 64) % cap;   // ring row of the oldest tap (operand kept positive)
#pragma unroll
        for (int k = 0; k < RNNT_KDW; ++k) {
            acc = fmaf(ldg1(wdw_t + k * RNNT_D + c), ldg1(gb + (long long)ridx * RNNT_D), acc);
            ridx = ridx + 1 == cap ? 0 : ridx + 1;
        }
        float v = acc * ldg1(bn_s + c) + ldg1(bn_t + c);
        v = v * sigmoidf_(v);
        stg1(out + id, v);
        if (xring) stg1(xring + ((long long)b * cap + (pos + r) % cap) * RNNT_D + c, ldg1(xres + id));
    }
}
__global__ void dwconv_bn_silu(DwP p) { dwconv_body(p); }
__global__ void dwconv_bn_silu_tab(const DwP* __restrict__ tab) {
    const DwP p = tab[blockIdx.z];
    dwconv_body(p);
}

// fill the 30 left-context rows of a fresh stream: g ring <- GLU(b_pw1) (zero input through the
// biased pointwise conv, convolution.py:122-124,138-139), xin ring <- 0.
__global__ void conv_ring_init(float* __restrict__ g, float* __restrict__ xring, const float* __restrict__ glu0 /*[L][256]*/,
                               int B, int cap) {
    const long long n = (long long)RNNT_L * B * cap * RNNT_D;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id & 255);
        const int l = (int)(id / ((long long)B * cap * RNNT_D));
        g[id] = glu0[l * RNNT_D + c];
        xring[id] = 0.f;
    }
}
