// Device kernels of the gfx950 streaming RNN-T path.  fp32 storage, exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, see MI355X guide §3) so that greedy
// tokens can match the reference's float32 CPU path.
//
// Kernel inventory (DESIGN.md §4 has the roofline per kernel):
//   gemm32<WK>        small-M "NT" GEMM, one 32x32 output tile per workgroup, K split over WK waves,
//                     deterministic LDS reduction, fused LayerNorm prologue and bias / SiLU / scale /
//                     residual / GLU / LSTM-cell / joint-tanh epilogues, generalised A/C addressing
//                     (implicit-GEMM conv2, K/V-cache append, ring buffers).
//   conv1_relu        Conv2d(1->256,k3,s2)+ReLU, channels-last output.
//   rel_attention     streaming rel-pos attention over the K/V cache with LDS-staged K/P/V tiles and
//                     online softmax (one workgroup per stream x head x 16-query block).
//   dwconv_bn_silu    causal depthwise k=31 over the post-GLU ring + folded BatchNorm + SiLU.
//   layer_norm        row LayerNorm (norm_final / after_norm).
//   greedy_update     argmax over the vocabulary + per-stream RNN-T greedy state machine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RNNT_D 256
#define RNNT_H 4
#define RNNT_DK 64
#define RNNT_FF 1024
#define RNNT_L 12
#define RNNT_LORDER 30
#define RNNT_KDW 31
#define RNNT_IDIM 80
#define RNNT_FSUB 19
#define RNNT_F1 39
#define RNNT_PE_LEN 5000

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum {
    EPI_BIAS = 0,      // C = acc + bias
    EPI_SILU = 1,      // C = silu(acc + bias)
    EPI_RELU = 2,      // C = relu(acc + bias)
    EPI_SCALE = 3,     // C = (acc + bias) * alpha
    EPI_RESID = 4,     // C = R + alpha * (acc + bias)          (R may alias C)
    EPI_GLU = 5,       // interleaved (a,gate) columns -> C[:, n/2] = a * sigmoid(gate)
    EPI_LSTM = 6,      // interleaved (i,f,g,o) columns + input table row -> h', c'
    EPI_TANH_ADD = 7,  // C = tanh(acc + bias + X[gather(m)][n])   (joint: enc_proj[t_b] + pred_proj)
    EPI_ARGMAX = 8,    // no store: per-row argmax of (acc + bias) over all columns into key[m] (greedy decode)
    EPI_DB = 9         // C = 10 * log10(max(acc + bias, 1e-10))   (AmplitudeToDB, power spectrogram)
};

struct GemmP {
    const float* A;
    const float* W;
    const float* bias;   // [N] or null
    float* C;
    const float* R;      // residual (EPI_RESID)
    const float* ln_g;   // LayerNorm prologue over K (requires K == 256, plain A rows); null = off
    const float* ln_b;
    int M, N, K;
    // A row addressing: off(m) = (m / a_n1)*a_s0 + ((m % a_n1) / a_n2)*a_s1 + (m % a_n2)*a_s2
    // K segments:       off(k) = (k / a_seg)*a_seg_stride + (k % a_seg)
    int a_n1, a_n2;
    long long a_s0, a_s1, a_s2;
    int a_seg;
    long long a_seg_stride;
    int ldw;
    // C addressing: off(m,n) = (m / c_n)*c_s0 + (((m % c_n) + c_r0) % c_mod)*c_s1 + n
    int c_n, c_r0, c_mod;
    long long c_s0, c_s1;
    int epi;
    float alpha;
    // EPI_LSTM: X = input-gate table [vocab][4*D] interleaved, I = token per row, X2 = c_in [M][D],
    //           C = h_out [M][D], Y2 = c_out [M][D]
    // EPI_TANH_ADD: X = projected encoder frames, row(m) = (m / x_n)*x_s0 + I[m / x_n]*x_s1 (I null -> m % x_n)
    const float* X;
    const float* X2;
    float* Y2;
    const int* I;
    int x_n;
    long long x_s0, x_s1;
    // host-computed: addressing fast paths (no integer division in the kernel) and division magics
    // (q = umulhi(n, magic), exact while n*d < 2^32; see fastdiv()).
    int a_plain, c_plain;
    int lstm_ld;   // EPI_LSTM: row stride (floats) of X2 / C / Y2; 0 -> 256
    // greedy decode: per-row buffer select (LSTM state ping-pong) and fused argmax
    const int* Asel;             // [M] 0/1: A row m lives in buffer Asel[m] (^ asel_invert); null = off
    long long asel_stride;       // floats between the two buffers (also used by the EPI_LSTM state rows when Asel != null)
    int asel_invert;
    unsigned long long* key;     // EPI_ARGMAX: per-row packed (ordered value, ~index) maximum, atomicMax
    const int* nframes;          // EPI_ARGMAX: rows with I[m] >= *nframes are idle (no key written)
    int a_tanh;                  // gemm_ns A prologue: a = tanh(A[row(m)][k] + X[(m / x_n) * x_s0 + k]) (joint lattice, joint.py:60-66)
    const int* act_idx;          // greedy decode: row m is active iff act_idx[m] < *act_lim; a workgroup whose rows are
    const int* act_lim;          //   all idle exits at once (idle budgeted steps must cost nothing); null = always active
    int dbg;       // microbenchmark ablation bits (0 in production): 1 skip global loads, 2 skip MFMAs, 4 skip epilogue
    unsigned a_n1_magic, a_n2_magic, a_seg_magic, c_n_magic, x_n_magic;
    int a_n1_shift, a_n2_shift, a_seg_shift, c_n_shift, x_n_shift;   // q = umulhi(n, magic) >> shift, exact for n < 2^31
};

struct GemmBatch {
    GemmP g[3];
};

// Explicit global-address-space accesses.  Pointers that come out of an in-memory descriptor are "generic" to the
// compiler, which then emits flat_load/flat_store: those count on BOTH vmcnt and lgkmcnt, so every wait for an LDS
// read also drains the global loads in flight and nothing overlaps.  These helpers force global_load/global_store.
#if defined(__HIP_DEVICE_COMPILE__)
#define RNNT_GAS __attribute__((address_space(1)))
typedef float f32x4g __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4(const float* p) {
    const f32x4g v = *(const RNNT_GAS f32x4g*)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}
// read-once streams (K/V cache rows): non-temporal, so that they do not evict the weights other kernels keep in L2
__device__ __forceinline__ float4 ldg4_nt(const float* p) {
    const f32x4g v = __builtin_nontemporal_load((const RNNT_GAS f32x4g*)p);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float ldg1(const float* p) { return *(const RNNT_GAS float*)p; }
__device__ __forceinline__ int ldgi(const int* p) { return *(const RNNT_GAS int*)p; }
__device__ __forceinline__ void stg1(float* p, float v) { *(RNNT_GAS float*)p = v; }
__device__ __forceinline__ void stg1_nt(float* p, float v) { __builtin_nontemporal_store(v, (RNNT_GAS float*)p); }
__device__ __forceinline__ void stg4(float* p, float4 v) { *(RNNT_GAS f32x4g*)p = (f32x4g){v.x, v.y, v.z, v.w}; }
#else   // host pass of the single-source compile: never executed
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ldg4_nt(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float ldg1(const float* p) { return *p; }
__device__ __forceinline__ int ldgi(const int* p) { return *p; }
__device__ __forceinline__ void stg1(float* p, float v) { *p = v; }
__device__ __forceinline__ void stg1_nt(float* p, float v) { *p = v; }
__device__ __forceinline__ void stg4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
#endif

__device__ __forceinline__ int fastdiv(int n, int d, unsigned magic, int shift) {   // exact for 0 <= n < 2^31
    return d == 1 ? n : (int)(__umulhi((unsigned)n, magic) >> shift);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ------------------------------------------------------------------------------------------------
// gemm32<WK>: C[32x32 tile] = epi(A[M,K] * W[N,K]^T).  grid = (ceil(N/32), ceil(M/32), groups),
// block = 64*WK threads.  Wave w accumulates K-slice [w*K/WK, (w+1)*K/WK) with 32x32x2 f32 MFMAs:
// lane (i = l&31, kh = l>>5) feeds A[m0+i][k + 4*kh + e] and W[n0+i][k + 4*kh + e], e = 0..3, from
// one float4 each (the MFMA's two k-slots are k+e and k+4+e, the same permutation on both operands).
// Both operands are K-contiguous, so fragments come straight from global/L2 with 16-byte loads:
// with M <= a few hundred rows no two waves of a workgroup share a fragment and LDS staging would
// only add a round trip (guide §5, "GEMV / M <= 16" row generalised to the split-K small-M case).
// ------------------------------------------------------------------------------------------------
template <int WK>
__global__ __launch_bounds__(64 * WK) void gemm32(GemmBatch gb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [WK][1024] partials (+ stats)
    const GemmP& p = gb.g[blockIdx.z];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    if (m0 >= p.M || n0 >= p.N) return;
    const int i = lane & 31, kh = lane >> 5;

    int am = m0 + i;
    if (am >= p.M) am = p.M - 1;
    int wn = n0 + i;
    if (wn >= p.N) wn = p.N - 1;
    const float* arow = p.A + (long long)(am / p.a_n1) * p.a_s0 + (long long)((am % p.a_n1) / p.a_n2) * p.a_s1 +
                        (long long)(am % p.a_n2) * p.a_s2;
    const float* wrow = p.W + (long long)wn * p.ldw;

    float mean = 0.f, rstd = 1.f;
    const bool ln = p.ln_g != nullptr;
    if (ln) {
        // LayerNorm statistics of the tile's 32 rows over K = 256 (two-pass, float32).
        float* st = smem + WK * 1024;
        for (int r = wave; r < 32; r += WK) {
            int rm = m0 + r;
            if (rm >= p.M) rm = p.M - 1;
            const float* rp = p.A + (long long)(rm / p.a_n1) * p.a_s0 + (long long)((rm % p.a_n1) / p.a_n2) * p.a_s1 +
                              (long long)(rm % p.a_n2) * p.a_s2;
            float4 v = *reinterpret_cast<const float4*>(rp + lane * 4);
            float s = wave_sum(v.x + v.y + v.z + v.w);
            float mu = s * (1.0f / 256.0f);
            float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
            float q = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw);
            if (lane == 0) {
                st[r * 2] = mu;
                st[r * 2 + 1] = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-5f);
            }
        }
        __syncthreads();
        mean = st[i * 2];
        rstd = st[i * 2 + 1];
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int ks = p.K / WK;
    const int k0 = wave * ks;
    const int kend = k0 + ks;
    int k = k0;
    for (; k + 32 <= kend; k += 32) {
        float4 a[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = k + 8 * u + 4 * kh;
            a[u] = *reinterpret_cast<const float4*>(arow + (long long)(kk / p.a_seg) * p.a_seg_stride + (kk % p.a_seg));
            w[u] = *reinterpret_cast<const float4*>(wrow + kk);
        }
        if (ln) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k + 8 * u + 4 * kh;
                const float4 g = *reinterpret_cast<const float4*>(p.ln_g + kk);
                const float4 b = *reinterpret_cast<const float4*>(p.ln_b + kk);
                a[u].x = (a[u].x - mean) * rstd * g.x + b.x;
                a[u].y = (a[u].y - mean) * rstd * g.y + b.y;
                a[u].z = (a[u].z - mean) * rstd * g.z + b.z;
                a[u].w = (a[u].w - mean) * rstd * g.w + b.w;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, w[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, w[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, w[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, w[u].w, acc, 0, 0, 0);
        }
    }
    for (; k + 8 <= kend; k += 8) {
        const int kk = k + 4 * kh;
        float4 a = *reinterpret_cast<const float4*>(arow + (long long)(kk / p.a_seg) * p.a_seg_stride + (kk % p.a_seg));
        const float4 w = *reinterpret_cast<const float4*>(wrow + kk);
        if (ln) {
            const float4 g = *reinterpret_cast<const float4*>(p.ln_g + kk);
            const float4 b = *reinterpret_cast<const float4*>(p.ln_b + kk);
            a.x = (a.x - mean) * rstd * g.x + b.x;
            a.y = (a.y - mean) * rstd * g.y + b.y;
            a.z = (a.z - mean) * rstd * g.z + b.z;
            a.w = (a.w - mean) * rstd * g.w + b.w;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc, 0, 0, 0);
    }

    // split-K reduction through LDS in fixed wave order (deterministic, no atomics).
#pragma unroll
    for (int r = 0; r < 16; ++r) smem[wave * 1024 + r * 64 + lane] = acc[r];
    __syncthreads();
    constexpr int NT = 64 * WK;
    constexpr int PER = 1024 / NT;   // WK <= 16 -> PER >= 1
    float sums[PER > 0 ? PER : 1];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int idx = tid + e * NT;
        float s = smem[idx];
#pragma unroll
        for (int w2 = 1; w2 < WK; ++w2) s += smem[w2 * 1024 + idx];
        sums[e] = s;
    }
    const int epi = p.epi;
    if (epi == EPI_GLU || epi == EPI_LSTM) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int idx = tid + e * NT;
            const int col = idx & 31;
            smem[idx] = sums[e] + (p.bias ? p.bias[min(n0 + col, p.N - 1)] : 0.f);
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int idx = tid + e * NT;
        const int reg = idx >> 6, ln_ = idx & 63;
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (ln_ >> 5);
        const int col = ln_ & 31;
        const int m = m0 + row, n = n0 + col;
        if (m >= p.M || n >= p.N) continue;
        const long long crow = (long long)(m / p.c_n) * p.c_s0 + (long long)(((m % p.c_n) + p.c_r0) % p.c_mod) * p.c_s1;
        if (epi == EPI_GLU) {
            if (col & 1) continue;
            const float a = smem[idx], g = smem[idx + 1];
            p.C[crow + (n >> 1)] = a * sigmoidf_(g);
        } else if (epi == EPI_LSTM) {
            if (col & 3) continue;
            const int tok = p.I[m];
            const float4 t = *reinterpret_cast<const float4*>(p.X + (long long)tok * (4 * RNNT_D) + n);
            const float gi = smem[idx] + t.x, gf = smem[idx + 1] + t.y, gg = smem[idx + 2] + t.z, go = smem[idx + 3] + t.w;
            const int j = n >> 2;
            const float cin = p.X2[(long long)m * RNNT_D + j];
            const float c2 = sigmoidf_(gf) * cin + sigmoidf_(gi) * tanhf(gg);
            const float h2 = sigmoidf_(go) * tanhf(c2);
            p.C[(long long)m * RNNT_D + j] = h2;
            p.Y2[(long long)m * RNNT_D + j] = c2;
        } else {
            float v = sums[e] + (p.bias ? p.bias[n] : 0.f);
            if (epi == EPI_SILU) v = v * sigmoidf_(v);
            else if (epi == EPI_RELU) v = fmaxf(v, 0.f);
            else if (epi == EPI_SCALE) v = v * p.alpha;
            else if (epi == EPI_RESID) v = p.R[crow + n] + p.alpha * v;
            else if (epi == EPI_DB) v = 10.0f * log10f(fmaxf(v, 1e-10f));
            else if (epi == EPI_TANH_ADD) {
                const int bi = m / p.x_n;
                const int fr = p.I ? p.I[bi] : (m % p.x_n);
                v = tanhf(v + p.X[(long long)bi * p.x_s0 + (long long)fr * p.x_s1 + n]);
            }
            p.C[crow + n] = v;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// gemm16<WK,NT>: the small-M workhorse.  One 16 x (16*NT) output tile per workgroup, K split over WK
// waves, v_mfma_f32_16x16x4_f32 (exact f32).  fp32 MFMA is only 256 FLOP/clk/CU, so at M = 64..192
// rows the lever is tile COUNT: 16-row tiles give 192..768 workgroups per GEMM instead of 48..192
// and every CU gets work.  Lane (i = l&15, kq = l>>4) feeds A[m0+i][k + 4*kq + e] and
// W[n0+16t+i][k + 4*kq + e], e = 0..3, from one float4 each (same K permutation on both operands).
// Addressing is division-free on the plain path; the general path (implicit-GEMM conv2, K/V append,
// rings, joint lattice) uses host-computed multiply-high magics.
// LayerNorm prologue: 16 lanes per row compute the two-pass statistics of the tile's 16 rows in
// parallel (one load round trip + 8 in-row shuffle steps).
// ------------------------------------------------------------------------------------------------
typedef float f32x4_ __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long long a_row_off(const GemmP& p, int m) {
    if (p.a_plain) return (long long)m * p.a_s2;
    const int q1 = fastdiv(m, p.a_n1, p.a_n1_magic, p.a_n1_shift);
    const int r1 = m - q1 * p.a_n1;
    const int q2 = fastdiv(r1, p.a_n2, p.a_n2_magic, p.a_n2_shift);
    const int r2 = m - fastdiv(m, p.a_n2, p.a_n2_magic, p.a_n2_shift) * p.a_n2;
    return (long long)q1 * p.a_s0 + (long long)q2 * p.a_s1 + (long long)r2 * p.a_s2;
}
__device__ __forceinline__ long long a_k_off(const GemmP& p, int kk) {
    if (p.a_plain) return kk;
    const int q = fastdiv(kk, p.a_seg, p.a_seg_magic, p.a_seg_shift);
    return (long long)q * p.a_seg_stride + (kk - q * p.a_seg);
}
__device__ __forceinline__ long long c_row_off(const GemmP& p, int m) {
    if (p.c_plain) return (long long)m * p.c_s1;
    const int q = fastdiv(m, p.c_n, p.c_n_magic, p.c_n_shift);
    int r = m - q * p.c_n + p.c_r0;
    if (r >= p.c_mod) r -= p.c_mod;
    return (long long)q * p.c_s0 + (long long)r * p.c_s1;
}

template <int WK, int MT, int NT>
__device__ __forceinline__ void gemm16_body(const GemmP& p) {
    // tile = (16*MT) x (16*NT) outputs per workgroup; every wave holds the full MT x NT accumulator set for its
    // K-slice, so one float4 of A feeds NT MFMAs and one float4 of W feeds MT (operand reuse in registers).
    __shared__ __attribute__((aligned(16))) float part[WK * MT * NT * 256];
    __shared__ float st[32 * MT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * (16 * MT), n0 = blockIdx.x * (16 * NT);
    if (m0 >= p.M || n0 >= p.N) return;
    const int i = lane & 15, kq = lane >> 4;
    if (p.act_idx) {   // uniform per workgroup: every wave evaluates the same 16*MT rows
        const int lim = ldgi(p.act_lim);
        bool any = false;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + 16 * mt + i;
            any = any || (m < p.M && ldgi(p.act_idx + m) < lim);
        }
        if (!__any(any)) return;
    }

    const float* arow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int am = min(m0 + 16 * mt + i, p.M - 1);
        arow[mt] = p.A + a_row_off(p, am);
        if (p.Asel) arow[mt] += (long long)(ldgi(p.Asel + am) ^ p.asel_invert) * p.asel_stride;
    }
    const float* wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wrow[t] = p.W + (long long)min(n0 + 16 * t + i, p.N - 1) * p.ldw;

    float mean[MT], rstd[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { mean[mt] = 0.f; rstd[mt] = 1.f; }
    const bool ln = p.ln_g != nullptr;
    if (ln) {
        const int grp = tid >> 4, l16 = tid & 15;
        for (int r = grp; r < 16 * MT; r += 4 * WK) {
            const float* rp = p.A + a_row_off(p, min(m0 + r, p.M - 1));
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ldg4(rp + 4 * (l16 + 16 * j));
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) sm += (v[j].x + v[j].y) + (v[j].z + v[j].w);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 16);
            const float mu = sm * (1.0f / 256.0f);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dx = v[j].x - mu, dy = v[j].y - mu, dz = v[j].z - mu, dw = v[j].w - mu;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 16);
            if (l16 == 0) {
                st[r * 2] = mu;
                st[r * 2 + 1] = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-5f);
            }
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            mean[mt] = st[(16 * mt + i) * 2];
            rstd[mt] = st[(16 * mt + i) * 2 + 1];
        }
    }

    f32x4_ acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = (f32x4_){0.f, 0.f, 0.f, 0.f};

    const int ks = p.K / WK;
    const int k0 = wave * ks, kend = k0 + ks;
    constexpr int UN = (MT * NT >= 8) ? 2 : 4;   // k-steps of 16 in flight per iteration (register budget)
    int k = k0;
    for (; k + 16 * UN <= kend; k += 16 * UN) {
        float4 a[UN][MT], w[UN][NT];
        if (p.dbg & 1) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[u][mt] = make_float4(1.f * k, 2.f, 3.f, 4.f + lane);
#pragma unroll
                for (int t = 0; t < NT; ++t) w[u][t] = make_float4(1.f, 2.f * k, 3.f + lane, 4.f);
            }
        } else {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kk = k + 16 * u + 4 * kq;
            const long long ko = a_k_off(p, kk);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[u][mt] = ldg4(arow[mt] + ko);
#pragma unroll
            for (int t = 0; t < NT; ++t) w[u][t] = ldg4(wrow[t] + kk);
        }
        }
        if (ln) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int kk = k + 16 * u + 4 * kq;
                const float4 g = ldg4(p.ln_g + kk);
                const float4 b = ldg4(p.ln_b + kk);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    a[u][mt].x = (a[u][mt].x - mean[mt]) * rstd[mt] * g.x + b.x;
                    a[u][mt].y = (a[u][mt].y - mean[mt]) * rstd[mt] * g.y + b.y;
                    a[u][mt].z = (a[u][mt].z - mean[mt]) * rstd[mt] * g.z + b.z;
                    a[u][mt].w = (a[u][mt].w - mean[mt]) * rstd[mt] * g.w + b.w;
                }
            }
        }
        if (p.dbg & 2) {
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[mt][t][0] += a[u][mt].x * w[u][t].x + a[u][mt].y * w[u][t].y + a[u][mt].z * w[u][t].z + a[u][mt].w * w[u][t].w;
        } else
#pragma unroll
        for (int u = 0; u < UN; ++u) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][mt].x, w[u][t].x, acc[mt][t], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][mt].y, w[u][t].y, acc[mt][t], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][mt].z, w[u][t].z, acc[mt][t], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][mt].w, w[u][t].w, acc[mt][t], 0, 0, 0);
        }
    }
    for (; k + 16 <= kend; k += 16) {
        const int kk = k + 4 * kq;
        const long long ko = a_k_off(p, kk);
        float4 a[MT], w[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = ldg4(arow[mt] + ko);
#pragma unroll
        for (int t = 0; t < NT; ++t) w[t] = ldg4(wrow[t] + kk);
        if (ln) {
            const float4 g = ldg4(p.ln_g + kk);
            const float4 b = ldg4(p.ln_b + kk);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                a[mt].x = (a[mt].x - mean[mt]) * rstd[mt] * g.x + b.x;
                a[mt].y = (a[mt].y - mean[mt]) * rstd[mt] * g.y + b.y;
                a[mt].z = (a[mt].z - mean[mt]) * rstd[mt] * g.z + b.z;
                a[mt].w = (a[mt].w - mean[mt]) * rstd[mt] * g.w + b.w;
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, w[t].x, acc[mt][t], 0, 0, 0);
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, w[t].y, acc[mt][t], 0, 0, 0);
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, w[t].z, acc[mt][t], 0, 0, 0);
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, w[t].w, acc[mt][t], 0, 0, 0);
            }
    }

    if (p.dbg & 4) {   // ablation: keep the accumulators alive, skip reduction and epilogue
        float sacc = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) sacc += acc[mt][t][0] + acc[mt][t][1] + acc[mt][t][2] + acc[mt][t][3];
        if (sacc == 12345.678f) p.C[0] = sacc;
        return;
    }
    // deterministic split-K reduction through LDS (fixed wave order)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[((wave * MT + mt) * NT + t) * 256 + r * 64 + lane] = acc[mt][t][r];
    __syncthreads();
    constexpr int NTH = 64 * WK, NEL = MT * NT * 256;
    const int epi = p.epi;
    const bool paired = (epi == EPI_GLU || epi == EPI_LSTM);
    for (int e0 = 0; e0 < NEL; e0 += NTH) {
        const int idx = e0 + tid;
        float sum = 0.f;
        if (idx < NEL) {
            sum = part[idx];
#pragma unroll
            for (int w2 = 1; w2 < WK; ++w2) sum += part[w2 * NEL + idx];
        }
        const int tile = idx >> 8, rem = idx & 255;
        const int mt = tile / NT, t = tile - mt * NT;
        const int reg = rem >> 6, ln_ = rem & 63;
        const int row = 16 * mt + (ln_ >> 4) * 4 + reg, col = 16 * t + (ln_ & 15);
        const int m = m0 + row, n = n0 + col;
        const bool inb = idx < NEL && m < p.M && n < p.N;
        if (paired) {
            __syncthreads();   // all partial reads of this pass done before slot 0 is overwritten
            if (idx < NEL) part[idx] = sum + (p.bias ? ldg1(p.bias + min(n, p.N - 1)) : 0.f);
            __syncthreads();
            if (!inb) continue;
            if (epi == EPI_GLU) {
                if (col & 1) continue;
                const float a = part[idx], g = part[idx + 1];
                stg1(p.C + c_row_off(p, m) + (n >> 1), a * sigmoidf_(g));
            } else {
                if (col & 3) continue;
                const int tok = ldgi(p.I + m);
                const float4 tb = ldg4(p.X + (long long)tok * (4 * RNNT_D) + n);
                const float gi = part[idx] + tb.x, gf = part[idx + 1] + tb.y, gg = part[idx + 2] + tb.z, go = part[idx + 3] + tb.w;
                const int j = n >> 2;
                long long si = (long long)m * (p.lstm_ld ? p.lstm_ld : RNNT_D) + j, so = si;
                if (p.Asel) {   // committed state in buffer sel, candidate written to the other buffer
                    const int sl = ldgi(p.Asel + m);
                    si += (long long)sl * p.asel_stride;
                    so += (long long)(sl ^ 1) * p.asel_stride;
                }
                const float cin = ldg1(p.X2 + si);
                const float c2 = sigmoidf_(gf) * cin + sigmoidf_(gi) * tanhf(gg);
                stg1(p.C + so, sigmoidf_(go) * tanhf(c2));
                stg1(p.Y2 + so, c2);
            }
            continue;
        }
        if (epi == EPI_ARGMAX) {
            // 16 consecutive lanes hold the 16 columns of one row of this tile: reduce, then one atomicMax per row.
            float v = inb ? sum + (p.bias ? ldg1(p.bias + n) : 0.f) : -INFINITY;
            int bi = inb ? n : 0x7fffffff;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                const float ov = __shfl_xor(v, o, 16);
                const int oi = __shfl_xor(bi, o, 16);
                if (ov > v || (ov == v && oi < bi)) { v = ov; bi = oi; }
            }
            if ((ln_ & 15) == 0 && idx < NEL && m < p.M && bi != 0x7fffffff && (!p.I || ldgi(p.I + m) < ldgi(p.nframes))) {
                unsigned u = __float_as_uint(v);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // order-preserving float -> uint
                const unsigned long long k64 = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)bi);
                atomicMax(p.key + m, k64);                        // max value, lowest index on ties (torch.argmax)
            }
            continue;
        }
        if (!inb) continue;
        const long long crow = c_row_off(p, m);
        float v = sum + (p.bias ? ldg1(p.bias + n) : 0.f);
        if (epi == EPI_SILU) v = v * sigmoidf_(v);
        else if (epi == EPI_RELU) v = fmaxf(v, 0.f);
        else if (epi == EPI_SCALE) v = v * p.alpha;
        else if (epi == EPI_RESID) v = ldg1(p.R + crow + n) + p.alpha * v;
        else if (epi == EPI_DB) v = 10.0f * log10f(fmaxf(v, 1e-10f));
        else if (epi == EPI_TANH_ADD) {
            const int bi = fastdiv(m, p.x_n, p.x_n_magic, p.x_n_shift);
            const int fr = p.I ? ldgi(p.I + bi) : (m - bi * p.x_n);
            v = tanhf(v + ldg1(p.X + (long long)bi * p.x_s0 + (long long)fr * p.x_s1 + n));
        }
        stg1(p.C + crow + n, v);
    }
}

template <int WK, int MT, int NT>
__global__ __launch_bounds__(64 * WK) void gemm16(GemmBatch gb) {
    gemm16_body<WK, MT, NT>(gb.g[blockIdx.z]);
}
// table-driven variant: one descriptor per blockIdx.z in device memory (wavefront schedule: up to 36 groups)
template <int WK, int MT, int NT>
__global__ __launch_bounds__(64 * WK) void gemm16_tab(const GemmP* __restrict__ tab) {
    const GemmP p = tab[blockIdx.z];
    gemm16_body<WK, MT, NT>(p);
}

// ------------------------------------------------------------------------------------------------
// gemm_ns<MT,NT>: grouped-launch GEMM, LDS-tiled, no split-K.  Workgroup = 4 waves (2x2), workgroup tile
// (32*MT) x (32*NT), each wave a (16*MT) x (16*NT) sub-tile over the full K; epilogue straight from the accumulators.
// Operands go global -> registers -> LDS in FULL 128-byte lines (8 consecutive lanes read one row's 32 floats):
// rocprofv3 showed that fragment-shaped loads (consecutive lanes = different rows) cost ~64 L1 accesses per wave
// instruction and held the MFMA pipe at 20 %.  K advances in blocks of 32 with two LDS buffers; the global loads of
// block b+1 are issued before the MFMAs of block b and written to LDS after them (one barrier per block).
// LDS rows are padded to 36 floats: the 16 rows of a ds_read_b128 fragment read start on 16 distinct 4-bank groups.
// The LayerNorm prologue is applied while the A tile is written to LDS.
// ------------------------------------------------------------------------------------------------
#ifdef NS_TRACE   // tools/microbench3.hip only: per-workgroup phase time stamps (100 MHz) of gemm_ns_body
__device__ long long ns_trace[8192 * 8];
#define NS_STAMP(k_) { if (threadIdx.x == 0 && blockIdx.x < 8192) ns_trace[blockIdx.x * 8 + (k_)] = (long long)__builtin_amdgcn_s_memrealtime(); }
#else
#define NS_STAMP(k_)
#endif
template <int MT, int NT, int NS_BK = 32, int PD = 1, bool ATANH = false, bool ANT = false>
__device__ __forceinline__ void gemm_ns_body(const GemmP& p, int bx, int by) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    constexpr int NS_LD = NS_BK + 4;       // row stride in floats: 16 fragment rows start on 16 distinct 4-bank groups
    constexpr int LPR = NS_BK / 4;         // float4 slots per tile row (8 for BK = 32, 16 for BK = 64)
    constexpr int RPP = 256 / LPR;         // tile rows staged per pass of the 256 threads
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
    constexpr int AJ = BM / RPP, WJ = BN / RPP;
    __shared__ __attribute__((aligned(16))) float As[2][BM * NS_LD];
    __shared__ __attribute__((aligned(16))) float Ws[2][BN * NS_LD];
    __shared__ float st[2 * BM];
    __shared__ __attribute__((aligned(16))) float lngb[2 * RNNT_D];   // LayerNorm gamma | beta (K = 256 when the prologue is on)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm0 = by * BM, bn0 = bx * BN;
    if (bm0 >= p.M || bn0 >= p.N) return;   // whole workgroup out of range (uniform)
    const int i = lane & 15, kq = lane >> 4;
    const bool ln = p.ln_g != nullptr;
    NS_STAMP(0)
    // staging assignment: thread covers tile rows srow + RPP*j, columns c4..c4+3 of the current K block
    const int c4 = (tid % LPR) * 4;
    const int srow = tid / LPR;
    const float* ag[AJ];
    const float* xg[AJ];
    const float* wg[WJ];
    float amean[AJ], arstd[AJ];
    constexpr bool atanh_ = ATANH;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int am = min(bm0 + srow + RPP * j, p.M - 1);
        ag[j] = p.A + a_row_off(p, am);
        xg[j] = atanh_ ? p.X + (long long)fastdiv(am, p.x_n, p.x_n_magic, p.x_n_shift) * p.x_s0 : p.A;
        if (p.Asel) ag[j] += (long long)(ldgi(p.Asel + am) ^ p.asel_invert) * p.asel_stride;
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) wg[j] = p.W + (long long)min(bn0 + srow + RPP * j, p.N - 1) * p.ldw;
    f32x4_ acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = (f32x4_){0.f, 0.f, 0.f, 0.f};

    const int wm = (wave >> 1) * (16 * MT), wn = (wave & 1) * (16 * NT);   // this wave's sub-tile inside the workgroup tile
    const int nb = p.K / NS_BK;
    const bool aplain = p.a_plain != 0;
    // PD K blocks are in flight in registers (slot = block % PD): a workgroup of this size keeps only ~12 KB per block
    // in flight, and with ~2 us to the Infinity Cache the K loop is bound by bytes in flight, not by the MFMA pipe.
    float4 ra[PD][AJ], rx[PD][ATANH ? AJ : 1], rw[PD][WJ];

#define NS_GLOAD(blk_, sl_)                                                                                    \
    {                                                                                                          \
        const int kk_ = (blk_) * NS_BK + c4;                                                                   \
        const long long ko_ = aplain ? (long long)kk_ : a_k_off(p, kk_);                                       \
        _Pragma("unroll") for (int j = 0; j < AJ; ++j) ra[sl_][j] = ANT ? ldg4_nt(ag[j] + ko_) : ldg4(ag[j] + ko_); \
        if (atanh_) { _Pragma("unroll") for (int j = 0; j < AJ; ++j) rx[sl_][ATANH ? j : 0] = ldg4(xg[j] + kk_); } \
        _Pragma("unroll") for (int j = 0; j < WJ; ++j) rw[sl_][j] = ldg4(wg[j] + kk_);                         \
    }
#define NS_LSTORE(buf_, sl_, blk_)                                                                             \
    {                                                                                                          \
        float4 rg = make_float4(1.f, 1.f, 1.f, 1.f), rb = make_float4(0.f, 0.f, 0.f, 0.f);                     \
        if (ln) {                                                                                              \
            rg = *reinterpret_cast<const float4*>(&lngb[(blk_) * NS_BK + c4]);                                 \
            rb = *reinterpret_cast<const float4*>(&lngb[RNNT_D + (blk_) * NS_BK + c4]);                       \
        }                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < AJ; ++j) {                                                       \
            float4 v_ = ra[sl_][j];                                                                            \
            if (atanh_) {                                                                                      \
                const float4 x_ = rx[sl_][ATANH ? j : 0];                                                      \
                v_.x = tanhf(v_.x + x_.x);                                                                     \
                v_.y = tanhf(v_.y + x_.y);                                                                     \
                v_.z = tanhf(v_.z + x_.z);                                                                     \
                v_.w = tanhf(v_.w + x_.w);                                                                     \
            }                                                                                                  \
            if (ln) {                                                                                          \
                v_.x = (v_.x - amean[j]) * arstd[j] * rg.x + rb.x;                                             \
                v_.y = (v_.y - amean[j]) * arstd[j] * rg.y + rb.y;                                             \
                v_.z = (v_.z - amean[j]) * arstd[j] * rg.z + rb.z;                                             \
                v_.w = (v_.w - amean[j]) * arstd[j] * rg.w + rb.w;                                             \
            }                                                                                                  \
            *reinterpret_cast<float4*>(&As[buf_][(srow + RPP * j) * NS_LD + c4]) = v_;                         \
        }                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < WJ; ++j)                                                         \
            *reinterpret_cast<float4*>(&Ws[buf_][(srow + RPP * j) * NS_LD + c4]) = rw[sl_][j];                 \
    }

#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < nb) NS_GLOAD(d, d)
    // the LayerNorm statistics are only needed when a block is written to LDS: their loads travel with the first blocks'
    if (ln) {
        lngb[tid] = ldg1(p.ln_g + tid);
        lngb[RNNT_D + tid] = ldg1(p.ln_b + tid);
        // statistics of the BM rows: 16 lanes per row, all rows of a lane group loaded before the first reduction (one
        // memory round trip instead of BM/16)
        const int grp = tid >> 4, l16 = tid & 15;
        constexpr int RG = BM / 16;
        float4 v[RG][4];
#pragma unroll
        for (int q = 0; q < RG; ++q) {
            const float* rp = p.A + a_row_off(p, min(bm0 + grp + 16 * q, p.M - 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[q][j] = ldg4(rp + 4 * (l16 + 16 * j));
        }
#pragma unroll
        for (int q = 0; q < RG; ++q) {
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) sm += (v[q][j].x + v[q][j].y) + (v[q][j].z + v[q][j].w);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 16);
            const float mu = sm * (1.0f / 256.0f);
            float qq = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dx = v[q][j].x - mu, dy = v[q][j].y - mu, dz = v[q][j].z - mu, dw = v[q][j].w - mu;
                qq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 16);
            if (l16 == 0) {
                st[(grp + 16 * q) * 2] = mu;
                st[(grp + 16 * q) * 2 + 1] = 1.0f / sqrtf(qq * (1.0f / 256.0f) + 1e-5f);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        amean[j] = ln ? st[(srow + RPP * j) * 2] : 0.f;
        arstd[j] = ln ? st[(srow + RPP * j) * 2 + 1] : 1.f;
    }
    NS_STAMP(1)
    NS_LSTORE(0, 0, 0)
    __syncthreads();
    NS_STAMP(2)
    for (int blk0 = 0; blk0 < nb; blk0 += PD) {
#pragma unroll
        for (int jj = 0; jj < PD; ++jj) {
            const int blk = blk0 + jj;
            if (blk < nb) {   // uniform
                const int buf = blk & 1;
                if (blk + PD < nb) NS_GLOAD(blk + PD, jj)   // slot jj was written to LDS one block ago
#pragma unroll
                for (int u = 0; u < NS_BK / 16; ++u) {
                    float4 a[MT], w[NT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(&As[buf][(wm + 16 * mt + i) * NS_LD + 16 * u + 4 * kq]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) w[t] = *reinterpret_cast<const float4*>(&Ws[buf][(wn + 16 * t + i) * NS_LD + 16 * u + 4 * kq]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, w[t].x, acc[mt][t], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, w[t].y, acc[mt][t], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, w[t].z, acc[mt][t], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, w[t].w, acc[mt][t], 0, 0, 0);
                }
                if (blk + 1 < nb) NS_LSTORE(buf ^ 1, (jj + 1) % PD, blk + 1)
                __syncthreads();
            }
        }
    }
#undef NS_GLOAD
#undef NS_LSTORE
    NS_STAMP(3)
    const int m0 = bm0 + wm, n0 = bn0 + wn;
    if (m0 >= p.M || n0 >= p.N) return;     // this wave's sub-tile is out of range (all barriers are behind us)

    // epilogue straight from the accumulators: lane (i, kq) holds rows 4*kq + r, column i of every 16x16 tile
    const int epi = p.epi;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + 16 * t + i;
        const bool nin = n < p.N;
        const float bias = (p.bias && nin) ? ldg1(p.bias + n) : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * mt + 4 * kq + r;
                const bool inb = nin && m < p.M;
                float v = acc[mt][t][r] + bias;
                if (epi == EPI_GLU) {
                    const float g = __shfl_xor(v, 1, 64);               // (value, gate) in adjacent columns / lanes
                    if (inb && !(i & 1)) stg1(p.C + c_row_off(p, m) + (n >> 1), v * sigmoidf_(g));
                    continue;
                }
                if (!inb) continue;
                const long long crow = c_row_off(p, m);
                if (epi == EPI_SILU) v = v * sigmoidf_(v);
                else if (epi == EPI_RELU) v = fmaxf(v, 0.f);
                else if (epi == EPI_SCALE) v = v * p.alpha;
                else if (epi == EPI_RESID) v = ldg1(p.R + crow + n) + p.alpha * v;
                else if (epi == EPI_DB) v = 10.0f * log10f(fmaxf(v, 1e-10f));
                else if (epi == EPI_TANH_ADD) {
                    const int bi = fastdiv(m, p.x_n, p.x_n_magic, p.x_n_shift);
                    const int fr = p.I ? ldgi(p.I + bi) : (m - bi * p.x_n);
                    v = tanhf(v + ldg1(p.X + (long long)bi * p.x_s0 + (long long)fr * p.x_s1 + n));
                }
                stg1(p.C + crow + n, v);
            }
        }
    }
    NS_STAMP(4)
}

// single-descriptor launch of the LDS-tiled GEMM (conv2 implicit GEMM at M ~ 36 k rows): 2-D grid, descriptor in kernarg
template <int MT, int NT, int BK = 32, int PD = 2, bool ATANH = false, bool ANT = false>
__global__ __launch_bounds__(256) void gemm_ns(GemmBatch gb, int ntn, int ntm) {
    // 1-D grid per descriptor, dealt round-robin over the 8 XCDs: XCD x runs M-tiles x, x+8, ... and, back to back, all
    // column tiles of each, so an A row block (for conv2: 590 KB of implicit-GEMM input) is fetched into ONE L2 instead
    // of into the L2 of every XCD a column tile landed on.  Placement is a speed hint only.
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int mt = (slot / ntn) * 8 + xcd;
    if (mt >= ntm) return;
    gemm_ns_body<MT, NT, BK, PD, ATANH, ANT>(gb.g[blockIdx.z], slot % ntn, mt);
}

// XCD-aware work mapping (guide T1): workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with a
// private 4 MiB L2.  A wavefront stage multiplies 12 different weight matrices at once (12+ MB): dealt naively,
// every XCD touches all of them and the operands stream from the Infinity Cache.  Here the 8 XCDs are split into
// 8/X groups of X XCDs; descriptor g belongs to group g % (8/X), and inside the group column tile n runs on XCD
// n % X, its M-tiles back to back.  So a weight slice is fetched into ONE L2 and an activation block into X of them
// (X = 8: every XCD takes one column tile of every descriptor; X = 2: a descriptor lives on two XCDs).
// Placement is a speed hint only.
template <int MT, int NT, int BK = 32, int PD = 2>
__global__ __launch_bounds__(256) void gemm_ns_tab(const GemmP* __restrict__ tab, int n_desc, int ntn, int ntm, int X) {
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int G8 = 8 / X, grp = xcd / X, xin = xcd - grp * X;
    const int cpx = (ntn + X - 1) / X;           // column tiles per XCD and descriptor
    const int per = cpx * ntm;
    const int gi = slot / per, rem = slot - gi * per;
    const int ni = rem / ntm, m = rem - ni * ntm;
    const int g = gi * G8 + grp, n = ni * X + xin;
    if (g >= n_desc || n >= ntn) return;
    const GemmP p = tab[g];   // by-value copy: the fields live in SGPRs instead of being re-read inside the K loop
    gemm_ns_body<MT, NT, BK, PD, false>(p, n, m);
}

// ------------------------------------------------------------------------------------------------
// conv1_relu: y1[b][t][f][c] = relu(b1[c] + sum_{kh,kw} x[b][2t+kh][2f+kw] * w1[c][kh][kw])
// (Conv2d(1,256,3,2)+ReLU, wenet/transformer/subsampling.py:189-190).  Channels-last so that the
// conv2 implicit GEMM reads 768 contiguous floats per kernel row.  One thread per (b,t,f,c).
// ------------------------------------------------------------------------------------------------
// Virtual streams: v = c*B + b reads fbank[b][starts[c] .. ) (starts == null -> c = 0, start 0): the
// wavefront path subsamples several equal-length chunks of every stream in one launch.
__global__ void conv1_relu(const float* __restrict__ x, const float* __restrict__ w1t /*[9][256]*/,
                           const float* __restrict__ b1, float* __restrict__ y1, int B, int T, int t1,
                           const int* __restrict__ starts, int n_chunks) {
    const long long n = (long long)n_chunks * B * t1 * RNNT_F1 * RNNT_D;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id & 255);
        long long r = id >> 8;
        const int f = (int)(r % RNNT_F1);
        r /= RNNT_F1;
        const int t = (int)(r % t1);
        const int v = (int)(r / t1);
        const int cidx = v / B, b = v - cidx * B;
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
__device__ __forceinline__ void rel_attention_stream_body(const AttnP& P, float* smem, int t2cap) {
    const float* __restrict__ kc = P.kc;
    const float* __restrict__ vc = P.vc;
    const int tq = P.tq, T2 = P.T2;
    float* S = smem;                       // [4][t2cap] scores, then probabilities
    float* red = smem + 4 * t2cap;         // [16 groups][4 queries][64]
    __shared__ float linv[4];
    const int b = blockIdx.x / RNNT_H, h = blockIdx.x % RNNT_H;
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
    rel_attention_stream_body(p, att_smem, t2cap);
}
__global__ __launch_bounds__(256) void rel_attention_stream_tab(const AttnP* __restrict__ tab, int t2cap) {
    extern __shared__ __attribute__((aligned(16))) float att_smem[];
    const AttnP p = tab[blockIdx.z];
    rel_attention_stream_body(p, att_smem, t2cap);
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

// ------------------------------------------------------------------------------------------------
// Feature front-end (data/dataloader.py:15-41, torchaudio MelSpectrogram(center=True, pad_mode="reflect") + AmplitudeToDB):
// reflect_pad makes the n_fft/2-padded signal, the windowed DFT is a GEMM over implicit frames (row stride = hop) against
// interleaved (w cos, -w sin) rows, power_spectrum squares and adds the pairs, the mel projection is a second GEMM with
// the dB conversion as its epilogue.
// ------------------------------------------------------------------------------------------------
__global__ void reflect_pad(const float* __restrict__ x, float* __restrict__ y, int B, int n, int pad, long long ystride) {
    const long long total = (long long)B * ystride;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(id / ystride);
        const int i = (int)(id - (long long)b * ystride);
        float v = 0.f;
        if (i < n + 2 * pad) {
            int j = i - pad;
            if (j < 0) j = -j;                      // reflect without repeating the edge sample
            if (j >= n) j = 2 * (n - 1) - j;
            v = x[(long long)b * n + j];
        }
        y[id] = v;
    }
}
// spec [M][2*nfp] interleaved (re, im) -> pw [M][kp]: re^2 + im^2 for k < nfreq, 0 for the padding columns
__global__ void power_spectrum(const float* __restrict__ spec, float* __restrict__ pw, long long M, int nfreq, int kp, int ldspec) {
    const long long total = M * kp;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const long long m = id / kp;
        const int k = (int)(id - m * kp);
        float v = 0.f;
        if (k < nfreq) {
            const float2 c = *reinterpret_cast<const float2*>(spec + m * ldspec + 2 * k);
            v = c.x * c.x + c.y * c.y;
        }
        pw[id] = v;
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

__global__ __launch_bounds__(64) void beam_reduce(const float* __restrict__ logits, int ldl, int vocab, int blank, int k, int step,
                                                int n_steps, BeamOut o) {
    const int r = blockIdx.x, lane = threadIdx.x;
    if (!o.active[r]) return;
    const float* x = logits + (long long)r * ldl;
    float v[8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = lane + 64 * j;
        v[j] = idx < vocab ? x[idx] : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) se += (lane + 64 * j) < vocab ? expf(v[j] - mx) : 0.f;
    const float lse = logf(wave_sum(se));
    const float blank_lp = (x[blank] - mx) - lse;
    const float max_lp = (mx - mx) - lse;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = lane + 64 * j;
        v[j] = (idx < vocab && idx != blank) ? (v[j] - mx) - lse : -INFINITY;
    }
    int best_tok = 0;
    for (int t = 0; t < k; ++t) {
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
            o.top_lp[((long long)r * n_steps + step) * k + t] = bv;
            o.top_tok[((long long)r * n_steps + step) * k + t] = bi;
        }
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

// small helpers ------------------------------------------------------------------------------------
__global__ void fill_f32(float* p, float v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_i32(int* p, int v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
// gather the reference's att_cache layout [L][H][len][128] (K|V) of one stream.
__global__ void gather_att_cache(const float* __restrict__ kc, const float* __restrict__ vc, float* __restrict__ dst, int b, int B,
                                 long long kv_stride, int kv_start, int len) {
    const long long n = (long long)RNNT_L * RNNT_H * len * 128;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(id & 127);
        long long r = id >> 7;
        const int j = (int)(r % len);
        r /= len;
        const int h = (int)(r % RNNT_H);
        const int l = (int)(r / RNNT_H);
        const float* src = (e < 64 ? kc : vc) + (((long long)l * B + b) * kv_stride + kv_start + j) * RNNT_D + h * RNNT_DK + (e & 63);
        dst[id] = *src;
    }
}
// reference cnn_cache layout [L][1][256][30] of one stream = LayerNorm(norm_conv) of the last 30
// conv-module input rows (zeros before stream start).  One wave per (l, frame).
__global__ void gather_cnn_cache(const float* __restrict__ xring, const float* __restrict__ lng /*[L][256]*/,
                                 const float* __restrict__ lnb, float* __restrict__ dst, int b, int B, int cap, int pos) {
    const int l = blockIdx.x / RNNT_LORDER, i = blockIdx.x % RNNT_LORDER;
    const int lane = threadIdx.x;
    const int frame = pos - RNNT_LORDER + i;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (frame >= 0) {
        const float* xp = xring + (((long long)l * B + b) * cap + frame % cap) * RNNT_D;
        const float4 v = *reinterpret_cast<const float4*>(xp + lane * 4);
        const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.0f);
        const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
        const float rstd = 1.0f / sqrtf(wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / 256.0f) + 1e-5f);
        const float4 gg = *reinterpret_cast<const float4*>(lng + l * RNNT_D + lane * 4);
        const float4 bb = *reinterpret_cast<const float4*>(lnb + l * RNNT_D + lane * 4);
        o.x = dx * rstd * gg.x + bb.x;
        o.y = dy * rstd * gg.y + bb.y;
        o.z = dz * rstd * gg.z + bb.z;
        o.w = dw * rstd * gg.w + bb.w;
    }
    float* d = dst + (long long)l * RNNT_D * RNNT_LORDER;
    d[(lane * 4 + 0) * RNNT_LORDER + i] = o.x;
    d[(lane * 4 + 1) * RNNT_LORDER + i] = o.y;
    d[(lane * 4 + 2) * RNNT_LORDER + i] = o.z;
    d[(lane * 4 + 3) * RNNT_LORDER + i] = o.w;
}
