// Device kernels of the gfx950 streaming RNN-T path.  fp32 storage.  Numerics modes (rnnt_finalize_weights): exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32: bit-for-bit a k-ordered fmaf chain, the default parity mode) or 16-bit split-operand MFMA for the
// dense contractions (bf16x3, f16x3: ~1e-5 / ~1e-6 relative, parity-gated; plain bf16: perf mode, token-match rate reported).
//
// Kernel inventory by header (DESIGN.md §5 has the roofline per kernel):
//   rnnt_common    epilogue kinds, GemmP descriptor (generalised A / C addressing: implicit-GEMM conv2, K/V-cache append,
//                  ring buffers, implicit STFT frames), address-space-1 load/store helpers, wave reductions
//   rnnt_gemm      gemm16 (16-row tiles, split-K, LSTM-cell / fused-argmax epilogues), gemm_ns / gemm_ns_tab (LDS-tiled
//                  grouped GEMM, LayerNorm prologue, XCD-aware mapping) -- exact f32 (v_mfma_f32_16x16x4_f32)
//   rnnt_gemm_bf   gemm_bf / gemm_bf_tab: the same GEMM contract on v_mfma_f32_16x16x32_{bf16,f16} with split operands
//                  (bf16x3 / f16x3: hi*hi + hi*lo + lo*hi, f32 accumulate; bf16: hi*hi)
//   rnnt_joint     joint_lattice_rows: the T x U joint lattice (tanh-add, projection, log-softmax) as one persistent kernel, pack_joint_w
//   rnnt_encoder   conv1_relu, layer_norm, rel_attention (LDS tiles, online softmax), rel_attention_stream (<= 4 queries,
//                  direct row streaming), dwconv_bn_silu, conv_ring_init
//   rnnt_decode    greedy_decide (launched path), greedy_stream (resident decoder), greedy_flow (cooperative experiment),
//                  publish_frames, probe_overlap_wait, unpack_keys
//   rnnt_frontend  reflect_pad, power_spectrum (rnnt_fbank)
//   rnnt_beam      beam_chain, beam_reduce, beam_gather, log_softmax_rows
//   rnnt_misc      fill_i32, gather_att_cache, gather_cnn_cache
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
    // padding mask of the conv module's output (convolution.py:148-150, full-context pass): row m = b * rowlen_n + f is stored only
    // if f < rowlen[b] (null: every row).  The masked rows keep their residual input: x += 0.
    const int* rowlen;
    int rowlen_n;
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
    // 16-bit operand planes of W (rnnt_gemm_bf.hip.h): same element index as W inside the weight blob; null = exact f32 only
    const unsigned short* Wh;
    const unsigned short* Wl;
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
__device__ __forceinline__ void stg4_nt(float* p, float4 v) { __builtin_nontemporal_store((f32x4g){v.x, v.y, v.z, v.w}, (RNNT_GAS f32x4g*)p); }
#else   // host pass of the single-source compile: never executed
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ldg4_nt(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float ldg1(const float* p) { return *p; }
__device__ __forceinline__ int ldgi(const int* p) { return *p; }
__device__ __forceinline__ void stg1(float* p, float v) { *p = v; }
__device__ __forceinline__ void stg1_nt(float* p, float v) { *p = v; }
__device__ __forceinline__ void stg4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void stg4_nt(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
#endif

__device__ __forceinline__ int fastdiv(int n, int d, unsigned magic, int shift) {   // exact for 0 <= n < 2^31
    return d == 1 ? n : (int)(__umulhi((unsigned)n, magic) >> shift);
}

// ONE logistic function for every kernel of the library (SiLU of the FFN / conv module, GLU gate, LSTM gates), in every numerics
// mode: 1 / (1 + 2^(-x log2 e)) on the hardware v_exp_f32 / v_rcp_f32 (1 ulp each; |error| < 3e-7 absolute), 5 VALU instead of the
// ~25 of `1.0f / (1.0f + expf(-x))` (libm expf + IEEE division) -- in ffn_as that epilogue alone was ~2.8 VALU per MFMA.
// Because every path (per-chunk API, layer-major whole-utterance schedule, wavefront fallback, resident decoders) calls THIS
// function, results that must agree bitwise between those paths still do (round 2 tried the fast form in ffn_as only, which
// moved a near-tie token relative to the per-chunk path, and dropped it).  exp2(+inf) = inf -> rcp -> 0; exp2(-inf) = 0 -> 1.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// The value of the butterfly partner lane ^ O for an ALL-REDUCE whose earlier steps ran in ascending O (1, 2, 4, ...): for O <= 8 a DPP
// modifier instead of the ds_bpermute __shfl_xor compiles to (quad permutes for 1 and 2; row_half_mirror / row_mirror for 4 and 8,
// which hand over lane 7 - i / 15 - i: the same VALUE as lane i ^ 4 / i ^ 8 once the quads / 8-groups are uniform, so the reduction
// is bitwise the butterfly's).  16 and 32 cross DPP rows and stay shuffles.
template <int O>
__device__ __forceinline__ float xor_partner(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (O == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    else if constexpr (O == 2) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    else if constexpr (O == 4) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    else if constexpr (O == 8) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    else return __shfl_xor(v, O, 64);
#else
    return v;
#endif
}
// (descending butterfly, bitwise the reductions every schedule has shared since round 1: not re-ordered)
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
