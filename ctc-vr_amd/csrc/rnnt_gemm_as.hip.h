// gemm_as: A-stationary split-operand GEMM for the layer contractions at large M (layer-major encoder, full-context encoder).
// Part of rnnt_kernels.hip.h (include that umbrella, not this file).
//
// The LDS-tiled gemm_bf moves BOTH operands through LDS every 32 k (ds_write_b128 runs at ~79 B/clk/CU: at 64 x 64 tiles the
// LDS stores alone take as long as the MFMAs) and re-derives the LayerNorm / 16-bit planes of an A tile once per column tile.
// Here a workgroup owns 16*MT rows and ALL of K = 256: the rows are read, normalised and split ONCE into an LDS operand image
// (rnnt_fused.hip.h layout, conflict-free ds_read_b128 fragments), and every wave streams its own weight fragments from L2
// straight into registers in MFMA-fragment order (packed at finalize: one wave instruction = 1 KiB contiguous), 8 KiB per wave
// in flight ahead of the MFMAs.  No barrier in the main loop, A fragments are read once per k-step for four column tiles
// (8 ds_read_b128 per 48 MFMAs at MT = 4).  Up to three weight matrices share one staged A (linear_q / k / v).
// K = 1024 (ffn w_2, N = 256): four K phases re-stage the image, accumulators stay in registers.
// Products and their order are those of gemm_bf (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi per 32 k, k ascending).
// Reference contractions: positionwise_feed_forward.py:50-58, attention.py:109-131, convolution.py:138-148.
#pragma once

#ifdef AS_TRACE   // tools/gemm_check.hip only: per-wave phase time stamps (100 MHz real-time counter)
__device__ long long as_trace[4096 * 8];
#define AS_STAMP(k_) { if (lane == 0 && blockIdx.x < 1024) as_trace[(blockIdx.x * 4 + wave) * 8 + (k_)] = (long long)__builtin_amdgcn_s_memrealtime(); }
#else
#define AS_STAMP(k_)
#endif
#define AS_MT_ 3
#ifndef BW_WD
#define BW_WD 3       // k-steps the weight fragments of gemm_bw run ahead (1..3)
#endif
#ifndef BW_ABL
#define BW_ABL 0     // tools/gemm_check timing experiments only (wrong results): 1 no A loads, 2 no weight loads, 4 no MFMAs, 8 no LDS stores, 16 no barriers
#endif
#ifndef FFN_ABL2
#define FFN_ABL2 0   // tools/gemm_check timing experiments only: 1 no weight loads, 2 no MFMAs / A-fragment reads
#endif
struct AsBatch {
    GemmP g[3];
    const uint4* wp[3];     // fragment-major packed weights (pack_frag) of g[i].W
    int ng;
};

// Epilogue of one wave's (16*MT) x (16*NTW) accumulator block.  Every contraction of this file runs with the MFMA operands
// SWAPPED (A = weight fragment, B = row fragment; bitwise the same sums as the other order -- tools/gemm_check checksums): the tile
// comes out transposed, i.e. a lane owns ONE row (16 mt + i) and four CONSECUTIVE columns (16 t + 4 q + 0..3), so bias,
// activation, residual and GLU apply to the accumulator registers and every global access is 16 bytes per lane (8 for the halved
// GLU rows) with no LDS staging; a wave instruction covers 16 rows x 64 bytes.  (Round 2 parked each row tile in per-wave LDS rows
// and read it back as float4: gemm_as 51 -> 41 us on LN + w_1 + SiLU at M = 12032, 16.2 -> 13.5 us on a q/k/v-like launch.)
template <int MT, int NTW = 4>
__device__ __forceinline__ void as_epilogue_t(const GemmP& p, const f32x4_ (&acc)[MT][NTW], int m0, int n0, int lane) {
    const int i = lane & 15, q = lane >> 4;
    const int epi = p.epi;
    float4 bias[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) bias[t] = p.bias ? ldg4(p.bias + n0 + 16 * t + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + 16 * mt + i;
        if (m >= p.M) continue;
        const long long crow = c_row_off(p, m);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int n = n0 + 16 * t + 4 * q;
            float4 v = make_float4(acc[mt][t][0] + bias[t].x, acc[mt][t][1] + bias[t].y, acc[mt][t][2] + bias[t].z, acc[mt][t][3] + bias[t].w);
            if (epi == EPI_GLU) {                                   // (value, gate) interleaved columns -> n / 2
                *reinterpret_cast<float2*>(p.C + crow + (n >> 1)) = make_float2(v.x * sigmoidf_(v.y), v.z * sigmoidf_(v.w));
                continue;
            }
            if (epi == EPI_SILU) { v.x *= sigmoidf_(v.x); v.y *= sigmoidf_(v.y); v.z *= sigmoidf_(v.z); v.w *= sigmoidf_(v.w); }
            else if (epi == EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            else if (epi == EPI_SCALE) { v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha; }
            else if (epi == EPI_RESID) {
                const float4 r4 = ldg4(p.R + crow + n);
                v.x = r4.x + p.alpha * v.x; v.y = r4.y + p.alpha * v.y; v.z = r4.z + p.alpha * v.z; v.w = r4.w + p.alpha * v.w;
            }
            stg4(p.C + crow + n, v);
        }
    }
}

template <int NUM, int MT>
__global__ __launch_bounds__(256) void gemm_as(AsBatch P) {
    using C = FuseCfg<NUM>;
    constexpr bool F16 = C::F16, LO = C::PLANES == 2;
    constexpr int U = C::PLANES, R = 16 * MT, ROWB = C::ROWB;     // ROWB = 512: 256 k x 2 bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char as_op[];   // [PLANES][R][512]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int bm0 = blockIdx.x * R;
    const GemmP& p0 = P.g[0];
    if (bm0 >= p0.M) return;
    const int KP = p0.K >> 8;                                     // K phases of 256
    const unsigned char* rowp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rowp[mt] = as_op + (16 * mt + i) * ROWB;

    // ---- A staging: rows bm0.. of phase kp -> operand image (LayerNorm prologue when set; K = 256 then) -------------------------
    auto stage = [&](int kp) {
        if (p0.ln_g) {
            // 16 lanes per row (4 rows per wave and pass, the passes' reductions independent): two-pass statistics, then the
            // normalised row goes into the image as 8-float chunks c = l16 + 16 j
            constexpr int NP = R / 16;                             // passes: rows (wave * 4 + g) + 16 * pass
            const int g = lane >> 4, l16 = lane & 15;
            float4 v[NP][4];
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                const float* rp = p0.A + a_row_off(p0, min(bm0 + wave * 4 + g + 16 * ps, p0.M - 1));
#pragma unroll
                for (int j = 0; j < 2; ++j) { v[ps][2 * j] = ldg4(rp + 8 * (l16 + 16 * j)); v[ps][2 * j + 1] = ldg4(rp + 8 * (l16 + 16 * j) + 4); }
            }
            float mu[NP], rs[NP];
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) sm += (v[ps][j].x + v[ps][j].y) + (v[ps][j].z + v[ps][j].w);
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 16);
                mu[ps] = sm * (1.0f / 256.0f);
            }
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                float qq = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dx = v[ps][j].x - mu[ps], dy = v[ps][j].y - mu[ps], dz = v[ps][j].z - mu[ps], dw = v[ps][j].w - mu[ps];
                    qq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 16);
                rs[ps] = 1.0f / sqrtf(qq * (1.0f / 256.0f) + 1e-5f);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = l16 + 16 * j;
                const float4 g0 = ldg4(p0.ln_g + 8 * c), g1 = ldg4(p0.ln_g + 8 * c + 4), b0_ = ldg4(p0.ln_b + 8 * c), b1_ = ldg4(p0.ln_b + 8 * c + 4);
#pragma unroll
                for (int ps = 0; ps < NP; ++ps) {
                    const int r = wave * 4 + g + 16 * ps;
                    const float m_ = mu[ps], s_ = rs[ps];
                    float4 x0 = v[ps][2 * j], x1 = v[ps][2 * j + 1];
                    x0.x = (x0.x - m_) * s_ * g0.x + b0_.x; x0.y = (x0.y - m_) * s_ * g0.y + b0_.y; x0.z = (x0.z - m_) * s_ * g0.z + b0_.z; x0.w = (x0.w - m_) * s_ * g0.w + b0_.w;
                    x1.x = (x1.x - m_) * s_ * g1.x + b1_.x; x1.y = (x1.y - m_) * s_ * g1.y + b1_.y; x1.z = (x1.z - m_) * s_ * g1.z + b1_.z; x1.w = (x1.w - m_) * s_ * g1.w + b1_.w;
                    uint4 h, l;
                    split8_16<F16, LO>(x0, x1, h, l);
                    const int off = op_off<NUM>(r, c);
                    *reinterpret_cast<uint4*>(as_op + off) = h;
                    if constexpr (LO) *reinterpret_cast<uint4*>(as_op + R * ROWB + off) = l;
                }
            }
        } else {
            constexpr int CJ = R * 32 / 256;                       // 8-float chunks per thread
            float4 va[CJ], vb[CJ];
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                const int e = tid + 256 * j, r = e >> 5, c = e & 31;
                const float* ap = p0.A + a_row_off(p0, min(bm0 + r, p0.M - 1)) + (kp << 8) + 8 * c;
                va[j] = ldg4(ap);
                vb[j] = ldg4(ap + 4);
            }
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                const int e = tid + 256 * j, r = e >> 5, c = e & 31;
                uint4 h, l;
                split8_16<F16, LO>(va[j], vb[j], h, l);
                const int off = op_off<NUM>(r, c);
                *reinterpret_cast<uint4*>(as_op + off) = h;
                if constexpr (LO) *reinterpret_cast<uint4*>(as_op + R * ROWB + off) = l;
            }
        }
    };
    // ---- weight stream: one unit = one k-step (32 k) of four column tiles = 4 * U vectors per lane ----------------------------------
    auto bload = [&](uint4 (&b)[4 * U], const uint4* __restrict__ Wp, int KT, int grp, int kt) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < U; ++u) {
#if defined(__HIP_DEVICE_COMPILE__)
                typedef unsigned u32x4g_ __attribute__((ext_vector_type(4)));
                const u32x4g_ v = *(const RNNT_GAS u32x4g_*)(Wp + ((long long)((grp * 4 + t) * KT + kt) * U + u) * 64 + lane);
                b[t * U + u] = make_uint4(v[0], v[1], v[2], v[3]);
#else
                b[t * U + u] = Wp[((long long)((grp * 4 + t) * KT + kt) * U + u) * 64 + lane];
#endif
            }
    };
    auto mma = [&](f32x4_ (&acc)[MT][4], const uint4 (&b)[4 * U], int ks) {
        uint4 ah[MT], al[LO ? MT : 1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int off = ((4 * ks + q) ^ i) << 4;
            ah[mt] = *reinterpret_cast<const uint4*>(rowp[mt] + off);
            if constexpr (LO) al[mt] = *reinterpret_cast<const uint4*>(rowp[mt] + R * ROWB + off);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (LO) {
                    acc[mt][t] = mfma16_<F16>(b[t * U], al[mt], acc[mt][t]);
                    acc[mt][t] = mfma16_<F16>(b[t * U + 1], ah[mt], acc[mt][t]);
                }
                acc[mt][t] = mfma16_<F16>(b[t * U], ah[mt], acc[mt][t]);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x4_ acc[MT][4];
    uint4 b0[4 * U], b1[4 * U];
    auto zero = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[mt][t] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    };

    if (KP > 1) {
        // ---- one matrix, N = 256: this wave's column group = wave; K phases re-stage the image ------------------------------------
        const int KT = p0.K >> 5;
        const uint4* Wp = P.wp[0];
        zero();
        bload(b0, Wp, KT, wave, 0);
        for (int kp = 0; kp < KP; ++kp) {
            if (kp) __syncthreads();                               // every wave is done with the previous phase's image
            stage(kp);
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 8; ks += 2) {
                const int kt = kp * 8 + ks;
                bload(b1, Wp, KT, wave, kt + 1);
                mma(acc, b0, ks);
                if (kt + 2 < KT) bload(b0, Wp, KT, wave, kt + 2);
                mma(acc, b1, ks + 1);
            }
        }
        as_epilogue_t<MT>(p0, acc, bm0, wave * 64, lane);
        return;
    }
    // ---- K = 256: the image is staged once; (matrix, column group) pairs of this wave one after the other --------------------------
    AS_STAMP(0)
    stage(0);
    AS_STAMP(1)
    __syncthreads();
    AS_STAMP(2)
    int gi = 0, grp = wave;
    while (gi < P.ng && grp * 64 >= P.g[gi].N) { ++gi; grp = wave; }
    if (gi < P.ng) bload(b0, P.wp[gi], 8, grp, 0);
    while (gi < P.ng) {
        int ngi = gi, ngrp = grp + 4;                              // the pair after this one
        while (ngi < P.ng && ngrp * 64 >= P.g[ngi].N) { ++ngi; ngrp = wave; }
        zero();
        const uint4* Wp = P.wp[gi];
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2) {
            bload(b1, Wp, 8, grp, ks + 1);
            mma(acc, b0, ks);
            if (ks + 2 < 8) bload(b0, Wp, 8, grp, ks + 2);
            else if (ngi < P.ng) bload(b0, P.wp[ngi], 8, ngrp, 0);   // first unit of the next pair: in flight across the epilogue
            mma(acc, b1, ks + 1);
        }
        AS_STAMP(3)
        as_epilogue_t<MT>(P.g[gi], acc, bm0, grp * 64, lane);
        AS_STAMP(4)
        gi = ngi; grp = ngrp;
    }
}

// ------------------------------------------------------------------------------------------------
// ffn_as: one PositionwiseFeedForward module with its pre-norm and half-step residual (encoder_layer.py:216-223,250-255;
// positionwise_feed_forward.py:50-58), optionally followed by the block's final LayerNorm (:257-258), as ONE kernel:
//     Y = LN_out?( X + alpha * (W2 silu(W1 LN(X) + b1) + b2) )
// A workgroup owns 16*MT rows.  The [rows x 1024] hidden activation never leaves the CU (unfused it is 49 MB written and read
// back per FFN at M = 12032: ~20 us of HBM time against ~15 us of MFMA time): the hidden columns are produced 256 at a time
// into a second LDS operand image and consumed at once as a K = 256 slice of the w_2 contraction, whose accumulators stay in
// registers.  Weights stream from L2 in fragment order as in gemm_as (2 MB per workgroup: the kernel is bound by the per-CU
// L2 fetch rate, ~50 GB/s).  Same products and k order as the unfused gemm_bf launches.
// ------------------------------------------------------------------------------------------------
struct FfnP {
    const float* X;
    float* Y;                         // may alias X
    const float *ln_g, *ln_b;         // norm_ff / norm_ff_macaron
    const uint4 *w1p, *w2p;           // packed fragments of w_1 [1024][256] and w_2 [256][1024]
    const float *b1, *b2;
    const float *lno_g, *lno_b;       // norm_final (null: off)
    float alpha;
    int M;
    // optional SECOND FFN module applied to the first one's result rows while they are in LDS (w1p2 != null): the layer's last
    // launch continues into the next layer's macaron FFN (both are row-local), Y = r1 + alpha2 * FFN2(LN2(r1)), r1 = the first
    // module's result (after norm_final); the tail then belongs to the second module.  Needs FFN_LDS + FFN_XROWS bytes of LDS.
    const float *ln_g2, *ln_b2;
    const uint4 *w1p2, *w2p2;
    const float *b1_2, *b2_2;
    float alpha2;
    // optional tail (FFN-macaron -> self-attention projections): n_tail matrices applied to LN_t(result rows), K = 256, N = 256
    // each, epilogue / output map from the descriptors (linear_q to a buffer, linear_k / linear_v rows into the cache)
    int n_tail;
    const float *lnt_g, *lnt_b;       // norm_mha / norm_conv
    // single-contraction head instead of the FFN (attention output projection): Y = X + alpha * (A0 W^T + b2) with W = w2p packed
    // [256][256]; A0 rows are staged as they are (no LayerNorm), w1p / b1 / ln_* unused
    const float* A0;
    // optional head IN FRONT of the FFN (pointwise_conv2 + residual -> FFN, one launch): x' = X + (H0 rows * Wh^T + bh) with Wh = whp
    // packed [256][256]; x' lives in LDS only (LN input and residual of the FFN).  hrowlen: rows m with m % hrowlen_n >= hrowlen[m /
    // hrowlen_n] keep X (the conv module's padding mask, convolution.py:148-150).  Needs FFN_LDS + FFN_XROWS bytes of LDS.
    const float* H0;
    // dw.g != null: the head's input rows are not read from H0 but FORMED here: depthwise conv (k = 31, causal) + BatchNorm + SiLU of
    // the post-GLU rows dw.g (dwconv_lm's arithmetic and ring write-back, one thread per channel and 8-row run), straight into the
    // operand image -- the depthwise launch and its [M][256] round trip disappear (convolution.py:126-146)
    DwLmP dw;
    const uint4* whp;
    const float* bh;
    const int* hrowlen;
    int hrowlen_n;
    GemmP tg[3];
    const uint4* twp[3];
};
#define FFN_TIMG 51200                // byte offset of the tail's operand image (behind the result rows)
#define FFN_FLD 260
#ifndef FFN_KU
#define FFN_KU 2
#endif

// [Xop | Hop] (2 images), overlaid by the result rows [R][FFN_FLD] and, behind those, the tail's operand image
#define FFN_LDS(NUM_, NW_) ((size_t)FFN_TIMG + (size_t)FuseCfg<NUM_>::PLANES * 16 * AS_MT_ * FuseCfg<NUM_>::ROWB)
#define FFN_XROWS (16 * AS_MT_ * FFN_FLD * 4)      // the head's x' rows, behind everything else
// NW waves per workgroup (4 or 8): every wave owns 256 / NW of the 256 output columns of a contraction (NTW = 16 / NW column
// tiles).  NW = 8 puts two waves on every SIMD, so one wave's waits on weight fragments hide under the other's MFMAs.
template <int NUM, int MT, int NW = 4>
__global__ __launch_bounds__(64 * NW) void ffn_as(FfnP P) {
    constexpr int NTW = 16 / NW, CW = 16 * NTW, NT = 64 * NW;
    using C = FuseCfg<NUM>;
    constexpr bool F16 = C::F16, LO = C::PLANES == 2;
    constexpr int U = C::PLANES, R = 16 * MT, ROWB = C::ROWB, IMG = U * R * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char ffn_smem[];   // [Xop | Hop]; the result rows overlay them, the tail's image sits at FFN_TIMG, the head's x' rows behind FFN_LDS
    unsigned char* Xop = ffn_smem;
    unsigned char* Hop = ffn_smem + IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int bm0 = blockIdx.x * R;
    if (bm0 >= P.M) return;
    float* fin = reinterpret_cast<float*>(ffn_smem);                            // [R][FFN_FLD] after the last contraction
    static_assert(R * FFN_FLD * 4 <= FFN_TIMG && 2 * IMG <= FFN_TIMG + IMG, "result rows / both images must fit FFN_LDS");
    auto bload = [&](uint4 (&b)[NTW * U], const uint4* __restrict__ Wp, int KT, int ct0 /*first column tile*/, int kt) {
#if FFN_ABL2 & 1
        for (int t = 0; t < NTW * U; ++t) b[t] = make_uint4(0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
        return;
#endif
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int u = 0; u < U; ++u) {
#if defined(__HIP_DEVICE_COMPILE__)
                typedef unsigned u32x4g_ __attribute__((ext_vector_type(4)));
                const u32x4g_ v = *(const RNNT_GAS u32x4g_*)(Wp + ((long long)((ct0 + t) * KT + kt) * U + u) * 64 + lane);
                b[t * U + u] = make_uint4(v[0], v[1], v[2], v[3]);
#else
                b[t * U + u] = Wp[((long long)((ct0 + t) * KT + kt) * U + u) * 64 + lane];
#endif
            }
    };
    auto mma = [&](f32x4_ (&acc)[MT][NTW], const uint4 (&b)[NTW * U], int ks, const unsigned char* op) {
#if FFN_ABL2 & 2
        for (int t = 0; t < NTW * U; ++t) asm volatile("" :: "v"(b[t].x), "v"(b[t].y), "v"(b[t].z), "v"(b[t].w));
        return;
#endif
        uint4 ah[MT], al[LO ? MT : 1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned char* rp = op + (16 * mt + i) * ROWB + (((4 * ks + q) ^ i) << 4);
            ah[mt] = *reinterpret_cast<const uint4*>(rp);
            if constexpr (LO) al[mt] = *reinterpret_cast<const uint4*>(rp + R * ROWB);
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (LO) {
                    acc[mt][t] = mfma16_<F16>(b[t * U], al[mt], acc[mt][t]);
                    acc[mt][t] = mfma16_<F16>(b[t * U + 1], ah[mt], acc[mt][t]);
                }
                acc[mt][t] = mfma16_<F16>(b[t * U], ah[mt], acc[mt][t]);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x4_ yacc[MT][NTW], hacc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NTW; ++t) yacc[mt][t] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    float* xrows = reinterpret_cast<float*>(ffn_smem + FFN_TIMG + IMG);        // [R][FFN_FLD] x' of the head (only then allocated)
    if (P.H0 || P.dw.g) {
        // ---- head: H0 rows (or the depthwise conv's output rows, formed here) -> Xop, one K = 256 contraction, x' = X + result -> xrows ------------------------------
        uint4 hb0[NTW * U], hb1[NTW * U];
        bload(hb0, P.whp, 8, wave * NTW, 0);
        if (P.dw.g) {
            // thread = (channel c, 8-row runs tid >> 8, + NT / 256, ...): a run's rows of ONE stream share a sliding window of 8 + 30 inputs
            static_assert(NT % 256 == 0 && R % LM_FB == 0, "one thread per channel");
            const DwLmP& D_ = P.dw;
            const int c = tid & 255;
            float w[RNNT_KDW];
#pragma unroll
            for (int k = 0; k < RNNT_KDW; ++k) w[k] = ldg1(D_.wdw_t + k * RNNT_D + c);
            const float bd = ldg1(D_.bdw + c), bs = ldg1(D_.bn_s + c), bt = ldg1(D_.bn_t + c);
            // one piece = up to LM_FB consecutive rows of ONE stream starting at tile row r0: window loads (`load`), then outputs (`emit`)
            auto load = [&](float (&win)[LM_FB + RNNT_LORDER], int r0, int& b, int& f0, int& cnt, int room) {
                const int m0 = bm0 + r0;
                cnt = 0; b = 0; f0 = 0;
                if (m0 >= P.M) return;
                b = m0 / D_.F; f0 = m0 - b * D_.F;
                cnt = min(room, D_.F - f0);
                const float* gb = D_.g + ((long long)b * D_.gs + f0) * RNNT_D + c;
#pragma unroll
                for (int i2 = 0; i2 < LM_FB + RNNT_LORDER; ++i2) win[i2] = (i2 < cnt + RNNT_LORDER) ? ldg1(gb + (long long)i2 * RNNT_D) : 0.f;
            };
            auto emit = [&](const float (&win)[LM_FB + RNNT_LORDER], int r0, int b, int f0, int cnt) {
#pragma unroll
                for (int j = 0; j < LM_FB; ++j) {
                    if (j >= cnt) break;
                    const int f = f0 + j;
                    float acc = bd;
#pragma unroll
                    for (int k = 0; k < RNNT_KDW; ++k) acc = fmaf(w[k], win[j + k], acc);
                    float v = acc * bs + bt;
                    v = v * sigmoidf_(v);
                    op_store1<NUM>(Xop, R, r0 + j, c, v);
                    if (f >= D_.F - D_.cap) {
                        const long long rr = ((long long)b * D_.cap + (D_.pos + f) % D_.cap) * RNNT_D + c;
                        stg1(D_.gring + rr, win[j + RNNT_LORDER]);
                        stg1(D_.xring + rr, ldg1(D_.xres + ((long long)b * D_.F + f) * RNNT_D + c));
                    }
                }
            };
            // this thread's runs, three at a time with all their windows in flight at once (one L2 round trip per batch, not per run)
            constexpr int NG = NT / 256, NRUN = (R / LM_FB + NG - 1) / NG, NB = NRUN < 3 ? NRUN : 3;   // thread groups, runs per thread, batch
            static_assert(NRUN % NB == 0, "runs per thread in batches");
#pragma unroll
            for (int rb = 0; rb < NRUN; rb += NB) {
                float win[NB][LM_FB + RNNT_LORDER];
                int pb[NB], pf[NB], pc[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int run = (tid >> 8) + NG * (rb + u);
                    pc[u] = LM_FB; pb[u] = 0; pf[u] = 0;
                    if (run < R / LM_FB) load(win[u], run * LM_FB, pb[u], pf[u], pc[u], LM_FB);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int run = (tid >> 8) + NG * (rb + u), r0 = run * LM_FB;
                    if (run >= R / LM_FB) continue;
                    emit(win[u], r0, pb[u], pf[u], pc[u]);
                    if (pc[u] < LM_FB) {                            // the run crosses into the next stream (or ends the batch): second piece / zero rows
                        int b2, f2, c2;
                        float win2[LM_FB + RNNT_LORDER];
                        load(win2, r0 + pc[u], b2, f2, c2, LM_FB - pc[u]);
                        emit(win2, r0 + pc[u], b2, f2, c2);
                        for (int j = pc[u] + c2; j < LM_FB; ++j) op_store1<NUM>(Xop, R, r0 + j, c, 0.f);
                    }
                }
            }
        } else {
        constexpr int CJ = (R * 32 + NT - 1) / NT;
        float4 va[CJ], vb[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int e = min(tid + NT * j, R * 32 - 1), r = e >> 5, c = e & 31;
            const float* ap = P.H0 + (long long)min(bm0 + r, P.M - 1) * RNNT_D + 8 * c;
            va[j] = ldg4(ap);
            vb[j] = ldg4(ap + 4);
        }
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int e = tid + NT * j, r = e >> 5, c = e & 31;
            if (e >= R * 32) continue;
            uint4 h, l;
            split8_16<F16, LO>(va[j], vb[j], h, l);
            const int off = op_off<NUM>(r, c);
            *reinterpret_cast<uint4*>(Xop + off) = h;
            if constexpr (LO) *reinterpret_cast<uint4*>(Xop + R * ROWB + off) = l;
        }
        }
        __syncthreads();                                            // Xop complete
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2) {
            bload(hb1, P.whp, 8, wave * NTW, ks + 1);
            mma(yacc, hb0, ks, Xop);
            if (ks + 2 < 8) bload(hb0, P.whp, 8, wave * NTW, ks + 2);
            mma(yacc, hb1, ks + 1, Xop);
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col = wave * CW + 16 * t + 4 * q;
            const float4 bb = ldg4(P.bh + col);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = 16 * mt + i, m = min(bm0 + row, P.M - 1);
                float4 x = ldg4(P.X + (long long)m * RNNT_D + col);
                const bool keep = P.hrowlen && (m % P.hrowlen_n) >= ldgi(P.hrowlen + m / P.hrowlen_n);
                if (!keep) { x.x += yacc[mt][t][0] + bb.x; x.y += yacc[mt][t][1] + bb.y; x.z += yacc[mt][t][2] + bb.z; x.w += yacc[mt][t][3] + bb.w; }
                *reinterpret_cast<float4*>(&xrows[row * FFN_FLD + col]) = x;
                yacc[mt][t] = (f32x4_){0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();                                            // x' complete, every wave is done with Xop
    }
    // the parameters of the FFN module being run (two modules when w1p2 is set)
    const float *ln_g = P.ln_g, *ln_b = P.ln_b, *b1p = P.b1, *b2p = P.b2, *lno_g = P.lno_g, *lno_b = P.lno_b;
    const uint4 *w1p = P.w1p, *w2p = P.w2p;
    float alpha = P.alpha;
    // Weight stream: per 256 hidden columns c, 8 k-steps of w_1's column group 4c + wave, then 8 k-steps of w_2's K slice c for
    // this wave's 64 output columns.  A pipeline unit = FFN_KU k-steps (8 KiB per wave each); one unit is consumed while the
    // next is in flight, across the hidden-slice epilogue and the barriers.  The kernel runs at the per-CU L2 fetch rate, which
    // is set by the bytes in flight.
    constexpr int KU = FFN_KU;
    uint4 b0[KU][NTW * U], b1[KU][NTW * U];     // (a ring of four units, three in flight, measured 40.1 vs 38.4 us: not kept)
    auto uload = [&](uint4 (&b)[KU][NTW * U], int phase /*2c: w_1, 2c+1: w_2*/, int ks0) {
        const int c = phase >> 1;
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            if (phase & 1) bload(b[k], w2p, 32, wave * NTW, c * 8 + ks0 + k);
            else bload(b[k], w1p, 8, c * 16 + wave * NTW, ks0 + k);
        }
    };
    auto umma = [&](f32x4_ (&acc)[MT][NTW], const uint4 (&b)[KU][NTW * U], int ks0, const unsigned char* op) {
#pragma unroll
        for (int k = 0; k < KU; ++k) mma(acc, b[k], ks0 + k, op);
    };
    // the tail (units of this wave: see there); the first matrix's first weight
    // unit is requested here, so it is in flight across the result-row phase (the weight stream would idle there otherwise)
    auto tload = [&](uint4 (&b)[KU][NTW * U], const uint4* Wp, int ct0, int ks0) {
#pragma unroll
        for (int k = 0; k < KU; ++k) bload(b[k], Wp, 8, ct0, ks0 + k);
    };
    constexpr int NPS = (R + 4 * NW - 1) / (4 * NW);            // result-row passes (4 NW rows each)
    float4 ykeep[NPS][4];
    const int n_mod = (P.w1p2 && !P.A0) ? 2 : 1;
    bool xl = P.H0 || P.dw.g;                                       // the module's input rows live in xrows (LDS), not in P.X
    // (two straight-line copies, not a rolled loop: around a rolled one hipcc hoists the per-lane weight addresses of both modules
    // and spills 160 registers)
#pragma unroll
    for (int mod = 0; mod < 2; ++mod) {
    if (mod >= n_mod) break;
    const bool last = mod + 1 == n_mod;
    if (mod == 1) {
        ln_g = P.ln_g2; ln_b = P.ln_b2; b1p = P.b1_2; b2p = P.b2_2; lno_g = nullptr; lno_b = nullptr; w1p = P.w1p2; w2p = P.w2p2; alpha = P.alpha2;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NTW; ++t) yacc[mt][t] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    }
    // ---- LN(x rows) -> Xop: 16 lanes per row, 4 rows per wave and pass ---------------------------------------------------------
    if (P.A0) {
        constexpr int CJ = (R * 32 + NT - 1) / NT;                 // 8-float chunks per thread
        float4 va[CJ], vb[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int e = min(tid + NT * j, R * 32 - 1), r = e >> 5, c = e & 31;
            const float* ap = P.A0 + (long long)min(bm0 + r, P.M - 1) * RNNT_D + 8 * c;
            va[j] = ldg4(ap);
            vb[j] = ldg4(ap + 4);
        }
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int e = tid + NT * j, r = e >> 5, c = e & 31;
            if (e >= R * 32) continue;
            uint4 h, l;
            split8_16<F16, LO>(va[j], vb[j], h, l);
            const int off = op_off<NUM>(r, c);
            *reinterpret_cast<uint4*>(Xop + off) = h;
            if constexpr (LO) *reinterpret_cast<uint4*>(Xop + R * ROWB + off) = l;
        }
    } else {
        constexpr int RP = 4 * NW, NP = (R + RP - 1) / RP;         // rows per pass (4 per wave), passes
        const int g = lane >> 4, l16 = lane & 15;
        float4 v[NP][4];
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int lr = min(wave * 4 + g + RP * ps, R - 1);
            if (xl) {                                               // the head's x' rows / the first module's result rows (LDS)
                const float* rp = xrows + lr * FFN_FLD;
#pragma unroll
                for (int j = 0; j < 2; ++j) { v[ps][2 * j] = *reinterpret_cast<const float4*>(rp + 8 * (l16 + 16 * j)); v[ps][2 * j + 1] = *reinterpret_cast<const float4*>(rp + 8 * (l16 + 16 * j) + 4); }
            } else {
                const float* rp = P.X + (long long)min(bm0 + lr, P.M - 1) * RNNT_D;
#pragma unroll
                for (int j = 0; j < 2; ++j) { v[ps][2 * j] = ldg4(rp + 8 * (l16 + 16 * j)); v[ps][2 * j + 1] = ldg4(rp + 8 * (l16 + 16 * j) + 4); }
            }
        }
        float mu[NP], rs[NP];
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) sm += (v[ps][j].x + v[ps][j].y) + (v[ps][j].z + v[ps][j].w);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 16);
            mu[ps] = sm * (1.0f / 256.0f);
        }
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            float qq = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dx = v[ps][j].x - mu[ps], dy = v[ps][j].y - mu[ps], dz = v[ps][j].z - mu[ps], dw = v[ps][j].w - mu[ps];
                qq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 16);
            rs[ps] = 1.0f / sqrtf(qq * (1.0f / 256.0f) + 1e-5f);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = l16 + 16 * j;
            const float4 g0 = ldg4(ln_g + 8 * c), g1 = ldg4(ln_g + 8 * c + 4), b0_ = ldg4(ln_b + 8 * c), b1_ = ldg4(ln_b + 8 * c + 4);
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                const int r = wave * 4 + g + RP * ps;
                if (r >= R) continue;
                const float m_ = mu[ps], s_ = rs[ps];
                float4 x0 = v[ps][2 * j], x1 = v[ps][2 * j + 1];
                x0.x = (x0.x - m_) * s_ * g0.x + b0_.x; x0.y = (x0.y - m_) * s_ * g0.y + b0_.y; x0.z = (x0.z - m_) * s_ * g0.z + b0_.z; x0.w = (x0.w - m_) * s_ * g0.w + b0_.w;
                x1.x = (x1.x - m_) * s_ * g1.x + b1_.x; x1.y = (x1.y - m_) * s_ * g1.y + b1_.y; x1.z = (x1.z - m_) * s_ * g1.z + b1_.z; x1.w = (x1.w - m_) * s_ * g1.w + b1_.w;
                uint4 h, l;
                split8_16<F16, LO>(x0, x1, h, l);
                const int off = op_off<NUM>(r, c);
                *reinterpret_cast<uint4*>(Xop + off) = h;
                if constexpr (LO) *reinterpret_cast<uint4*>(Xop + R * ROWB + off) = l;
            }
        }
    }
    if (P.A0) {
        bload(b0[0], P.w2p, 8, wave * NTW, 0);
        __syncthreads();                                            // Xop complete
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2) {
            bload(b0[1], P.w2p, 8, wave * NTW, ks + 1);
            mma(yacc, b0[0], ks, Xop);
            if (ks + 2 < 8) bload(b0[0], P.w2p, 8, wave * NTW, ks + 2);
            mma(yacc, b0[1], ks + 1, Xop);
        }
        __syncthreads();                                            // every wave is done with Xop
    } else {
    uload(b0, 0, 0);
    __syncthreads();                                                // Xop complete
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NTW; ++t) hacc[mt][t] = (f32x4_){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2 * KU) {
            uload(b1, 2 * c, ks + KU);
            umma(hacc, b0, ks, Xop);
            if (ks + 2 * KU < 8) uload(b0, 2 * c, ks + 2 * KU);
            else uload(b0, 2 * c + 1, 0);
            umma(hacc, b1, ks + KU, Xop);
        }
        {
            // hidden columns c*256 + wave*CW .. +CW = silu(acc + b1) -> Hop straight from the (transposed) C layout: a lane holds 4
            // consecutive columns of one row = an 8-byte half chunk of each plane (39.8 -> 38.5 us against the round-2 staging round trip)
            const float* bp = b1p + c * 256 + wave * CW + 4 * q;
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const float4 bb = ldg4(bp + 16 * t);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float4 x = make_float4(hacc[mt][t][0] + bb.x, hacc[mt][t][1] + bb.y, hacc[mt][t][2] + bb.z, hacc[mt][t][3] + bb.w);
                    x.x *= sigmoidf_(x.x); x.y *= sigmoidf_(x.y); x.z *= sigmoidf_(x.z); x.w *= sigmoidf_(x.w);
                    op_store4<NUM>(Hop, R, 16 * mt + i, wave * CW + 16 * t + 4 * q, x);
                }
            }
        }
        __syncthreads();                                            // Hop complete
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2 * KU) {
            uload(b1, 2 * c + 1, ks + KU);
            umma(yacc, b0, ks, Hop);
            if (ks + 2 * KU < 8) uload(b0, 2 * c + 1, ks + 2 * KU);
            else if (c < 3) uload(b0, 2 * c + 2, 0);
            umma(yacc, b1, ks + KU, Hop);
        }
        __syncthreads();                                            // every wave is done with Hop (after the last slice: with Xop too)
    }
    }
    if (last && P.n_tail > 0) { tload(b0, P.twp[0], wave * NTW, 0); tload(b1, P.twp[0], wave * NTW, KU); }   // (both buffers: a load issued behind the Y stores waits for their acknowledgement)
    // ---- result rows through LDS: residual, optional norm_final, float4 stores -------------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
            *reinterpret_cast<f32x4_*>(&fin[(16 * mt + i) * FFN_FLD + wave * CW + 16 * t + 4 * q]) = yacc[mt][t];
    __syncthreads();
    {
        const int l16 = tid & 15;
        float4 b2v[4], gv[4], bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b2v[j] = ldg4(b2p + 4 * (l16 + 16 * j));
            if (lno_g) { gv[j] = ldg4(lno_g + 4 * (l16 + 16 * j)); bv[j] = ldg4(lno_b + 4 * (l16 + 16 * j)); }
        }
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int row = (tid >> 4) + 4 * NW * ps, m = bm0 + row;
            if (row >= R) continue;
            const long long go = (long long)min(m, P.M - 1) * RNNT_D;
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = 4 * (l16 + 16 * j);
                const float4 y = *reinterpret_cast<const float4*>(&fin[row * FFN_FLD + col]);
                const float4 x = xl ? *reinterpret_cast<const float4*>(&xrows[row * FFN_FLD + col]) : ldg4(P.X + go + col);
                v[j] = make_float4(x.x + alpha * (y.x + b2v[j].x), x.y + alpha * (y.y + b2v[j].y), x.z + alpha * (y.z + b2v[j].z), x.w + alpha * (y.w + b2v[j].w));
            }
            if (lno_g) {
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) sm += (v[j].x + v[j].y) + (v[j].z + v[j].w);
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 16);
                const float mu = sm * (1.0f / 256.0f);
                float qq = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
                    qq += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 16);
                const float rstd = 1.0f / sqrtf(qq * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j].x = v[j].x * rstd * gv[j].x + bv[j].x; v[j].y = v[j].y * rstd * gv[j].y + bv[j].y;
                    v[j].z = v[j].z * rstd * gv[j].z + bv[j].z; v[j].w = v[j].w * rstd * gv[j].w + bv[j].w;
                }
            }
            if (!last) {                                            // the second module's input rows (same thread reads and writes a cell)
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&xrows[row * FFN_FLD + 4 * (l16 + 16 * j)]) = v[j];
                continue;
            }
            if (P.n_tail > 0) {                                     // with a tail the rows are stored behind its contractions (see there)
#pragma unroll
                for (int j = 0; j < 4; ++j) ykeep[ps][j] = v[j];
            } else if (m < P.M) {
#pragma unroll
                for (int j = 0; j < 4; ++j) stg4(P.Y + go + 4 * (l16 + 16 * j), v[j]);
            }
            if (P.n_tail > 0) {                                     // LN_t(result row) -> the tail's operand image
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) sm += (v[j].x + v[j].y) + (v[j].z + v[j].w);
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 16);
                const float mu = sm * (1.0f / 256.0f);
                float qq = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
                    qq += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 16);
                const float rstd = 1.0f / sqrtf(qq * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 gt = ldg4(P.lnt_g + 4 * (l16 + 16 * j)), bt = ldg4(P.lnt_b + 4 * (l16 + 16 * j));
                    op_store4<NUM>(ffn_smem + FFN_TIMG, R, row, 4 * (l16 + 16 * j),
                                   make_float4(v[j].x * rstd * gt.x + bt.x, v[j].y * rstd * gt.y + bt.y, v[j].z * rstd * gt.z + bt.z, v[j].w * rstd * gt.w + bt.w));
                }
            }
        }
    }
    if (!last) { xl = true; __syncthreads(); }                      // result rows complete in xrows; every thread is done with fin (= the images' space)
    }
    if (P.n_tail > 0) {
        static_assert(R * FFN_FLD * 4 <= FFN_TIMG, "the tail image must start behind the result rows");
        const unsigned char* Top = ffn_smem + FFN_TIMG;
        __syncthreads();                                            // tail image complete
        // All contractions first, all output stores last: vmcnt is one in-order counter for loads and stores, so a weight load
        // issued behind an epilogue's stores is not usable before those stores are acknowledged (measured: ~2.5 us per
        // epilogue in the critical path of every wave).  The result rows wait in registers for the same reason.
        // Units of this wave: (matrix g, 256-column half h) -> columns (h * NW + wave) * CW .. + CW of matrix g; at most three
        // (linear_q / k / v: 3 x 256 columns; pointwise_conv1: 1 x 512).
        f32x4_ tacc[3][MT][NTW];
        int ug = 0, uh = 0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (ug >= P.n_tail) break;
            int ng = ug, nh = uh + 1;
            if (nh * 256 >= P.tg[ug].N) { ++ng; nh = 0; }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NTW; ++t) tacc[u][mt][t] = (f32x4_){0.f, 0.f, 0.f, 0.f};
            const uint4* Wp = P.twp[ug];
            const int ct0 = (uh * NW + wave) * NTW;
#pragma unroll
            for (int ks = 0; ks < 8; ks += 2 * KU) {            // the FFN's unit pipeline: one unit consumed, the next in flight
                if (u > 0 || ks > 0) tload(b1, Wp, ct0, ks + KU);
                umma(tacc[u], b0, ks, Top);
                if (ks + 2 * KU < 8) tload(b0, Wp, ct0, ks + 2 * KU);
                else if (ng < P.n_tail) tload(b0, P.twp[ng], (nh * NW + wave) * NTW, 0);
                umma(tacc[u], b1, ks + KU, Top);
            }
            ug = ng; uh = nh;
        }
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {                          // the result rows kept from the result-row phase
            const int row = (tid >> 4) + 4 * NW * ps, m = bm0 + row;
            if (row < R && m < P.M) {
#pragma unroll
                for (int j = 0; j < 4; ++j) stg4(P.Y + (long long)m * RNNT_D + 4 * ((tid & 15) + 16 * j), ykeep[ps][j]);
            }
        }
        ug = 0; uh = 0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (ug >= P.n_tail) break;
            as_epilogue_t<MT, NTW>(P.tg[ug], tacc[u], bm0, (uh * NW + wave) * CW, lane);
            if (++uh * 256 >= P.tg[ug].N) { ++ug; uh = 0; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_bw: long-K contraction with N = 256 (conv2 implicit GEMM K = 2304, embed Linear K = 4864) -- 128 x 256 workgroup tile,
// A through LDS (generalised GemmP addressing, planes formed while the tile is staged, two buffers), WEIGHTS NOT through LDS:
// every wave owns 64 output columns and streams their fragments from L2 in packed order straight to registers, one k-step
// ahead.  Against gemm_bf's 128 x 128 tile this halves the LDS write traffic (ds_write_b128 at ~79 B/clk/CU was as long as
// the MFMAs), reads every A element from HBM once (N is not split) and gives each wave 96 MFMAs per 8 KiB of weights, which
// is what the per-CU L2 fetch rate (~45 GB/s) can feed.  Same products and k order as gemm_bf.
// ------------------------------------------------------------------------------------------------
// NW waves (4 or 8): every wave owns 256 / NW output columns of all 128 rows; 8 = two waves per SIMD.
template <int NUM, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm_bw(GemmP p, const uint4* __restrict__ wp) {
    using C = FuseCfg<NUM>;
    constexpr bool F16 = C::F16, LO = C::PLANES == 2;
    constexpr int U = C::PLANES, MT = 8, NTW = 16 / NW, NT = 64 * NW, AJ = NT >= 512 ? 1 : 512 / NT;   // AJ: 8-float A chunks per thread and k-step (NW = 16: the first 512 threads stage)
    __shared__ uint4 Ah[2][512];
    __shared__ uint4 Al[LO ? 2 : 1][LO ? 512 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int bm0 = blockIdx.x * 128;
    if (bm0 >= p.M) return;
    const int srow = (tid & 511) >> 2, sc = tid & 3;
    const bool stager = NT <= 512 || tid < 512;
    const float* ag[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) ag[j] = p.A + a_row_off(p, min(bm0 + srow + (NT / 4) * j, p.M - 1));
    const bool aplain = p.a_plain != 0;
    const int KT = p.K >> 5;
    // the A stream comes from HBM (1.09 GB for conv2): its loads run FOUR k-steps ahead of the MFMAs in a register ring
    float4 ra[4][AJ][2];
    auto gload = [&](float4 (&r)[AJ][2], int blk) {
        if (BW_ABL & 1) { for (int j = 0; j < AJ; ++j) { r[j][0] = make_float4(0.5f, 0.25f, 1.f, 2.f); r[j][1] = r[j][0]; } return; }
        if (!stager) return;
        const int kk = blk * 32 + 8 * sc;
        const long long ko = aplain ? (long long)kk : a_k_off(p, kk);
#pragma unroll
        for (int j = 0; j < AJ; ++j) { r[j][0] = ldg4_nt(ag[j] + ko); r[j][1] = ldg4_nt(ag[j] + ko + 4); }   // streamed once: keep the weights in L2
    };
    auto lstore = [&](int buf, const float4 (&r)[AJ][2]) {
        if (!stager) return;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int rr = srow + (NT / 4) * j;
            uint4 h, l;
            split8_16<F16, LO>(r[j][0], r[j][1], h, l);
            const int slot = rr * 4 + (sc ^ bf_swz(rr));
            Ah[buf][slot] = h;
            if constexpr (LO) Al[buf][slot] = l;
        }
    };
    auto bload = [&](uint4 (&b)[NTW * U], int kt) {
        if (BW_ABL & 2) { for (int t = 0; t < NTW * U; ++t) b[t] = make_uint4(0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); return; }
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int u = 0; u < U; ++u) {
#if defined(__HIP_DEVICE_COMPILE__)
                typedef unsigned u32x4g_ __attribute__((ext_vector_type(4)));
                const u32x4g_ v = *(const RNNT_GAS u32x4g_*)(wp + ((long long)((wave * NTW + t) * KT + kt) * U + u) * 64 + lane);
                b[t * U + u] = make_uint4(v[0], v[1], v[2], v[3]);
#else
                b[t * U + u] = wp[((long long)((wave * NTW + t) * KT + kt) * U + u) * 64 + lane];
#endif
            }
    };
    f32x4_ acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[mt][t] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    const int fsw = q ^ bf_swz(i);
    auto mma = [&](int buf, const uint4 (&b)[NTW * U]) {
        if (BW_ABL & 4) { for (int t = 0; t < NTW * U; ++t) asm volatile("" :: "v"(b[t].x), "v"(b[t].y), "v"(b[t].z), "v"(b[t].w)); return; }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int slot = (16 * mt + i) * 4 + fsw;
            const uint4 ah = Ah[buf][slot];
            uint4 al = ah;
            if constexpr (LO) al = Al[buf][slot];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if constexpr (LO) {
                    acc[mt][t] = mfma16_<F16>(b[t * U], al, acc[mt][t]);
                    acc[mt][t] = mfma16_<F16>(b[t * U + 1], ah, acc[mt][t]);
                }
                acc[mt][t] = mfma16_<F16>(b[t * U], ah, acc[mt][t]);
            }
        }
    };
    // weight fragments BW_WD k-steps ahead in a ring of four register sets (one k-step of MFMAs is shorter than an L2 round trip
    // under load: with one k-step of run-ahead every k-step began with a wait)
    uint4 wb[4][NTW * U];
#pragma unroll
    for (int k = 0; k < 4; ++k) gload(ra[k], k);                     // KT >= 4 and KT % 4 == 0 (host check)
#pragma unroll
    for (int k = 0; k < BW_WD; ++k) bload(wb[k], k);
    lstore(0, ra[0]);
    __syncthreads();
#ifdef AS_TRACE
    long long bw_t[4] = {0, 0, 0, 0}, bw_last = __builtin_amdgcn_s_memtime();
#define BW_STAMP(k_) { const long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); bw_t[k_] += t_ - bw_last; bw_last = t_; }
#else
#define BW_STAMP(k_)
#endif
    for (int blk = 0; blk < KT; blk += 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = blk + s;
            // LDS buffer k & 1 holds A(k); ring slots (s+1..s+3) % 4 hold A(k+1..k+3); slot s is free again
            // loads return in issue order: the weight fragments go first, the far-ahead A rows behind them
            if (k + BW_WD < KT) bload(wb[(s + BW_WD) & 3], k + BW_WD);
            if (k + 4 < KT) gload(ra[s], k + 4);
            BW_STAMP(0)                                              // load issue
            mma(s & 1, wb[s]);
            BW_STAMP(1)                                              // wait for the weight fragments + A fragment reads + MFMAs
            if (k + 1 < KT && !(BW_ABL & 8)) lstore((s + 1) & 1, ra[(s + 1) & 3]);
            BW_STAMP(2)                                              // wait for the A rows + split + LDS stores
            if (!(BW_ABL & 16)) __syncthreads();
            BW_STAMP(3)                                              // barrier
        }
    }
#ifdef AS_TRACE
    if (lane == 0 && blockIdx.x < 1024 && wave < 4) for (int k_ = 0; k_ < 4; ++k_) as_trace[(blockIdx.x * 4 + wave) * 8 + k_] = bw_t[k_];
#endif
    as_epilogue_t<MT, NTW>(p, acc, bm0, wave * 16 * NTW, lane);
}
