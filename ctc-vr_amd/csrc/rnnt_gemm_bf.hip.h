// 16-bit split-operand GEMM family: the gemm_ns contract (GemmP descriptor, LayerNorm / tanh-add prologues, every
// epilogue of ns_epilogue) on v_mfma_f32_16x16x32_{bf16,f16} (16x the f32 MFMA rate per instruction).
// Part of rnnt_kernels.hip.h (include that umbrella, not this file).
//
// Numerics (rnnt_finalize_weights mode):
//   bf16x3 / f16x3  every f32 operand x is carried as two 16-bit planes x = hi + lo (hi = round16(x), lo = round16(x - hi)) and
//                   a product is  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  accumulated in f32 (the dropped lo*lo term and the
//                   representation error are both ~2^-17 (bf16) / ~2^-22 (f16) of |a*b|): 3 MFMAs per 16x16x32 block instead of
//                   8 f32 MFMAs of twice the cycles each = 5.3x the exact-f32 rate.  Weights are split ONCE at finalize (two
//                   planes with the weight blob's element index), activations while their tile is staged into LDS.
//   bf16            hi*hi only (plain bf16 operands, f32 accumulate): the perf mode, not a parity mode.
// Reference contractions: positionwise_feed_forward.py:50-58, attention.py:109-131, subsampling.py:188-193,
// convolution.py:138-148, model/component/joint.py:48-69.
#pragma once

enum { RNNT_NUM_F32 = 0, RNNT_NUM_BF16X3 = 1, RNNT_NUM_BF16 = 2, RNNT_NUM_F16X3 = 3 };

typedef short bf16x8_ __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
typedef float f32x2v_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v_ __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v_ __attribute__((ext_vector_type(2)));

// two f32 -> one packed pair of 16-bit values (round to nearest even) + the two residuals
template <bool F16>
__device__ __forceinline__ unsigned pack2_16(float x, float y, float& rx, float& ry) {
    if constexpr (F16) {
        const f16x2v_ h = __builtin_convertvector((f32x2v_){x, y}, f16x2v_);
        rx = x - (float)h[0];
        ry = y - (float)h[1];
        return __builtin_bit_cast(unsigned, h);
    } else {
        const bf16x2v_ h = __builtin_convertvector((f32x2v_){x, y}, bf16x2v_);
        const unsigned u = __builtin_bit_cast(unsigned, h);
        rx = x - __uint_as_float(u << 16);
        ry = y - __uint_as_float(u & 0xffff0000u);
        return u;
    }
}
// 8 consecutive f32 -> hi plane (8 x 16 bit) and lo plane
template <bool F16, bool LO>
__device__ __forceinline__ void split8_16(const float4& a, const float4& b, uint4& hi, uint4& lo) {
    float r0, r1, r2, r3, r4, r5, r6, r7, d0, d1;
    hi.x = pack2_16<F16>(a.x, a.y, r0, r1);
    hi.y = pack2_16<F16>(a.z, a.w, r2, r3);
    hi.z = pack2_16<F16>(b.x, b.y, r4, r5);
    hi.w = pack2_16<F16>(b.z, b.w, r6, r7);
    if constexpr (LO) {
        lo.x = pack2_16<F16>(r0, r1, d0, d1);
        lo.y = pack2_16<F16>(r2, r3, d0, d1);
        lo.z = pack2_16<F16>(r4, r5, d0, d1);
        lo.w = pack2_16<F16>(r6, r7, d0, d1);
    }
}
template <bool F16>
__device__ __forceinline__ f32x4_ mfma16_(const uint4& a, const uint4& b, f32x4_ c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_, a), __builtin_bit_cast(f16x8_, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_, a), __builtin_bit_cast(bf16x8_, b), c, 0, 0, 0);
}

// finalize: every float of the weight blob -> its hi / lo 16-bit planes (same element index)
template <bool F16>
__global__ void split_planes(const float* __restrict__ src, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, long long n8) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += (long long)gridDim.x * blockDim.x) {
        const float4 a = ldg4(src + 8 * e), b = ldg4(src + 8 * e + 4);
        uint4 h, l;
        split8_16<F16, true>(a, b, h, l);
        *reinterpret_cast<uint4*>(hi + 8 * e) = h;
        *reinterpret_cast<uint4*>(lo + 8 * e) = l;
    }
}

__device__ __forceinline__ uint4 ldg_u4(const unsigned short* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
    const u32x4g v = *(const RNNT_GAS u32x4g*)p;
    return make_uint4(v[0], v[1], v[2], v[3]);
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}

// LDS image of an operand tile: rows of 32 k (64 bytes = 4 chunks of 8 k); chunk c of row r lives at slot c ^ swz(r).  The 16
// lanes ds_read_b128 serves in one LDS cycle are NOT 16 consecutive lanes ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS): with
// fragment lane (i = l & 15, q = l >> 4) reading row i, chunk q, this XOR puts each such group on 16 distinct 16-byte slots of
// the 256-byte bank row, so fragment reads are conflict-free without padding.
__device__ __forceinline__ int bf_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }

// ------------------------------------------------------------------------------------------------
// gemm_bf_body<NSPLIT,F16,MT,NT,ATANH>: workgroup = 4 waves (2x2), tile (32*MT) x (32*NT) with MT, NT <= 2, K in blocks of 32 (one MFMA k-step),
// two LDS buffers, next block's global loads issued before the MFMAs of the current one and written (split into planes) after
// them; one barrier per block.  A: f32 in HBM (generalised GemmP addressing), 8 consecutive k per thread = full 128-byte lines
// per 4 threads; W: 16-byte chunks of the pre-split planes.  Epilogue from the accumulators (ns_epilogue).
// ------------------------------------------------------------------------------------------------
template <int NSPLIT, bool F16, int MT, int NT, bool ATANH = false>
__device__ __forceinline__ void gemm_bf_body(const GemmP& p, int bx, int by) {
    static_assert(MT <= 2 && NT <= 2, "128-row tiles gave run-to-run different LayerNorm statistics (cause not found): removed in round 3");
    constexpr int BM = 32 * MT, BN = 32 * NT, BK = 32;
    constexpr bool LO = NSPLIT == 2;
    constexpr int ACH = BM * 4, WCH = BN * 4;                 // 16-byte chunks per plane and K block
    constexpr int AJ = (ACH + 255) / 256, WJ = (WCH + 255) / 256;
    __shared__ uint4 Ah[2][ACH], Wh[2][WCH];
    __shared__ uint4 Al[LO ? 2 : 1][LO ? ACH : 1], Wl[LO ? 2 : 1][LO ? WCH : 1];
    __shared__ float st[2 * BM];
    __shared__ __attribute__((aligned(16))) float lngb[2 * RNNT_D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm0 = by * BM, bn0 = bx * BN;
    if (bm0 >= p.M || bn0 >= p.N) return;   // uniform
    const int i = lane & 15, kq = lane >> 4;
    const bool ln = p.ln_g != nullptr;
    const int srow = tid >> 2, sc = tid & 3;                  // staging: tile row srow + 64*j, chunk sc (k = 8*sc .. 8*sc+7)
    const float* ag[AJ];
    const float* xg[ATANH ? AJ : 1];
    const unsigned short* whg[WJ];
    const unsigned short* wlg[WJ];
    float amean[AJ], arstd[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int am = min(bm0 + srow + 64 * j, p.M - 1);
        ag[j] = p.A + a_row_off(p, am);
        if constexpr (ATANH) xg[j] = p.X + (long long)fastdiv(am, p.x_n, p.x_n_magic, p.x_n_shift) * p.x_s0;
        if (p.Asel) ag[j] += (long long)(ldgi(p.Asel + am) ^ p.asel_invert) * p.asel_stride;
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const long long wo = (long long)min(bn0 + srow + 64 * j, p.N - 1) * p.ldw;
        whg[j] = p.Wh + wo;
        wlg[j] = p.Wl + wo;
    }
    f32x4_ acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    const int wm = (wave >> 1) * (16 * MT), wn = (wave & 1) * (16 * NT);
    const int nb = p.K / BK;
    const bool aplain = p.a_plain != 0;
    float4 ra[AJ][2], rx[ATANH ? AJ : 1][2];
    uint4 rwh[WJ], rwl[LO ? WJ : 1];

    auto gload = [&](int blk) {
        const int kk = blk * BK + 8 * sc;
        const long long ko = aplain ? (long long)kk : a_k_off(p, kk);
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            if (AJ * 256 == ACH || srow + 64 * j < BM) {
                ra[j][0] = ldg4(ag[j] + ko);
                ra[j][1] = ldg4(ag[j] + ko + 4);
                if constexpr (ATANH) { rx[j][0] = ldg4(xg[j] + kk); rx[j][1] = ldg4(xg[j] + kk + 4); }
            }
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            if (WJ * 256 == WCH || srow + 64 * j < BN) {
                rwh[j] = ldg_u4(whg[j] + kk);
                if constexpr (LO) rwl[j] = ldg_u4(wlg[j] + kk);
            }
        }
    };
    auto lstore = [&](int buf, int blk) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int r = srow + 64 * j;
            if (AJ * 256 == ACH || r < BM) {
                float4 v0 = ra[j][0], v1 = ra[j][1];
                if constexpr (ATANH) {
                    v0.x = tanhf(v0.x + rx[j][0].x); v0.y = tanhf(v0.y + rx[j][0].y); v0.z = tanhf(v0.z + rx[j][0].z); v0.w = tanhf(v0.w + rx[j][0].w);
                    v1.x = tanhf(v1.x + rx[j][1].x); v1.y = tanhf(v1.y + rx[j][1].y); v1.z = tanhf(v1.z + rx[j][1].z); v1.w = tanhf(v1.w + rx[j][1].w);
                }
                if (ln) {
                    const int kk = blk * BK + 8 * sc;
                    const float4 g0 = *reinterpret_cast<const float4*>(&lngb[kk]), g1 = *reinterpret_cast<const float4*>(&lngb[kk + 4]);
                    const float4 b0 = *reinterpret_cast<const float4*>(&lngb[RNNT_D + kk]), b1 = *reinterpret_cast<const float4*>(&lngb[RNNT_D + kk + 4]);
                    const float mu = amean[j], rs = arstd[j];
                    v0.x = (v0.x - mu) * rs * g0.x + b0.x; v0.y = (v0.y - mu) * rs * g0.y + b0.y;
                    v0.z = (v0.z - mu) * rs * g0.z + b0.z; v0.w = (v0.w - mu) * rs * g0.w + b0.w;
                    v1.x = (v1.x - mu) * rs * g1.x + b1.x; v1.y = (v1.y - mu) * rs * g1.y + b1.y;
                    v1.z = (v1.z - mu) * rs * g1.z + b1.z; v1.w = (v1.w - mu) * rs * g1.w + b1.w;
                }
                uint4 h, l;
                split8_16<F16, LO>(v0, v1, h, l);
                const int slot = r * 4 + (sc ^ bf_swz(r));
                Ah[buf][slot] = h;
                if constexpr (LO) Al[buf][slot] = l;
            }
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int r = srow + 64 * j;
            if (WJ * 256 == WCH || r < BN) {
                const int slot = r * 4 + (sc ^ bf_swz(r));
                Wh[buf][slot] = rwh[j];
                if constexpr (LO) Wl[buf][slot] = rwl[j];
            }
        }
    };

    gload(0);
    if (ln) {
        lngb[tid] = ldg1(p.ln_g + tid);
        lngb[RNNT_D + tid] = ldg1(p.ln_b + tid);
        // statistics of the BM rows: 16 lanes per row, all rows of a lane group loaded before the first reduction
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
        const int r = min(srow + 64 * j, BM - 1);
        amean[j] = ln ? st[r * 2] : 0.f;
        arstd[j] = ln ? st[r * 2 + 1] : 1.f;
    }
    lstore(0, 0);
    __syncthreads();
    const int fsw = (kq ^ bf_swz(i));        // wm, wn and 16*mt are multiples of 16: the swizzle depends on i only
    for (int blk = 0; blk < nb; ++blk) {
        const int buf = blk & 1;
        if (blk + 1 < nb) gload(blk + 1);
        uint4 ah[MT], al[LO ? MT : 1], bh[NT], bl[LO ? NT : 1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int slot = (wm + 16 * mt + i) * 4 + fsw;
            ah[mt] = Ah[buf][slot];
            if constexpr (LO) al[mt] = Al[buf][slot];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int slot = (wn + 16 * t + i) * 4 + fsw;
            bh[t] = Wh[buf][slot];
            if constexpr (LO) bl[t] = Wl[buf][slot];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (LO) {
                    acc[mt][t] = mfma16_<F16>(al[mt], bh[t], acc[mt][t]);
                    acc[mt][t] = mfma16_<F16>(ah[mt], bl[t], acc[mt][t]);
                }
                acc[mt][t] = mfma16_<F16>(ah[mt], bh[t], acc[mt][t]);
            }
        if (blk + 1 < nb) lstore(buf ^ 1, blk + 1);
        __syncthreads();
    }
    const int m0 = bm0 + wm, n0 = bn0 + wn;
    if (m0 >= p.M || n0 >= p.N) return;
    ns_epilogue<MT, NT>(p, acc, m0, n0, i, kq);
}

// single-descriptor launch, XCD-aware 1-D grid (same mapping as gemm_ns)
template <int NSPLIT, bool F16, int MT, int NT, bool ATANH = false>
__global__ __launch_bounds__(256) void gemm_bf(GemmBatch gb, int ntn, int ntm) {
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int mt = (slot / ntn) * 8 + xcd;
    if (mt >= ntm) return;
    gemm_bf_body<NSPLIT, F16, MT, NT, ATANH>(gb.g[blockIdx.z], slot % ntn, mt);
}
// table form (wavefront stages), same work mapping as gemm_ns_tab
template <int NSPLIT, bool F16, int MT, int NT>
__global__ __launch_bounds__(256) void gemm_bf_tab(const GemmP* __restrict__ tab, int n_desc, int ntn, int ntm, int X) {
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int G8 = 8 / X, grp = xcd / X, xin = xcd - grp * X;
    const int cpx = (ntn + X - 1) / X;
    const int per = cpx * ntm;
    const int gi = slot / per, rem = slot - gi * per;
    const int ni = rem / ntm, m = rem - ni * ntm;
    const int g = gi * G8 + grp, n = ni * X + xin;
    if (g >= n_desc || n >= ntn) return;
    const GemmP p = tab[g];
    gemm_bf_body<NSPLIT, F16, MT, NT, false>(p, n, m);
}
