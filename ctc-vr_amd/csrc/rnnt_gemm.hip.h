// GEMM kernels (exact f32): gemm16 (16-row tiles, split-K), gemm_ns / gemm_ns_tab (LDS-tiled, grouped).
// The 16-bit split-operand family (bf16x3 / bf16 / f16x3) lives in rnnt_gemm_bf.hip.h and shares ns_epilogue below.
// Part of rnnt_kernels.hip.h (include that umbrella, not this file).
#pragma once

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
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kk = k + 16 * u + 4 * kq;
            const long long ko = a_k_off(p, kk);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[u][mt] = ldg4(arow[mt] + ko);
#pragma unroll
            for (int t = 0; t < NT; ++t) w[u][t] = ldg4(wrow[t] + kk);
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
        if (p.rowlen) { const int b_ = m / p.rowlen_n; if (m - b_ * p.rowlen_n >= ldgi(p.rowlen + b_)) continue; }   // padded frame (see GemmP::rowlen)
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

// Epilogue straight from the accumulators of an (16*MT) x (16*NT) wave sub-tile at (m0, n0): lane (i, kq) holds rows
// 4*kq + r, column i of every 16x16 tile (the C/D layout is the same for the f32 and the 16-bit MFMA shapes).
template <int MT, int NT>
__device__ __forceinline__ void ns_epilogue(const GemmP& p, const f32x4_ (&acc)[MT][NT], int m0, int n0, int i, int kq) {
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
                bool inb = nin && m < p.M;
                if (p.rowlen && inb) { const int b_ = m / p.rowlen_n; inb = m - b_ * p.rowlen_n < ldgi(p.rowlen + b_); }
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

    ns_epilogue<MT, NT>(p, acc, m0, n0, i, kq);
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
