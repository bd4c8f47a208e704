// Fused Conformer-block kernels of the streaming wavefront: one workgroup runs HALF a block for a tile of rows, with the
// residual stream, the MFMA operand image and the FFN hidden slab resident in LDS and the weights streamed from L2 straight
// into registers in MFMA-fragment order.  Part of rnnt_kernels.hip.h (include that umbrella, not this file).
//
//   block_front  x += 1/2 FFN_macaron(LN(x));  q, k, v = linear_{q,k,v}(LN(x))   (k, v appended to the layer's cache)
//   (rel_attention_stream_tab / rel_attention_tab between the two)
//   block_back   x += linear_out(att);  x += conv_module(LN(x));  x += 1/2 FFN(LN(x));  x = LN_final(x)
// (ConformerEncoderLayer.forward, wenet/transformer/encoder_layer.py:188-265; positionwise_feed_forward.py:50-58;
// attention.py:109-131,170-178; convolution.py:98-153.)
//
// Why: a (chunk, layer) pair of the 64-stream batch is 192 rows against 1.5 M weights.  As 11 launches per wavefront stage
// every launch is one workgroup lifetime of ramp + prologue + K loop + epilogue (~20-40 us) and the stage costs ~400 us;
// nothing in a block needs another stream's rows, so a workgroup that owns S streams x the stage's frames can run the whole
// dependent chain itself: 3 launches per stage, no intermediate in HBM, the LayerNorms / SiLU / GLU / residuals / depthwise
// conv / BatchNorm between the contractions are register or LDS epilogues.  The bound becomes the per-CU L2 fetch rate of the
// weight stream (4 bytes per weight in the exact-f32 and the split 16-bit modes, 2 in plain bf16: ~3 MB per half block).
// Weights are packed at finalize so that one wave instruction reads 1 KiB contiguous = one B fragment of a 16 x 32 tile
// (no LDS staging: no two waves of a workgroup share a fragment), with one tile (16 KiB per wave, 128 KiB per CU) in flight
// ahead of the MFMAs, across epilogues and barriers.
#pragma once

#define FUSE_MAXC 4      // chunks of one layer a workgroup tile may span
#define FUSE_ROWS 48     // rows per workgroup tile (3 MFMA row tiles)
#define FUSE_MAXF 16     // frames of one stream in a tile (depthwise-conv window registers)

struct LayerDev {        // one per encoder layer, device memory (built at finalize for the context's numerics mode)
    const uint4 *w1m, *w2m, *wq, *wk, *wv, *wo, *pw1, *pw2, *w1, *w2;   // fragment-major packed weights
    const float *b1m, *b2m, *bq, *bk, *bv, *bo, *bpw1, *bpw2, *b1, *b2;
    const float *ln_ffm_g, *ln_ffm_b, *ln_mha_g, *ln_mha_b, *ln_conv_g, *ln_conv_b, *ln_ff_g, *ln_ff_b, *ln_fin_g, *ln_fin_b;
    const float *wdw_t, *bdw, *bn_s, *bn_t;
    float *kc, *vc, *gr, *xr;          // K / V caches [B][tcap][256], post-GLU ring and conv-input ring [B][cap][256]
};
struct FuseItem {        // one (layer, run of <= FUSE_MAXC consecutive chunks) of a wavefront stage
    int layer, n_chunks, nf, pad;
    int tq[FUSE_MAXC], f0[FUSE_MAXC];          // frames per stream of chunk j, first frame index of chunk j within the item
    long long xrow[FUSE_MAXC];                 // x row of (stream 0, frame 0) of chunk j; row(b, f) = xrow + b * tq + f
    int kvrow[FUSE_MAXC];                      // cache row of chunk j's first new frame
    int ringpos[FUSE_MAXC];                    // ring position of chunk j's first frame
    float* q[FUSE_MAXC];                       // per chunk [B * tq][256]: queries out (front) / attention output in (back)
    float* a[FUSE_MAXC];
};

// ---- packing -----------------------------------------------------------------------------------------------------------
// W [N][ldw] (rows n0.., K columns) -> fragment-major: unit (nt, kt) = 32 k of 16 rows.
//   16-bit modes: [nt][kt][plane][lane][8 x 16 bit], lane (i = l & 15, q = l >> 4) holds W[16 nt + i][32 kt + 8 q + 0..7]
//   exact f32:    [nt][kt][half][lane][4 x f32],     lane (i, kq)                holds W[16 nt + i][32 kt + 16 half + 4 kq + 0..3]
// Rows >= N are zero.  One thread per (nt, kt, lane).
template <int NUM>
__global__ void pack_frag(const float* __restrict__ W, int N, int K, int ldw, uint4* __restrict__ dst) {
    const int KT = K / 32, NTn = (N + 15) / 16;
    const long long total = (long long)NTn * KT * 64;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        const long long u = e >> 6;
        const int kt = (int)(u % KT), nt = (int)(u / KT);
        const int i = lane & 15, q = lane >> 4;
        const int n = 16 * nt + i;
        if constexpr (NUM == RNNT_NUM_F32) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < N) v = ldg4(W + (long long)n * ldw + 32 * kt + 16 * h + 4 * q);
                dst[(u * 2 + h) * 64 + lane] = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w));
            }
        } else {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (n < N) { a = ldg4(W + (long long)n * ldw + 32 * kt + 8 * q); b = ldg4(W + (long long)n * ldw + 32 * kt + 8 * q + 4); }
            uint4 hi, lo;
            split8_16<NUM == RNNT_NUM_F16X3, true>(a, b, hi, lo);
            if constexpr (NUM == RNNT_NUM_BF16) dst[u * 64 + lane] = hi;
            else { dst[(u * 2) * 64 + lane] = hi; dst[(u * 2 + 1) * 64 + lane] = lo; }
        }
    }
}

// ---- building blocks ---------------------------------------------------------------------------------------------------------
template <int NUM> struct FuseCfg {
    static constexpr bool F32 = NUM == RNNT_NUM_F32;
    static constexpr bool F16 = NUM == RNNT_NUM_F16X3;
    static constexpr int U = NUM == RNNT_NUM_BF16 ? 1 : 2;          // 16-byte vectors per lane and 32 k of a fragment stream
    static constexpr int ROWB = F32 ? 1024 : 512;                   // bytes per operand row and plane
    static constexpr int PLANES = (NUM == RNNT_NUM_BF16X3 || NUM == RNNT_NUM_F16X3) ? 2 : 1;
    static constexpr int OPB = ROWB * PLANES;                       // operand image bytes per row
};

// pipeline unit of the weight stream: KSU k-steps (32 k each) of one n-tile (16 output columns) = 8 vectors per lane = 8 KiB
// per wave (4 k-steps of the two-vector streams, 8 of plain bf16)
template <int NUM> struct BTile { uint4 v[8]; };
template <int NUM> constexpr int fuse_ksu() { return 8 / FuseCfg<NUM>::U; }

template <int NUM>
__device__ __forceinline__ void btile_load(BTile<NUM>& t, const uint4* __restrict__ Wp, int nt, int KT, int kt0, int lane) {
    constexpr int U = FuseCfg<NUM>::U;
    const uint4* p = Wp + ((long long)(nt * KT + kt0) * U) * 64 + lane;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef unsigned u32x4g_ __attribute__((ext_vector_type(4)));
        const u32x4g_ v = *(const RNNT_GAS u32x4g_*)(p + e * 64);
        t.v[e] = make_uint4(v[0], v[1], v[2], v[3]);
#else
        t.v[e] = p[e * 64];
#endif
    }
}

// operand image in LDS: row r, 16-byte chunk c at byte  plane * R * ROWB + r * ROWB + ((c ^ (r & 15)) << 4): the 16 lanes one
// ds_read_b128 cycle serves ({0-3,12-15,20-27}, ... of fragment lane (i, q) = row i, chunk 4 ks + q) land on 16 distinct slots.
template <int NUM>
__device__ __forceinline__ int op_off(int r, int c) { return r * FuseCfg<NUM>::ROWB + ((c ^ (r & 15)) << 4); }

// acc[mt] += A[16 mt .. +15][0..255] * tile^T   (A = operand image `op` of R rows).  The scheduling barriers keep the A fragments
// of one k-step live at a time: left alone the scheduler hoists every ds_read of the tile and spills hundreds of registers.
template <int NUM, int MT, int KS0>
__device__ __forceinline__ void mma_tile(f32x4_ (&acc)[MT], const BTile<NUM>& t, const unsigned char* op, int R, int lane) {
    using C = FuseCfg<NUM>;
    const int i = lane & 15, q = lane >> 4;
    const unsigned char* rowp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rowp[mt] = op + (16 * mt + i) * C::ROWB;
#pragma unroll
    for (int ks = KS0; ks < KS0 + fuse_ksu<NUM>(); ++ks) {
        if constexpr (C::F32) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint4 bw = t.v[2 * (ks - KS0) + h];
                float4 a[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(rowp[mt] + (((8 * ks + 4 * h + q) ^ i) << 4));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, __uint_as_float(bw.x), acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, __uint_as_float(bw.y), acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, __uint_as_float(bw.z), acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, __uint_as_float(bw.w), acc[mt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            uint4 ah[MT], al[C::PLANES == 2 ? MT : 1];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int off = ((4 * ks + q) ^ i) << 4;
                ah[mt] = *reinterpret_cast<const uint4*>(rowp[mt] + off);
                if constexpr (C::PLANES == 2) al[mt] = *reinterpret_cast<const uint4*>(rowp[mt] + R * C::ROWB + off);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (C::PLANES == 2) {
                    acc[mt] = mfma16_<C::F16>(al[mt], t.v[2 * (ks - KS0)], acc[mt]);
                    acc[mt] = mfma16_<C::F16>(ah[mt], t.v[2 * (ks - KS0) + 1], acc[mt]);
                    acc[mt] = mfma16_<C::F16>(ah[mt], t.v[2 * (ks - KS0)], acc[mt]);
                } else {
                    acc[mt] = mfma16_<false>(ah[mt], t.v[ks - KS0], acc[mt]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// store one value / four consecutive-k values of row r into the operand image
template <int NUM>
__device__ __forceinline__ void op_store1(unsigned char* op, int R, int r, int col, float v) {
    using C = FuseCfg<NUM>;
    if constexpr (C::F32) {
        *reinterpret_cast<float*>(op + op_off<NUM>(r, col >> 2) + (col & 3) * 4) = v;
    } else {
        float rx, ry, d0, d1;
        const unsigned h = pack2_16<C::F16>(v, 0.f, rx, ry);
        const int off = op_off<NUM>(r, col >> 3) + (col & 7) * 2;
        *reinterpret_cast<unsigned short*>(op + off) = (unsigned short)(h & 0xffffu);
        if constexpr (C::PLANES == 2) {
            const unsigned l = pack2_16<C::F16>(rx, 0.f, d0, d1);
            *reinterpret_cast<unsigned short*>(op + R * C::ROWB + off) = (unsigned short)(l & 0xffffu);
        }
    }
}
template <int NUM>
__device__ __forceinline__ void op_store4(unsigned char* op, int R, int r, int col4, const float4& v) {   // col4 % 4 == 0
    using C = FuseCfg<NUM>;
    if constexpr (C::F32) {
        *reinterpret_cast<float4*>(op + op_off<NUM>(r, col4 >> 2)) = v;
    } else {
        float r0, r1, r2, r3, d0, d1;
        uint2 h, l;
        h.x = pack2_16<C::F16>(v.x, v.y, r0, r1);
        h.y = pack2_16<C::F16>(v.z, v.w, r2, r3);
        const int off = op_off<NUM>(r, col4 >> 3) + (col4 & 4) * 2;
        *reinterpret_cast<uint2*>(op + off) = h;
        if constexpr (C::PLANES == 2) {
            l.x = pack2_16<C::F16>(r0, r1, d0, d1);
            l.y = pack2_16<C::F16>(r2, r3, d0, d1);
            *reinterpret_cast<uint2*>(op + R * C::ROWB + off) = l;
        }
    }
}

// LayerNorm of the R rows of X (LDS, f32 [R][256]) into the operand image; one wave per row, two-pass statistics
template <int NUM>
__device__ __forceinline__ void ln_to_op(const float* X, unsigned char* op, int R, const float* __restrict__ g, const float* __restrict__ b,
                                         int wave, int lane, int nwaves) {
    const float4 gg = ldg4(g + 4 * lane), bb = ldg4(b + 4 * lane);
    for (int r = wave; r < R; r += nwaves) {
        const float4 v = *reinterpret_cast<const float4*>(X + r * RNNT_D + 4 * lane);
        const float mu = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
        const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
        const float rstd = 1.0f / sqrtf(wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / 256.0f) + 1e-5f);
        op_store4<NUM>(op, R, r, 4 * lane, make_float4(dx * rstd * gg.x + bb.x, dy * rstd * gg.y + bb.y, dz * rstd * gg.z + bb.z, dw * rstd * gg.w + bb.w));
    }
}

// per-row bookkeeping of a workgroup tile
struct FuseRows {
    long long xoff[FUSE_ROWS];     // float offset of the row in x (-1: padding row)
    int b[FUSE_ROWS], j[FUSE_ROWS], f[FUSE_ROWS];   // global stream, chunk index within the item, frame within the chunk
};

// shared state of both kernels ------------------------------------------------------------------------------------------------
#define FUSE_THREADS 512
#define FUSE_WAVES 8

// One dense 256 -> 256 block: acc[mt][t] (t = this wave's two column tiles 2 w, 2 w + 1 of the 16) over the operand image.
// The weight stream moves in units of 8 KiB per wave (BTile): on entry b0 holds this wave's first unit; every further unit is
// fetched into the buffer the previous multiply has just released, and the last fetch is the first unit of the NEXT block
// (nWp .. nkt0), so one unit per wave (64 KiB per CU) is always in flight -- across epilogues and barriers.
template <int NUM, int MT>
__device__ __forceinline__ void dense256(f32x4_ (&acc)[MT][2], BTile<NUM>& b0, BTile<NUM>& b1, const unsigned char* op, int R,
                                         const uint4* Wp, int nt0, int KT, int kt0,
                                         const uint4* nWp, int nnt0, int nKT, int nkt0, int wave, int lane) {
    constexpr int KSU = fuse_ksu<NUM>();
    const int ntA = nt0 + 2 * wave, ntB = ntA + 1;
    f32x4_ a[MT];
    if constexpr (KSU == 8) {                                   // one unit per n-tile
        btile_load<NUM>(b1, Wp, ntB, KT, kt0, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = acc[mt][0];
        mma_tile<NUM, MT, 0>(a, b0, op, R, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][0] = a[mt];
        if (nWp) btile_load<NUM>(b0, nWp, nnt0 + 2 * wave, nKT, nkt0, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = acc[mt][1];
        mma_tile<NUM, MT, 0>(a, b1, op, R, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][1] = a[mt];
    } else {                                                    // two units per n-tile
        btile_load<NUM>(b1, Wp, ntA, KT, kt0 + 4, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = acc[mt][0];
        mma_tile<NUM, MT, 0>(a, b0, op, R, lane);
        btile_load<NUM>(b0, Wp, ntB, KT, kt0, lane);
        mma_tile<NUM, MT, 4>(a, b1, op, R, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][0] = a[mt];
        btile_load<NUM>(b1, Wp, ntB, KT, kt0 + 4, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = acc[mt][1];
        mma_tile<NUM, MT, 0>(a, b0, op, R, lane);
        if (nWp) btile_load<NUM>(b0, nWp, nnt0 + 2 * wave, nKT, nkt0, lane);
        mma_tile<NUM, MT, 4>(a, b1, op, R, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][1] = a[mt];
    }
}

template <int MT>
__device__ __forceinline__ void acc_zero(f32x4_ (&acc)[MT][2]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f32x4_){0.f, 0.f, 0.f, 0.f}; acc[mt][1] = (f32x4_){0.f, 0.f, 0.f, 0.f}; }
}

// FFN of the rows in X (x += 1/2 W2 silu(W1 LN(x) + b1) + b2, positionwise_feed_forward.py:50-58; encoder_layer.py:216-223,250-255):
// four 256-wide slabs of the hidden layer; the slab's activations become the operand image `hop` of the second product, whose
// accumulators stay in registers over the four slabs.  Entry: b0 holds this wave's first tile of W1 slab 0; exit: b0 holds the
// first tile of (nWp, nnt0, nKT, nkt0).  Barriers inside; X must be complete and visible on entry (caller's barrier).
template <int NUM, int MT>
__device__ __forceinline__ void ffn_block(float* X, unsigned char* op, unsigned char* hop, int R, const uint4* W1, const float* __restrict__ b1,
                                          const uint4* W2, const float* __restrict__ b2, const float* __restrict__ lng, const float* __restrict__ lnb,
                                          BTile<NUM>& b0, BTile<NUM>& b1t, const uint4* nWp, int nnt0, int nKT, int nkt0, int wave, int lane) {
    const int i = lane & 15, kq = lane >> 4;
    ln_to_op<NUM>(X, op, R, lng, lnb, wave, lane, FUSE_WAVES);
    __syncthreads();
    f32x4_ y[MT][2];
    acc_zero<MT>(y);
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        f32x4_ h[MT][2];
        acc_zero<MT>(h);
        dense256<NUM, MT>(h, b0, b1t, op, R, W1, 16 * s, 8, 0, W2, 0, 32, 8 * s, wave, lane);
        if (s > 0) __syncthreads();                       // every wave has finished reading the previous slab's image
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = 32 * wave + 16 * t + i;         // column inside the slab
            const float bias = ldg1(b1 + 256 * s + n);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = h[mt][t][r] + bias;
                    v = v * sigmoidf_(v);
                    op_store1<NUM>(hop, R, 16 * mt + 4 * kq + r, n, v);
                }
        }
        __syncthreads();
        if (s < 3) dense256<NUM, MT>(y, b0, b1t, hop, R, W2, 0, 32, 8 * s, W1, 16 * (s + 1), 8, 0, wave, lane);
        else dense256<NUM, MT>(y, b0, b1t, hop, R, W2, 0, 32, 8 * s, nWp, nnt0, nKT, nkt0, wave, lane);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 32 * wave + 16 * t + i;
        const float bias = ldg1(b2 + n);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) X[(16 * mt + 4 * kq + r) * RNNT_D + n] += 0.5f * (y[mt][t][r] + bias);
    }
    __syncthreads();
}

__device__ __forceinline__ void fuse_rows_init(FuseRows& rw, const FuseItem& it, int grp, int S, int B, int R) {
    for (int r = threadIdx.x; r < R; r += FUSE_THREADS) {
        const int sl = r / it.nf, fi = r - sl * it.nf;
        const int b = grp * S + sl;
        int j = 0;
#pragma unroll
        for (int c = 1; c < FUSE_MAXC; ++c)
            if (c < it.n_chunks && fi >= it.f0[c]) j = c;
        const int f = fi - it.f0[j];
        const bool ok = sl < S && b < B;
        rw.b[r] = ok ? b : -1;
        rw.j[r] = j;
        rw.f[r] = f;
        rw.xoff[r] = ok ? (it.xrow[j] + (long long)b * it.tq[j] + f) * RNNT_D : -1;
    }
}

// ------------------------------------------------------------------------------------------------
// block_front<NUM, MT>: grid = ceil(n_items / 8) * 8 * n_groups workgroups of 512 threads; workgroup = (item, stream group).
// XCD-aware: the groups of one item (one layer's weights) run on one XCD (block id % 8), so the weight stream of a layer is
// pulled into ONE L2 and the other groups hit it.  Placement is a speed hint only.
// ------------------------------------------------------------------------------------------------
template <int NUM, int MT>
__global__ __launch_bounds__(FUSE_THREADS) void block_front(const LayerDev* __restrict__ layers, const FuseItem* __restrict__ items, float* __restrict__ x,
                                                            int n_items, int n_groups, int S, int B, long long tcap) {
    using C = FuseCfg<NUM>;
    constexpr int R = 16 * MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    float* X = reinterpret_cast<float*>(fsm);                       // [R][256] f32 residual stream
    unsigned char* op = fsm + R * 1024;                             // operand image
    unsigned char* hop = op + R * C::OPB;                           // FFN hidden slab image
    FuseRows& rw = *reinterpret_cast<FuseRows*>(hop + R * C::OPB);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int item = (slot / n_groups) * 8 + xcd, grp = slot % n_groups;
    if (item >= n_items) return;
    const FuseItem& it = items[item];
    const LayerDev& L = layers[it.layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    BTile<NUM> b0, b1;
    btile_load<NUM>(b0, L.w1m, 2 * wave, 8, 0, lane);               // the weight stream starts before anything else
    fuse_rows_init(rw, it, grp, S, B, R);
    __syncthreads();
    for (int e = tid; e < R * 64; e += FUSE_THREADS) {
        const int r = e >> 6, c4 = (e & 63) * 4;
        const long long xo = rw.xoff[r];
        *reinterpret_cast<float4*>(X + r * RNNT_D + c4) = xo >= 0 ? ldg4(x + xo + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // x += 1/2 FFN_macaron(LN(x))
    ffn_block<NUM, MT>(X, op, hop, R, L.w1m, L.b1m, L.w2m, L.b2m, L.ln_ffm_g, L.ln_ffm_b, b0, b1, L.wq, 0, 8, 0, wave, lane);
    // q, k, v = linear(LN(x)) (attention.py:109-131); k, v appended to the cache rows of their frames (attention.py:207-211)
    ln_to_op<NUM>(X, op, R, L.ln_mha_g, L.ln_mha_b, wave, lane, FUSE_WAVES);
    for (int e = tid; e < R * 64; e += FUSE_THREADS) {              // the updated residual stream goes back to HBM meanwhile
        const int r = e >> 6, c4 = (e & 63) * 4;
        const long long xo = rw.xoff[r];
        if (xo >= 0) stg4(x + xo + c4, *reinterpret_cast<const float4*>(X + r * RNNT_D + c4));
    }
    __syncthreads();
#pragma unroll 1
    for (int m = 0; m < 3; ++m) {
        f32x4_ acc[MT][2];
        acc_zero<MT>(acc);
        const uint4* Wp = m == 0 ? L.wq : (m == 1 ? L.wk : L.wv);
        const uint4* nW = m == 0 ? L.wk : (m == 1 ? L.wv : nullptr);
        dense256<NUM, MT>(acc, b0, b1, op, R, Wp, 0, 8, 0, nW, 0, 8, 0, wave, lane);
        const float* bias = m == 0 ? L.bq : (m == 1 ? L.bk : L.bv);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = 32 * wave + 16 * t + i;
            const float bs = ldg1(bias + n);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int r = 16 * mt + 4 * kq + r4;
                    const int b = rw.b[r];
                    if (b < 0) continue;
                    const int j = rw.j[r], f = rw.f[r];
                    const float v = acc[mt][t][r4] + bs;
                    if (m == 0) stg1(it.q[j] + ((long long)b * it.tq[j] + f) * RNNT_D + n, v);
                    else stg1((m == 1 ? L.kc : L.vc) + ((long long)b * tcap + it.kvrow[j] + f) * RNNT_D + n, v);
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// block_back<NUM, MT>: same grid.  `cap` = ring capacity in frames.
// ------------------------------------------------------------------------------------------------
template <int NUM, int MT>
__global__ __launch_bounds__(FUSE_THREADS) void block_back(const LayerDev* __restrict__ layers, const FuseItem* __restrict__ items, float* __restrict__ x,
                                                           int n_items, int n_groups, int S, int B, int cap) {
    using C = FuseCfg<NUM>;
    constexpr int R = 16 * MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    float* X = reinterpret_cast<float*>(fsm);
    unsigned char* op = fsm + R * 1024;
    unsigned char* hop = op + R * C::OPB;                           // FFN hidden slab image / post-GLU frames (f32 [R][256])
    FuseRows& rw = *reinterpret_cast<FuseRows*>(hop + R * 1024);
    float* G = reinterpret_cast<float*>(hop);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int item = (slot / n_groups) * 8 + xcd, grp = slot % n_groups;
    if (item >= n_items) return;
    const FuseItem& it = items[item];
    const LayerDev& L = layers[it.layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    BTile<NUM> b0, b1;
    btile_load<NUM>(b0, L.wo, 2 * wave, 8, 0, lane);
    fuse_rows_init(rw, it, grp, S, B, R);
    __syncthreads();
    for (int e = tid; e < R * 64; e += FUSE_THREADS) {
        const int r = e >> 6, c4 = (e & 63) * 4;
        const long long xo = rw.xoff[r];
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), av = xv;
        if (xo >= 0) {
            xv = ldg4(x + xo + c4);
            av = ldg4(it.a[rw.j[r]] + ((long long)rw.b[r] * it.tq[rw.j[r]] + rw.f[r]) * RNNT_D + c4);
        }
        *reinterpret_cast<float4*>(X + r * RNNT_D + c4) = xv;
        op_store4<NUM>(op, R, r, c4, av);
    }
    __syncthreads();
    // x += linear_out(att) (attention.py:178)
    {
        f32x4_ acc[MT][2];
        acc_zero<MT>(acc);
        dense256<NUM, MT>(acc, b0, b1, op, R, L.wo, 0, 8, 0, L.pw1, 0, 8, 0, wave, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = 32 * wave + 16 * t + i;
            const float bs = ldg1(L.bo + n);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[(16 * mt + 4 * kq + r) * RNNT_D + n] += acc[mt][t][r] + bs;
        }
    }
    __syncthreads();
    // conv module (convolution.py:98-153): the conv-input ring keeps the pre-LayerNorm rows (streaming_cnn_cache view)
    for (int e = tid; e < R * 64; e += FUSE_THREADS) {
        const int r = e >> 6, c4 = (e & 63) * 4;
        const int b = rw.b[r];
        if (b < 0) continue;
        const int pos = (it.ringpos[rw.j[r]] + rw.f[r]) % cap;
        stg4(L.xr + ((long long)b * cap + pos) * RNNT_D + c4, *reinterpret_cast<const float4*>(X + r * RNNT_D + c4));
    }
    ln_to_op<NUM>(X, op, R, L.ln_conv_g, L.ln_conv_b, wave, lane, FUSE_WAVES);
    __syncthreads();
    // pointwise_conv1 + GLU: packed rows are (value, gate) interleaved, so column pairs (2 c, 2 c + 1) -> G[:, c]
#pragma unroll 1
    for (int hb = 0; hb < 2; ++hb) {
        f32x4_ acc[MT][2];
        acc_zero<MT>(acc);
        if (hb == 0) dense256<NUM, MT>(acc, b0, b1, op, R, L.pw1, 0, 8, 0, L.pw1, 16, 8, 0, wave, lane);
        else dense256<NUM, MT>(acc, b0, b1, op, R, L.pw1, 16, 8, 0, L.pw2, 0, 8, 0, wave, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = 256 * hb + 32 * wave + 16 * t + i;        // interleaved column
            const float bs = ldg1(L.bpw1 + n);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float v = acc[mt][t][r4] + bs;
                    const float g = __shfl_xor(v, 1, 64);
                    if (!(i & 1)) {
                        const int r = 16 * mt + 4 * kq + r4;
                        const float o = v * sigmoidf_(g);
                        G[r * RNNT_D + (n >> 1)] = o;
                        const int b = rw.b[r];
                        if (b >= 0) stg1(L.gr + ((long long)b * cap + (it.ringpos[rw.j[r]] + rw.f[r]) % cap) * RNNT_D + (n >> 1), o);
                    }
                }
        }
    }
    __syncthreads();
    // depthwise conv k = 31 (causal, 30 frames of left context from the ring) + BatchNorm(eval) + SiLU -> operand image of
    // pointwise_conv2.  One thread per (stream, channel): the 30 context values are read once and slide over the frames.
    {
        const int nf = it.nf;
        const int pos0 = it.ringpos[0];
        for (int e = tid; e < S * RNNT_D; e += FUSE_THREADS) {
            const int sl = e >> 8, c = e & 255;
            const int r0 = sl * nf;
            if (r0 >= R) continue;
            const int b = rw.b[r0];
            float win[RNNT_LORDER + FUSE_MAXF];
            if (b >= 0) {
                const float* gb = L.gr + (long long)b * cap * RNNT_D + c;
                int ridx = (pos0 - RNNT_LORDER + cap * 64) % cap;
#pragma unroll
                for (int k = 0; k < RNNT_LORDER; ++k) {
                    win[k] = ldg1(gb + (long long)ridx * RNNT_D);
                    ridx = ridx + 1 == cap ? 0 : ridx + 1;
                }
            } else {
#pragma unroll
                for (int k = 0; k < RNNT_LORDER; ++k) win[k] = 0.f;
            }
            float wk[RNNT_KDW];
#pragma unroll
            for (int k = 0; k < RNNT_KDW; ++k) wk[k] = ldg1(L.wdw_t + k * RNNT_D + c);
            const float bd = ldg1(L.bdw + c), bs = ldg1(L.bn_s + c), bt = ldg1(L.bn_t + c);
#pragma unroll
            for (int f = 0; f < FUSE_MAXF; ++f) {
                if (f < nf && r0 + f < R) {
                    win[RNNT_LORDER + f] = G[(r0 + f) * RNNT_D + c];
                    float acc = bd;
#pragma unroll
                    for (int k = 0; k < RNNT_KDW; ++k) acc = fmaf(wk[k], win[f + k], acc);
                    float v = acc * bs + bt;
                    v = v * sigmoidf_(v);
                    op_store1<NUM>(op, R, r0 + f, c, v);
                }
            }
        }
        // rows of absent streams / padding: the image keeps the LayerNorm rows there (finite), their results are never stored
    }
    __syncthreads();
    {
        f32x4_ acc[MT][2];
        acc_zero<MT>(acc);
        dense256<NUM, MT>(acc, b0, b1, op, R, L.pw2, 0, 8, 0, L.w1, 0, 8, 0, wave, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = 32 * wave + 16 * t + i;
            const float bs = ldg1(L.bpw2 + n);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[(16 * mt + 4 * kq + r) * RNNT_D + n] += acc[mt][t][r] + bs;
        }
    }
    __syncthreads();
    // x += 1/2 FFN(LN(x)); x = LN_final(x)
    ffn_block<NUM, MT>(X, op, hop, R, L.w1, L.b1, L.w2, L.b2, L.ln_ff_g, L.ln_ff_b, b0, b1, nullptr, 0, 8, 0, wave, lane);
    {
        const float4 gg = ldg4(L.ln_fin_g + 4 * lane), bb = ldg4(L.ln_fin_b + 4 * lane);
        for (int r = wave; r < R; r += FUSE_WAVES) {
            const long long xo = rw.xoff[r];
            if (xo < 0) continue;
            const float4 v = *reinterpret_cast<const float4*>(X + r * RNNT_D + 4 * lane);
            const float mu = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
            const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
            const float rstd = 1.0f / sqrtf(wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / 256.0f) + 1e-5f);
            stg4(x + xo + 4 * lane, make_float4(dx * rstd * gg.x + bb.x, dy * rstd * gg.y + bb.y, dz * rstd * gg.z + bb.z, dw * rstd * gg.w + bb.w));
        }
    }
}
