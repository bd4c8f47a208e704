// EXPERIMENT, not part of the library: a vocabulary-tile-outer form of the joint lattice kernel, kept with its harness
// (tools/jl_check.hip, JL_STREAM=1) because its ablations are the evidence DESIGN.md cites for why joint_lattice_rows keeps the
// row-owner form.  Measured on MI355X, B64 x T249 x U28 x V412 (us, bf16x3 / bf16):
//     logits               378 / 338      (joint_lattice_rows: 377 / 263)
//     log-softmax          805 / 765      the un-normalised tile is re-read and re-written one row tile later: 3 x the HBM traffic
//     logits, no stores    276 / 172      (JS_ABLATE=1)
//     + no W DMA           267 / 162      (JS_ABLATE=3)
//     + no stage barrier   182 / 106      (JS_ABLATE=11; the 3-MFMA floor of the split mode is 165 us)
// i.e. 16 waves per CU with 4-register accumulators do reach the MFMA floor, but (a) a 16-wave barrier per 16-column stage costs
// 55-85 us, (b) 64-byte-per-row column-slice stores (one DRAM page per row and stage) do not overlap with compute the way the
// row-owner kernel's 26-KiB contiguous bursts do, and (c) log-softmax needs the whole row before the first store.
#pragma once
// ================================================================================================================================
// joint_lattice_stream<NSPLIT,F16,LSM>: the lattice in VOCABULARY-TILE-OUTER order (round 3, second form).
//
// joint_lattice_rows keeps a wave's 16 rows x 416 columns in 104 accumulator registers until the row's log-sum-exp is known: two
// waves per SIMD, burst stores, W_out re-streamed per 64 rows.  Here a wave finishes ONE 16 x 16 tile over all of K (8 k-steps x 3
// products = 24 MFMAs into 4 registers), stores it at once, and only carries the row's running (max, sum):
//   * stores are progressive: one 16-byte-per-lane store per stage and wave, no burst, nothing to hide;
//   * 4 accumulator registers instead of 104: the kernel fits 128 VGPRs, so a workgroup is 16 waves (4 per SIMD) = 256 lattice rows
//     per pass over W_out -- a quarter of the LDS-DMA issue per row, and four waves per SIMD to cover each other's waits;
//   * log-softmax: the tile is stored UN-normalised; one row tile later (when its log-sum-exp is known) every wave re-reads its
//     own values (L2 / Infinity Cache: they were written ~25 us earlier), subtracts and re-stores them, one tile per stage,
//     interleaved with the next row tile's work, so the fix-up is as progressive as the first store.
// W_out streams through a 6-slot LDS ring, stage = one vocabulary tile = 8 k-steps x planes (16 / 8 KiB), fetched 4 stages ahead;
// every wave issues one 1-KiB piece per stage.  All vector-memory operations of the loop are inline asm with hand-counted waits
// (s_waitcnt vmcnt counts loads, stores and LDS-DMA together, in issue order): each stage issues the same operations in the
// same order -- D (DMA piece), S (tile store), then with a previous row tile F (fix-up store) and L (fix-up load, 3 stages ahead)
// -- and rows beyond M store into a dump buffer, so the counts are the same for every wave of a kind.
// ================================================================================================================================
#define JS_ROWS 256
#define JS_NSLOT 6
#define JS_PF 4                  // stages the DMA runs ahead
#ifndef JS_ABLATE
#define JS_ABLATE 0             // measurement only: 1 no tile stores, 2 no W DMA, 8 no per-stage barrier
#endif
#ifndef JS_RB
#define JS_RB 4                  // W pieces per ds_read batch (two batches in flight)
#endif
#define JS_LA 3                  // stages the fix-up load runs ahead of its use
#define JS_SLOT(NSPLIT) (8 * ((NSPLIT) == 2 ? 2 : 1) * 1024)
#define JS_FIX_BYTES (16 * JS_LA * 1024)      // per wave JS_LA 1-KiB landing slots of the fix-up loads
#define JS_LDS_ALLOC(NSPLIT) (JS_NSLOT * JS_SLOT(NSPLIT) + JS_FIX_BYTES + JR_NT * 16 * 4 + 16)

struct JointSP {
    const float* e;              // [B*T][256]  JR_PRESCALE * joint.enc_ffn(enc)
    const float* p;              // [B*U][256]  JR_PRESCALE * joint.pred_ffn(pred)
    const unsigned char* wfrag;  // pack_joint_w_stream: [ntv stages][8 k-steps][planes][64 lanes][8 x 16 bit]
    const float* bias;           // [V]
    float* out;                  // [M][V]
    float* dump;                 // [256][V]: where rows >= M store (every wave issues the same operations)
    long long M;
    int T, U, V;
    int ntiles;                  // ceil(M / 256)
    int ntv;                     // vocabulary tiles = ceil(V / 16), >= 6
    int* counter;                // dynamic row-tile queue (zeroed by the host before every launch)
};

// W_out [V][256] f32 -> the stage stream of joint_lattice_stream: piece (t, s, pl) holds for lane l the 8 k-consecutive 16-bit values
// W[16 t + (l & 15)][32 s + 8 (l >> 4) + 0..7] of plane pl.  Rows >= V are zero.
template <bool F16, bool LO>
__global__ void pack_joint_w_stream(const float* __restrict__ w, int V, unsigned char* __restrict__ dst) {
    constexpr int PL = LO ? 2 : 1;
    const int ntv = (V + 15) / 16;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // ((t * 8 + s) * PL + pl) * 64 + lane
    if (idx >= ntv * 8 * PL * 64) return;
    const int lane = idx & 63, pj = idx >> 6, pl = pj % PL, s = (pj / PL) & 7, t = pj / (PL * 8);
    const int row = 16 * t + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (row < V) { a = ldg4(w + (long long)row * RNNT_D + k0); b = ldg4(w + (long long)row * RNNT_D + k0 + 4); }
    uint4 h, l;
    split8_16<F16, true>(a, b, h, l);
    *reinterpret_cast<uint4*>(dst + (long long)idx * 16) = pl ? l : h;
}

// Vector-memory operations of the stream kernel.  Stores are plain C++ (hipcc never waits on them).  The fix-up loads must run
// three stages ahead with a counted wait, and a load that lands in a REGISTER cannot be hidden from hipcc (an asm load whose
// destination it may copy before the wait -- seen -- or a plain load whose wait it places itself: vmcnt(1..3), a stall on the DMA
// just issued -- seen).  So they are LDS-DMAs too: per-lane global address -> the wave's own 1-KiB landing slot, read back with a
// plain ds_read after the counted wait.  The asm memory clobbers keep every operation in its program position.
__device__ __forceinline__ void js_store16(float* p, const f32x4_& v) { *reinterpret_cast<f32x4_*>(p) = v; }
__device__ __forceinline__ void js_dma_lane(const float* p, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(p), "s"(lds_addr) : "memory");
}
#define JS_WAIT(N_) asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory")
__device__ __forceinline__ void js_vmcnt15() { asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); }
__device__ __forceinline__ void js_vmcnt7() { asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }

template <int NSPLIT, bool F16, bool LSM>
__global__ __launch_bounds__(1024) void joint_lattice_stream(JointSP P) {
    constexpr bool LO = NSPLIT == 2;
    constexpr int PL = LO ? 2 : 1, PPS = 8 * PL, SLOTB = PPS * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char js_smem[];
    float* biasl = reinterpret_cast<float*>(js_smem + JS_NSLOT * SLOTB + JS_FIX_BYTES);
    int* const nxt_lds = reinterpret_cast<int*>(js_smem + JS_NSLOT * SLOTB + JS_FIX_BYTES + JR_NT * 16 * 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const bool hasD = wave < PPS;                                  // this wave issues one DMA piece per stage
    const int ntv = P.ntv;
    const unsigned voff = lane * 16;
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds0 = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)js_smem);
#else
    const unsigned lds0 = 0;
#endif
    int tile = blockIdx.x;
    if (tile >= P.ntiles) return;                                  // uniform
    // ---- W ring: the stage stream wraps every ntv stages; fetch position and slot run JS_PF stages ahead of the compute position ----
    int f_stage = 0, f_slot = 0, c_slot = 0;
    auto fetch = [&]() {
        if (hasD && !(JS_ABLATE & 2)) jr_dma1(P.wfrag + ((long long)f_stage * SLOTB + 1024 * wave), voff, lds0 + f_slot * SLOTB + 1024 * wave);
        f_stage = f_stage + 1 == ntv ? 0 : f_stage + 1;
        f_slot = f_slot + 1 == JS_NSLOT ? 0 : f_slot + 1;
    };
#pragma unroll
    for (int k = 0; k < JS_PF; ++k) fetch();
    for (int v = tid; v < JR_NT * 16; v += 1024) biasl[v] = v < P.V ? ldg1(P.bias + v) : -INFINITY;
    uint4 ah[8], al[LO ? 8 : 1];
    auto form_a = [&](int tl) {
        const int am = min(tl * JS_ROWS + wave * 16 + i, (int)P.M - 1);   // M < 2^31 (host check)
        const int bt = am / P.U;
        const int u = am - bt * P.U;
        const int bb = bt / P.T;
        const float* eg = P.e + (long long)bt * RNNT_D + 8 * kq;
        const float* pg = P.p + (long long)(bb * P.U + u) * RNNT_D + 8 * kq;
#pragma unroll
        for (int s0 = 0; s0 < 8; s0 += 2) {                        // two k-steps of loads at a time (32 registers: the budget is 128)
            float4 ld[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ld[q][0] = ldg4(eg + 32 * (s0 + q)); ld[q][1] = ldg4(eg + 32 * (s0 + q) + 4);
                ld[q][2] = ldg4(pg + 32 * (s0 + q)); ld[q][3] = ldg4(pg + 32 * (s0 + q) + 4);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 e0 = ld[q][0], e1 = ld[q][1], p0 = ld[q][2], p1 = ld[q][3];
                float4 v0, v1;
                v0.x = jr_tanh_pre(e0.x + p0.x); v0.y = jr_tanh_pre(e0.y + p0.y); v0.z = jr_tanh_pre(e0.z + p0.z); v0.w = jr_tanh_pre(e0.w + p0.w);
                v1.x = jr_tanh_pre(e1.x + p1.x); v1.y = jr_tanh_pre(e1.y + p1.y); v1.z = jr_tanh_pre(e1.z + p1.z); v1.w = jr_tanh_pre(e1.w + p1.w);
                uint4 h, l;
                split8_16<F16, LO>(v0, v1, h, l);
                ah[s0 + q] = h;
                if constexpr (LO) al[s0 + q] = l;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int popped = 0;
    if (wave == 0 && lane == 0) jr_queue_pop(P.counter, popped);
    form_a(tile);
    jr_vmcnt0();
    asm volatile("" : "+v"(popped));
    if (wave == 0 && lane == 0) nxt_lds[0] = (int)gridDim.x + popped;
    __syncthreads();                                               // the first JS_PF stages, the bias table and the second tile's index are in LDS
    int qi = 0;
    bool has_prev = false;
    float prev_lse = 0.f;
    float* prev_row = P.dump;
    const int fix0 = JS_NSLOT * SLOTB + wave * (JS_LA * 1024);     // this wave's landing slots of the fix-up loads (tile k -> slot k % JS_LA)
    int fx = 0;                                                    // landing slot of the tile fixed up in this stage
    const int rd = lane * 16;
    const int tfull = P.V >> 4;                                    // tiles below are whole; tile tfull holds (V & 15) / 4 quads (V % 4 == 0)
    const bool qin = kq < ((P.V & 15) >> 2);
    float* const dumpq = P.dump + (long long)(wave * 16 + i) * P.V + 4 * kq;   // where quads beyond the vocabulary go (same operation count)
    while (true) {
        const int next = __builtin_amdgcn_readfirstlane(nxt_lds[qi]);
        const bool more = next < P.ntiles;                         // uniform
        const long long m = (long long)tile * JS_ROWS + wave * 16 + i;
        float* row = (m < P.M ? P.out + m * P.V : P.dump + (long long)(wave * 16 + i) * P.V) + 4 * kq;
        float mx = -INFINITY, sm = 0.f;
        if (LSM && has_prev) {                                     // fix-up loads of the previous row tile's tiles 0, 1, 2
#pragma unroll
            for (int k = 0; k < JS_LA; ++k) js_dma_lane(prev_row + 16 * k, lds0 + fix0 + 1024 * k);
            fx = 0;
        }
        // one stage = one vocabulary tile
        auto stage = [&](int t) {
            if (t == 0 && more && wave == 0 && lane == 0) jr_queue_pop(P.counter, popped);   // the tile after next; published at stage 4
            fetch();                                               // D: stage + JS_PF into the slot freed two stages ago
            const unsigned char* sl = js_smem + c_slot * SLOTB + rd;
            f32x4_ acc = *reinterpret_cast<const f32x4_*>(&biasl[16 * t + 4 * kq]);
            // W pieces in batches of JS_RB, the next batch's ds_reads issued before this batch's MFMAs (hipcc on its own keeps
            // two reads in flight per wave: LDS latency, not MFMA rate, then sets the stage time)
            constexpr int NB = PPS / JS_RB;
            uint4 wb[2][JS_RB];
#pragma unroll
            for (int q = 0; q < JS_RB; ++q) wb[0][q] = *reinterpret_cast<const uint4*>(sl + q * 1024);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                __builtin_amdgcn_sched_barrier(0);
                if (b + 1 < NB) {
#pragma unroll
                    for (int q = 0; q < JS_RB; ++q) wb[(b + 1) & 1][q] = *reinterpret_cast<const uint4*>(sl + ((b + 1) * JS_RB + q) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < JS_RB; q += PL) {
                    const int s = (b * JS_RB + q) / PL;
                    if constexpr (LO) {
                        acc = mfma16_<F16>(wb[b & 1][q], al[s], acc);
                        acc = mfma16_<F16>(wb[b & 1][q + 1], ah[s], acc);
                    }
                    acc = mfma16_<F16>(wb[b & 1][q], ah[s], acc);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LSM) {                                   // running maximum / sum of this lane's quarter of the row
                const float L2E = 1.4426950408889634f;
                const float m4 = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
                const float mn = fmaxf(mx, m4), mnl = -mn * L2E;
                sm = sm * __builtin_amdgcn_exp2f(fmaf(mx, L2E, mnl)) + ((__builtin_amdgcn_exp2f(fmaf(acc[0], L2E, mnl)) + __builtin_amdgcn_exp2f(fmaf(acc[1], L2E, mnl))) +
                                                                       (__builtin_amdgcn_exp2f(fmaf(acc[2], L2E, mnl)) + __builtin_amdgcn_exp2f(fmaf(acc[3], L2E, mnl))));
                mx = mn;
            }
            const bool vin = t < tfull || (t == tfull && qin);
            if (!(JS_ABLATE & 1) || acc[0] == 1234.5f) js_store16(vin ? row + 16 * t : dumpq, acc);   // S: the tile, un-normalised in the log-softmax form
            if (LSM && has_prev) {
                // F: fix-up of the previous row tile's tile t (its load is JS_LA stages old: 4 (LA - 1) + 2 younger operations,
                // 3 (LA - 1) + 1 for a wave without DMA; fewer in the first stages of a row tile), L: the load for tile t + JS_LA
                if (t >= JS_LA) { if (hasD) JS_WAIT(10); else JS_WAIT(7); }
                else { if (hasD) JS_WAIT(4); else JS_WAIT(3); }
                f32x4_ v = *reinterpret_cast<const f32x4_*>(js_smem + fix0 + 1024 * fx + rd);
                v[0] -= prev_lse; v[1] -= prev_lse; v[2] -= prev_lse; v[3] -= prev_lse;
                js_store16(vin ? prev_row + 16 * t : dumpq, v);
                const int tl = t + JS_LA < ntv ? t + JS_LA : ntv - 1;          // (past the end: a harmless re-load keeps the count)
                js_dma_lane(prev_row + 16 * tl, lds0 + fix0 + 1024 * fx);
                fx = fx + 1 == JS_LA ? 0 : fx + 1;
            }
            // bottom: this wave's piece of the next stage has landed (JS_PF * ops-per-stage - 1 younger operations);
            // everyone is done reading this stage's slot
            if (hasD) { if (LSM && has_prev && t >= JS_PF) js_vmcnt15(); else js_vmcnt7(); }
            if (t == 4 && more) {                                  // the pop of stage 0 is >= 10 operations old: complete after the wait above
                if (!hasD) js_vmcnt7();
                asm volatile("" : "+v"(popped));
                if (wave == 0 && lane == 0) nxt_lds[qi ^ 1] = (int)gridDim.x + popped;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(JS_ABLATE & 8)) __builtin_amdgcn_s_barrier();
            c_slot = c_slot + 1 == JS_NSLOT ? 0 : c_slot + 1;
        };
#pragma unroll 1
        for (int t = 0; t < ntv; ++t) stage(t);
        if constexpr (LSM) {                                       // the row's log-sum-exp from its four quarter-rows (lanes i, i + 16, i + 32, i + 48)
            const float L2E = 1.4426950408889634f;
            float ma = fmaxf(mx, __shfl_xor(mx, 16, 64));
            ma = fmaxf(ma, __shfl_xor(ma, 32, 64));
            float sa = sm * __builtin_amdgcn_exp2f((mx - ma) * L2E);
            sa += __shfl_xor(sa, 16, 64);
            sa += __shfl_xor(sa, 32, 64);
            prev_lse = ma + __logf(sa);
            prev_row = row;
            has_prev = true;
        }
        if (!more) break;
        form_a(next);                                              // (its loads are hipcc's: it drains the queue before their first use)
        tile = next;
        qi ^= 1;
    }
    jr_vmcnt0();                                                   // every DMA of the run-ahead has landed: the LDS may be re-used by another workgroup
    if constexpr (LSM) {                                           // fix-up of the last row tile
        const int tfu = tfull, nq = ntv;
        for (int t = 0; t < nq; ++t) {
            if (!(t < tfu || (t == tfu && qin))) continue;
            f32x4_ v = *reinterpret_cast<const f32x4_*>(prev_row + 16 * t);
            v[0] -= prev_lse; v[1] -= prev_lse; v[2] -= prev_lse; v[3] -= prev_lse;
            js_store16(prev_row + 16 * t, v);
        }
    }
}
