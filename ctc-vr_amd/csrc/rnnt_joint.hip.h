// joint_lattice_rows<NSPLIT,F16,LSM>: TransducerJoint.forward in lattice form (model/component/joint.py:48-69; training lattice
// online_rnnt_model.py:243, beam log_softmax :446-447) as ONE kernel, round-3 form:
//     out[b,t,u,:] = (log_softmax?)( tanh(e[b,t,:] + p[b,u,:]) * W_out^T + b_out )
// Part of rnnt_kernels.hip.h (include that umbrella, not this file).
//
// The lattice is output-bound (1648 bytes per cell against 211 kFLOP): 735 MB of f32 leave the chip, so the kernel
// is built around the store stream.
//   * Row-owner waves.  The MFMA runs TRANSPOSED: its M side is the vocabulary (A operand = a W_out fragment), its N side is 16
//     lattice rows (B operand = the tanh'd activation fragment).  In the 16x16 C layout a lane then owns ONE lattice row
//     (lane & 15) and, per vocabulary tile, four CONSECUTIVE vocabulary entries (4 * (lane >> 4) + r): the result leaves the
//     accumulators as 16-byte stores with no LDS round trip, and a wave owns its 16 rows x the WHOLE vocabulary (26 tiles,
//     104 accumulator registers), so bias, row maximum and log-sum-exp are lane-local plus two shuffles -- no cross-wave exchange.
//   * The activation operand never touches LDS: lane (i = lane & 15, q = lane >> 4) needs tanh(e + p)[row i][32 s + 8 q + 0..7]
//     for the 8 k-steps, loads those 8 floats of e and p itself, and splits them into hi / lo planes in registers
//     (64 registers for K = 256).
//   * W_out streams from L2 by LDS-DMA (global_load_lds_dwordx4, no VGPRs, no ds_write) through a 3-slot ring of 26 KiB stages in
//     fragment order (pack_joint_w): one 1-KiB piece = one MFMA operand, read back lane-linear by ds_read_b128 (conflict-free).
//     Split modes: stage = (k-step, vocabulary half) = 13 tiles x (hi, lo); bf16: stage = k-step = 26 tiles x hi.
//     One barrier per stage; the stage two ahead is in flight across it (counted vmcnt).
//   * Workgroup = 4 waves = 64 lattice rows, persistent over row tiles, TWO workgroups per CU (2 x 80 KB of LDS, 2 waves per
//     SIMD): while one workgroup's waves sit in their store phase (26 x 16-byte-per-lane stores each, paced by HBM) the other
//     one's own the MFMA pipe.  That overlap is what the round-2 kernel (one 128-row workgroup per CU, W_out restaged through
//     VGPRs + ds_write by every workgroup, two barriers + an LDS round trip per output half) did not have.
#pragma once

#define JR_NT 26                 // vocabulary tiles of 16 per wave: V <= 416
#define JR_ROWS 64               // lattice rows per workgroup tile (4 waves x 16)
// ring stage = 26 1-KiB pieces: split modes (k-step, vocabulary half) = 13 tiles x (hi, lo); plain bf16 k-step = 26 tiles x hi
#define JR_PIECES 26
#define JR_SLOT (JR_PIECES * 1024)
#define JR_LDS_BYTES (3 * JR_SLOT + JR_NT * 16 * 4)          // ring + bias table; + 16 bytes of queue words behind it
#define JR_LDS_ALLOC (JR_LDS_BYTES + 16)
#define JR_STAGGER 20                                        // start-delay step of the workgroups (x hash 0..15 x 64 clocks): about one tile period in all
#define JR_WGS_PER_CU 2                                      // by LDS (2 x 80 KB); a third workgroup (half-size stages, <= 168 VGPRs) measured slower
// e and p arrive multiplied by 2 log2(e) (the epilogue scale of the two small GEMMs), so that
//     tanh(x) = 1 - 2 / (exp(2 x) + 1) = 1 - 2 / (exp2(e' + p') + 1)
// costs add, v_exp, add, v_rcp, fma (no clamps: exp2 overflow -> inf -> rcp 0 -> 1; underflow -> 0 -> 1 - 2 = -1).
#define JR_PRESCALE 2.8853900817779268f
__device__ __forceinline__ float jr_tanh_pre(float t) { return fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(t) + 1.0f), 1.0f); }   // v_rcp_f32: 1 ulp (__frcp_rn is a 10-instruction IEEE division)
#ifndef JR_DMA_SPREAD
#define JR_DMA_SPREAD 1          // 1: a stage's DMA issue in two groups between its MFMAs; 0: all at the top of the stage
#endif
#ifndef JR_ABLATE
#define JR_ABLATE 0              // timing experiments only (wrong results): 1 no output stores, 2 no W DMA, 4 no e / p loads, 8 contiguous dummy stores
#endif
#ifndef JR_TRACE
#define JR_TRACE 0               // diagnostic build: per-wave phase cycle sums (s_memtime) into jr_trace (never in the product build)
#endif
#if JR_TRACE
__device__ long long jr_trace[2048 * 8];
#define JR_STAMP(k) do { const long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); tr[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define JR_STAMP(k) do { } while (0)
#endif
#ifndef JR_STORE
#define JR_STORE 1               // output stores: 1 plain (fastest here: the 64-byte pieces of a line merge in L2), 0 non-temporal, 2 sc1
#endif
#ifndef JR_STAGE_OUT
#define JR_STAGE_OUT 0           // 0: 16-byte stores straight from the accumulators; 1: rows leave through the free ring slot as whole 1-KiB runs
                                 //    (measured: the LDS round trip costs more than the better store shape gains, 12.5k vs 6.8k clocks per tile)
#endif

struct JointRP {
    const float* e;              // [B*T][256]  JR_PRESCALE * joint.enc_ffn(enc)
    const float* p;              // [B*U][256]  JR_PRESCALE * joint.pred_ffn(pred)
    const unsigned char* wfrag;  // pack_joint_w stream: [16 or 8 stages][26 pieces][64 lanes][8 x 16 bit]
    const float* bias;           // [V]
    float* out;                  // [M][V]
    long long M;                 // B*T*U
    int T, U, V;
    int ntiles;                  // ceil(M / 64)
    int* counter;                // dynamic row-tile queue: zeroed by the host before every launch (tiles >= gridDim.x are drawn from it)
    int stagger;                 // start delay step (s_sleep units of 64 clocks) x a per-workgroup hash in 0..15; 0 = none
};

// W_out [V][256] f32 -> the ring's stage stream.  Piece (stage g, slot j) holds for lane l the 8 k-consecutive 16-bit values
// W[16 t + (l & 15)][32 s + 8 (l >> 4) + 0..7] of plane pl;  split modes: s = g >> 1, t = 13 (g & 1) + (j >> 1), pl = j & 1;
// bf16: s = g, t = j, pl = 0.  Rows >= V are zero.  hi = round16(x), lo = round16(x - hi): the planes of split_planes.
template <bool F16, bool LO>
__global__ void pack_joint_w(const float* __restrict__ w, int V, unsigned char* __restrict__ dst) {
    constexpr int NSTG = LO ? 16 : 8;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // (g, j, lane)
    if (idx >= NSTG * JR_PIECES * 64) return;
    const int lane = idx & 63, pj = idx >> 6, j = pj % JR_PIECES, g = pj / JR_PIECES;
    const int s = LO ? (g >> 1) : g, t = LO ? 13 * (g & 1) + (j >> 1) : j, pl = LO ? (j & 1) : 0;
    const int row = 16 * t + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (row < V) { a = ldg4(w + (long long)row * RNNT_D + k0); b = ldg4(w + (long long)row * RNNT_D + k0 + 4); }
    uint4 h, l;
    split8_16<F16, true>(a, b, h, l);
    *reinterpret_cast<uint4*>(dst + (long long)idx * 16) = pl ? l : h;
}

// LDS-DMA pieces: 64 lanes x 16 bytes each, global (UNIFORM 64-bit base in SGPRs + one 32-bit lane offset: the saddr form, so
// there are no per-lane 64-bit address registers for the compiler to hoist and spill) -> LDS (M0 = wave-uniform byte address,
// + 16 * lane); the instruction's immediate offset applies to BOTH addresses.  Inline asm: hipcc must not count these on
// vmcnt (it would drain them before every LDS read); the waits below are counted by hand, and the kernel must not spill
// (a compiler-inserted scratch access between a DMA and its counted wait would be miscounted: build.py checks scratch == 0).
// M0 is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void jr_dma4(const unsigned char* gbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void jr_dma3(const unsigned char* gbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void jr_dma2(const unsigned char* gbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void jr_dma1(const unsigned char* gbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}
#if JR_ABLATE & 4
#define JR_LD(p) make_float4(0.25f, -0.5f, 0.125f, 1.0f)
#else
#define JR_LD(p) ldg4(p)
#endif
__device__ __forceinline__ void jr_store(float* p, float4 v) {
#if JR_STORE == 0
    stg4_nt(p, v);
#elif JR_STORE == 1
    stg4(p, v);
#else
    const f32x4_ vv = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(vv) : "memory");
#endif
}
// returning atomic add on the tile queue, hidden from hipcc's vmcnt bookkeeping like the DMAs (the caller's counted wait covers it)
__device__ __forceinline__ void jr_queue_pop(int* ctr, int& old) {
    asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(old) : "v"(ctr), "v"(1) : "memory");
}
__device__ __forceinline__ void jr_vmcnt7() { asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }
__device__ __forceinline__ void jr_vmcnt5() { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
__device__ __forceinline__ void jr_vmcnt4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
__device__ __forceinline__ void jr_vmcnt3() { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
__device__ __forceinline__ void jr_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int NSPLIT, bool F16, bool LSM>
__global__ __launch_bounds__(256, JR_WGS_PER_CU) void joint_lattice_rows(JointRP P) {
    constexpr bool LO = NSPLIT == 2;
    constexpr int NSTG = LO ? 16 : 8;                          // ring stages per row tile
    constexpr int TPS = LO ? 13 : 26;                          // vocabulary tiles per stage
    constexpr int SLOT = JR_SLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned char jr_smem[];
    float* biasl = reinterpret_cast<float*>(jr_smem + 3 * SLOT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    // this wave's DMA share of a stage: pieces 7 w .. 7 w + 6 (waves 0-2), 21 .. 25 (wave 3), issued as 4 + 3 (4 + 1); everything
    // but voff is scalar
    const unsigned voff = lane * 16;
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lds0 = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)jr_smem);
#else
    const unsigned lds0 = 0;
#endif
    const unsigned char* const wsrc = P.wfrag + 7 * 1024 * wave;
    int woff = 0;                                              // byte offset of the next stage to fetch (running, wraps per row tile)
    auto dma_first = [&](int slot_off) { if (JR_ABLATE & 2) return; jr_dma4(wsrc + woff, voff, lds0 + slot_off + 7 * 1024 * wave); };
    auto dma_second = [&](int slot_off) {
        const unsigned char* src = wsrc + woff + 4096;
        const unsigned dst = lds0 + slot_off + 7 * 1024 * wave + 4096;
        woff += SLOT;
        if (woff == NSTG * SLOT) woff = 0;
        if (JR_ABLATE & 2) return;
        if (wave < 3) jr_dma3(src, voff, dst); else jr_dma1(src, voff, dst);
    };
    auto dma_stage = [&](int slot_off) { dma_first(slot_off); dma_second(slot_off); };
    auto wait_prev_stage = [&]() {                             // all of this wave's DMAs except the youngest stage's have landed
        if (wave < 3) jr_vmcnt7(); else jr_vmcnt5();
    };
    int tile = blockIdx.x;
    if (tile >= P.ntiles) return;                              // uniform
    // Every workgroup of the grid starts at the same time and every row tile costs the same, so without this the whole chip
    // alternates between a compute phase and a store phase (measured: ~470 clocks per store instruction, HBM idle in between).
    // A one-off start delay spreads the workgroups' phases over a tile period; the dynamic tile queue keeps the tail balanced.
    if (P.stagger > 0) {
        const int steps = (int)(((unsigned)blockIdx.x * 0x9E3779B1u) >> 28);
        for (int k = 0; k < steps * P.stagger; ++k) __builtin_amdgcn_s_sleep(1);
    }
    int* const nxt_lds = reinterpret_cast<int*>(jr_smem + JR_LDS_BYTES);   // [2]: next tile of this workgroup (ping-pong)
    dma_stage(0);
    dma_stage(SLOT);
    for (int v = tid; v < JR_NT * 16; v += 256) biasl[v] = v < P.V ? ldg1(P.bias + v) : -INFINITY;
    uint4 ah[8], al[LO ? 8 : 1];
    // Operand formation: every load of the tile in flight at once (128 registers: the accumulators are dead here), so the L2 round
    // trip -- and the wait behind the previous tile's stores -- is paid once per tile, not once per k-step.  (Issuing part of the
    // loads before the softmax arithmetic / the stores was tried: hipcc rotates the loop and keeps prologue loads alive across
    // the k-loop, spilling 30+ registers.  Interleaving the store burst with the next tile's loads and formation in four rounds --
    // seven stores, two k-steps -- compiles clean at 256 registers and is slower: 418 vs 408 us split, 346 vs 281 us plain bf16.)
    auto form_a = [&](int tl) {
        const int am = min(tl * JR_ROWS + wave * 16 + i, (int)P.M - 1);   // M < 2^31 (host check)
        const int bt = am / P.U;                               // (b, t) row of e
        const int u = am - bt * P.U;
        const int bb = bt / P.T;
        const float* eg = P.e + (long long)bt * RNNT_D + 8 * kq;
        const float* pg = P.p + (long long)(bb * P.U + u) * RNNT_D + 8 * kq;
        float4 ld[8][4];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            ld[s][0] = JR_LD(eg + 32 * s); ld[s][1] = JR_LD(eg + 32 * s + 4); ld[s][2] = JR_LD(pg + 32 * s); ld[s][3] = JR_LD(pg + 32 * s + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float4 e0 = ld[s][0], e1 = ld[s][1], p0 = ld[s][2], p1 = ld[s][3];
            float4 v0, v1;
            v0.x = jr_tanh_pre(e0.x + p0.x); v0.y = jr_tanh_pre(e0.y + p0.y); v0.z = jr_tanh_pre(e0.z + p0.z); v0.w = jr_tanh_pre(e0.w + p0.w);
            v1.x = jr_tanh_pre(e1.x + p1.x); v1.y = jr_tanh_pre(e1.y + p1.y); v1.z = jr_tanh_pre(e1.z + p1.z); v1.w = jr_tanh_pre(e1.w + p1.w);
            uint4 h, l;
            split8_16<F16, LO>(v0, v1, h, l);
            ah[s] = h;
            if constexpr (LO) al[s] = l;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int popped = 0;
    if (wave == 0 && lane == 0) jr_queue_pop(P.counter, popped);
    form_a(tile);
    jr_vmcnt0();
    asm volatile("" : "+v"(popped));
    if (wave == 0 && lane == 0) nxt_lds[0] = (int)gridDim.x + popped;
    __syncthreads();                                           // stages 0 and 1, the bias table and the second tile's index are in LDS
    int qi = 0;
#if JR_TRACE
    long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_amdgcn_s_memtime();
    const long long tbegin = tlast;
#endif
    int s0 = 0, s1 = SLOT, s2 = 2 * SLOT;                // byte offsets of the slots holding stage G, G+1 and (free) G+2
    const int rd = lane * 16;
    while (true) {
        const int next = __builtin_amdgcn_readfirstlane(nxt_lds[qi]);
        const bool more = next < P.ntiles;                     // uniform
        f32x4_ acc[JR_NT];                                     // starts as the bias (-inf beyond the vocabulary: those columns stay -inf)
#pragma unroll
        for (int t = 0; t < JR_NT; ++t) acc[t] = *reinterpret_cast<const f32x4_*>(&biasl[16 * t + 4 * kq]);
#pragma unroll
        for (int g = 0; g < NSTG; ++g) {
            // stage G+2 goes into the slot stage G-1 was read from (every wave passed the barrier behind those reads)
            const bool pf = g + 2 < NSTG || more;
            if (g == 0 && more && wave == 0 && lane == 0) jr_queue_pop(P.counter, popped);   // the tile after next
            if (pf && !JR_DMA_SPREAD) dma_stage(s2);
            const unsigned char* sl = jr_smem + s0 + rd;
            const int s = LO ? (g >> 1) : g;
            // fragments two tiles ahead of the MFMAs that use them (3 register sets), fenced so that the scheduler keeps it so
            constexpr int PL = LO ? 2 : 1;                     // pieces per vocabulary tile
            uint4 wf[3][PL];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int c = 0; c < PL; ++c) wf[q][c] = *reinterpret_cast<const uint4*>(sl + (PL * q + c) * 1024);
#pragma unroll
            for (int jt = 0; jt < TPS; ++jt) {
                const int t = LO ? 13 * (g & 1) + jt : jt;
                if (jt + 2 < TPS)
#pragma unroll
                    for (int c = 0; c < PL; ++c) wf[(jt + 2) % 3][c] = *reinterpret_cast<const uint4*>(sl + (PL * (jt + 2) + c) * 1024);
                if constexpr (LO) {
                    acc[t] = mfma16_<F16>(wf[jt % 3][0], al[s], acc[t]);
                    acc[t] = mfma16_<F16>(wf[jt % 3][1], ah[s], acc[t]);
                    acc[t] = mfma16_<F16>(wf[jt % 3][0], ah[s], acc[t]);
                } else {
                    acc[t] = mfma16_<F16>(wf[jt % 3][0], ah[s], acc[t]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (JR_DMA_SPREAD && pf && jt == 1) dma_first(s2);
                if (JR_DMA_SPREAD && pf && jt == TPS / 2 + 1) dma_second(s2);
            }
            // stage G+1 complete in LDS for everyone, and everyone is done reading stage G's slot
            JR_STAMP(0);                                       // MFMA + LDS reads + DMA issue
            if (pf) wait_prev_stage(); else jr_vmcnt0();
            if (g == 0 && more) {
                asm volatile("" : "+v"(popped));               // the pop is older than this stage's DMAs: the wait above covered it
                if (wave == 0 && lane == 0) nxt_lds[qi ^ 1] = (int)gridDim.x + popped;
            }
            JR_STAMP(1);                                       // wait for the DMA
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            JR_STAMP(2);                                       // barrier
            const int tmp = s0; s0 = s1; s1 = s2; s2 = tmp;
        }
        const long long m = (long long)tile * JR_ROWS + wave * 16 + i;
        // ---- epilogue on the accumulators: bias, (log-softmax over the lane's row), 16-byte stores ---------------------------
        if constexpr (LSM) {
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < JR_NT; ++t) mx = fmaxf(fmaxf(mx, fmaxf(acc[t][0], acc[t][1])), fmaxf(acc[t][2], acc[t][3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float L2E = 1.4426950408889634f, mxl = -mx * L2E;
            float sm = 0.f;
#pragma unroll
            for (int t = 0; t < JR_NT; ++t)
                sm += (__builtin_amdgcn_exp2f(fmaf(acc[t][0], L2E, mxl)) + __builtin_amdgcn_exp2f(fmaf(acc[t][1], L2E, mxl))) +
                      (__builtin_amdgcn_exp2f(fmaf(acc[t][2], L2E, mxl)) + __builtin_amdgcn_exp2f(fmaf(acc[t][3], L2E, mxl)));
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            const float lse = mx + __logf(sm);                 // log_softmax = x - logsumexp(x)
#pragma unroll
            for (int t = 0; t < JR_NT; ++t) { acc[t][0] -= lse; acc[t][1] -= lse; acc[t][2] -= lse; acc[t][3] -= lse; }
        }
        JR_STAMP(3);                                           // epilogue arithmetic
        if (JR_ABLATE & 1) {
#pragma unroll
            for (int t = 0; t < JR_NT; ++t) asm volatile("" :: "v"(acc[t]));
        } else if (JR_ABLATE & 8) {                           // timing experiment: the same bytes as 1-KiB-contiguous store instructions
            float* ob = P.out + ((long long)tile * JR_ROWS + wave * 16) * P.V + lane * 4;
#pragma unroll
            for (int t = 0; t < JR_NT; ++t) jr_store(ob + t * 256, make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]));
        } else if (JR_STAGE_OUT) {
            // Straight from the accumulators a store instruction writes 16 rows x 64 bytes, and because a row is 4 V bytes (1648:
            // 16-byte aligned only) every 64-byte piece straddles two sectors: 32 partial-sector writes per instruction (measured:
            // ~400 clocks of issue per store).  Instead the wave's 16 rows -- ONE contiguous, 128-byte aligned run of 64 V bytes in
            // the output -- leave in four rounds of 4 rows through the ring slot that is free between two row tiles (s2: its next
            // DMA is issued in the next tile's first stage, behind the barrier below): the 16 lanes owning rows 4r..4r+3 write
            // their quads into a flat [4][V] image (this wave's quarter of the slot), the whole wave reads it back lane-linear and
            // stores 1-KiB runs.  LDS operations of one wave execute in order: no wait between the rounds' writes and reads.
            unsigned char* stg = jr_smem + s2 + wave * (JR_SLOT / 4);
            const long long rows_left = P.M - ((long long)tile * JR_ROWS + wave * 16);     // valid rows of this wave (may be <= 0)
            float* obase = P.out + ((long long)tile * JR_ROWS + wave * 16) * P.V;
            const int tfull = P.V >> 4;
            const bool qin = kq < ((P.V & 15) >> 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if ((i >> 2) == r) {
                    float* srow = reinterpret_cast<float*>(stg) + (i & 3) * P.V + 4 * kq;
#pragma unroll
                    for (int t = 0; t < JR_NT; ++t)
                        if (t < tfull || (t == tfull && qin)) *reinterpret_cast<f32x4_*>(srow + 16 * t) = acc[t];
                }
                const int nunits = (int)min((long long)P.V, max(0LL, rows_left - 4 * r) * (P.V >> 2));   // 16-byte units of this round
#pragma unroll
                for (int c = 0; c < (JR_NT * 16 + 63) / 64; ++c) {
                    const int un = c * 64 + lane;
                    if (un < nunits) {
                        const float4 v = *reinterpret_cast<const float4*>(stg + un * 16);
                        jr_store(obase + (long long)r * 4 * P.V + un * 4, v);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // every wave is done with the slot before the next stage's DMA lands in it
        } else if (m < P.M) {
            float* orow = P.out + m * P.V + 4 * kq;
            const int tfull = P.V >> 4;                        // tiles below are whole, tile tfull holds (V & 15) / 4 quads (V % 4 == 0)
            const bool qin = kq < ((P.V & 15) >> 2);
#pragma unroll
            for (int t = 0; t < JR_NT; ++t)
                if (t < tfull || (t == tfull && qin)) jr_store(orow + 16 * t, make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]));
        }
        JR_STAMP(4);                                           // store issue
        if (!more) break;
        __builtin_amdgcn_sched_barrier(0);
        form_a(next);
        JR_STAMP(5);                                           // next tile's tanh'd planes (its loads queue behind the stores)
        tile = next;
        qi ^= 1;
    }
#if JR_TRACE
    if (lane == 0 && blockIdx.x < 512) {
        long long* o = jr_trace + (blockIdx.x * 4 + wave) * 8;
        for (int k = 0; k < 6; ++k) o[k] = tr[k];
        o[6] = __builtin_amdgcn_s_memtime() - tbegin;
        jr_vmcnt0();
        o[7] = __builtin_amdgcn_s_memtime() - tbegin;
    }
#endif
}
