// Standalone check + timing of the joint lattice kernel joint_lattice_rows (rnnt_joint.hip.h) against a CPU double reference on
// (JL_STREAM=1: also the vocabulary-tile-outer experiment of tools/jl_stream_experiment.hip.h; -DJS_ABLATE=bits, -DJS_RB=4|8)
// sampled rows; B64 x T249 x U28 x V412.  Build variants: -DJR_STORE=0|1|2, -DJR_ABLATE=bits, -DJR_TRACE=1.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/jl_check tools/jl_check.hip && tools/jl_check [wgs_per_cu | -grid] [stagger] [B T U V]
#include "../ctc-vr_amd/csrc/rnnt_kernels.hip.h"
#include "jl_stream_experiment.hip.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static int g_wgs = 2, g_cus = 256, g_stagger = 0;
static int* g_counter = nullptr;

template <int NS, bool F16, bool LSM>
static float run_new(const JointRP& rp, int reps) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_lattice_rows<NS, F16, LSM>), hipFuncAttributeMaxDynamicSharedMemorySize, JR_LDS_ALLOC));
    const dim3 grid((unsigned)std::min(rp.ntiles, g_wgs < 0 ? -g_wgs : (g_wgs ? g_wgs : JR_WGS_PER_CU) * g_cus));   // negative: absolute grid size
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemsetAsync(g_counter, 0, 16, 0));
    hipLaunchKernelGGL((joint_lattice_rows<NS, F16, LSM>), grid, dim3(256), JR_LDS_ALLOC, 0, rp);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(g_counter, 0, 16, 0));
        hipLaunchKernelGGL((joint_lattice_rows<NS, F16, LSM>), grid, dim3(256), JR_LDS_ALLOC, 0, rp);
    }
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

static float* g_dump = nullptr;
template <int NS, bool F16, bool LSM>
static float run_stream(const JointRP& rp, const unsigned char* wfs, int reps) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_lattice_stream<NS, F16, LSM>), hipFuncAttributeMaxDynamicSharedMemorySize, JS_LDS_ALLOC(NS)));
    JointSP sp; sp.e = rp.e; sp.p = rp.p; sp.wfrag = wfs; sp.bias = rp.bias; sp.out = rp.out; sp.dump = g_dump; sp.M = rp.M; sp.T = rp.T; sp.U = rp.U; sp.V = rp.V;
    sp.ntiles = (int)((rp.M + JS_ROWS - 1) / JS_ROWS); sp.ntv = (rp.V + 15) / 16; sp.counter = g_counter;
    const dim3 grid((unsigned)std::min(sp.ntiles, g_cus));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemsetAsync(g_counter, 0, 16, 0));
    hipLaunchKernelGGL((joint_lattice_stream<NS, F16, LSM>), grid, dim3(1024), JS_LDS_ALLOC(NS), 0, sp);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(g_counter, 0, 16, 0));
        hipLaunchKernelGGL((joint_lattice_stream<NS, F16, LSM>), grid, dim3(1024), JS_LDS_ALLOC(NS), 0, sp);
    }
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    if (argc > 1) g_wgs = atoi(argv[1]);
    if (argc > 2) g_stagger = atoi(argv[2]);
    CK(hipMalloc(&g_counter, 16));
    int B = 64, T = 249, U = 28, V = 412;
    if (argc > 6) { B = atoi(argv[3]); T = atoi(argv[4]); U = atoi(argv[5]); V = atoi(argv[6]); }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); g_cus = prop.multiProcessorCount;
    const long long M = (long long)B * T * U;
    printf("lattice B=%d T=%d U=%d V=%d: M=%lld rows, %d CUs, %d workgroups per CU, stagger %d\n", B, T, U, V, M, g_cus, g_wgs, g_stagger);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> he((size_t)B * T * 256), hp((size_t)B * U * 256), hw((size_t)V * 256), hb(V);
    for (auto& v : he) v = nd(rng);
    for (auto& v : hp) v = nd(rng) * 0.5f;
    for (auto& v : hw) v = nd(rng) * 0.06f;
    for (auto& v : hb) v = nd(rng) * 0.1f;
    float *e, *p, *es, *ps, *w, *bias, *out1; unsigned char *wf2, *wf1;
    CK(hipMalloc(&e, he.size() * 4)); CK(hipMalloc(&p, hp.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&bias, V * 4));
    CK(hipMalloc(&out1, (size_t)M * V * 4 + 65536));
    CK(hipMalloc(&wf2, 16 * JR_SLOT)); CK(hipMalloc(&wf1, 8 * JR_SLOT));
    CK(hipMemcpy(e, he.data(), he.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    {   // the new kernel takes e and p prescaled by 2 log2(e) (in the library: the epilogue scale of the two small GEMMs)
        std::vector<float> t(he.size()); for (size_t k = 0; k < t.size(); ++k) t[k] = he[k] * JR_PRESCALE;
        CK(hipMalloc(&es, t.size() * 4)); CK(hipMemcpy(es, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        t.resize(hp.size()); for (size_t k = 0; k < t.size(); ++k) t[k] = hp[k] * JR_PRESCALE;
        CK(hipMalloc(&ps, t.size() * 4)); CK(hipMemcpy(ps, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hb.data(), V * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((pack_joint_w<false, true>), dim3((16 * JR_PIECES * 64 + 255) / 256), dim3(256), 0, 0, w, V, wf2);
    hipLaunchKernelGGL((pack_joint_w<false, false>), dim3((8 * JR_PIECES * 64 + 255) / 256), dim3(256), 0, 0, w, V, wf1);
    unsigned char *ws2, *ws1;
    const int ntv = (V + 15) / 16;
    CK(hipMalloc(&ws2, (size_t)ntv * JS_SLOT(2))); CK(hipMalloc(&ws1, (size_t)ntv * JS_SLOT(1))); CK(hipMalloc(&g_dump, (size_t)JS_ROWS * V * 4));
    hipLaunchKernelGGL((pack_joint_w_stream<false, true>), dim3((ntv * 16 * 64 + 255) / 256), dim3(256), 0, 0, w, V, ws2);
    hipLaunchKernelGGL((pack_joint_w_stream<false, false>), dim3((ntv * 8 * 64 + 255) / 256), dim3(256), 0, 0, w, V, ws1);
    CK(hipDeviceSynchronize());
    JointRP rp; rp.e = es; rp.p = ps; rp.wfrag = wf2; rp.bias = bias; rp.out = out1; rp.M = M; rp.T = T; rp.U = U; rp.V = V; rp.ntiles = (int)((M + JR_ROWS - 1) / JR_ROWS); rp.counter = g_counter; rp.stagger = g_stagger;
    const double bytes = 4.0 * ((double)B * T * 256 + (double)B * U * 256) + 4.0 * (256.0 * V + V) + 4.0 * (double)M * V;
    std::vector<float> h0((size_t)M * V), h1((size_t)M * V);
    (void)e; (void)p;
    auto compare = [&](const char* what, bool lsm, bool lo) {
        CK(hipMemcpy(h1.data(), out1, h1.size() * 4, hipMemcpyDeviceToHost));
        // CPU double reference on sampled rows (the 16-bit planes bound the error: ~1e-5 split, ~1e-2 plain bf16)
        double wref = 0;
        for (int sidx = 0; sidx < 40; ++sidx) {
            const long long m = sidx == 39 ? M - 1 : (sidx * 1000003LL) % M;
            const long long bt = m / U; const int u = (int)(m - bt * U); const long long bb = bt / T;
            std::vector<double> a(256), lg(V);
            for (int k = 0; k < 256; ++k) a[k] = tanh((double)he[bt * 256 + k] + hp[(bb * U + u) * 256 + k]);
            double mx = -1e300;
            for (int v = 0; v < V; ++v) { double s = hb[v]; for (int k = 0; k < 256; ++k) s += a[k] * hw[(size_t)v * 256 + k]; lg[v] = s; mx = std::max(mx, s); }
            double sum = 0; for (int v = 0; v < V; ++v) sum += exp(lg[v] - mx);
            const double lse = mx + log(sum);
            for (int v = 0; v < V; ++v) { const double d = fabs(h1[m * V + v] - (lsm ? lg[v] - lse : lg[v])); if (!(d <= wref)) wref = d; }
        }
        printf("  %-28s vs CPU double (40 rows): max err %.3g %s\n", what, wref, wref < (lo ? 2e-4 : 5e-2) ? "ok" : "FAIL");
    };
#define BOTH(NS, LSM, NAME)                                                                                           \
    {                                                                                                                 \
        rp.wfrag = NS == 2 ? wf2 : wf1;                                                                               \
        CK(hipMemset(out1, 0xee, (size_t)M * V * 4));                                                                 \
        const float t_new = run_new<NS, false, LSM>(rp, 5);                                                           \
        printf("%s: %.1f us (%.2f TB/s, %.3f of 8 TB/s)\n", NAME, t_new, bytes / t_new / 1e6, bytes / t_new / 8e6);    \
        compare(NAME, LSM, NS == 2);                                                                                  \
    }
    BOTH(2, false, "bf16x3 logits")
    BOTH(2, true, "bf16x3 log-softmax")
    BOTH(1, false, "bf16 logits")
    BOTH(1, true, "bf16 log-softmax")
#define STREAM(NS, LSM, NAME)                                                                                         \
    if (std::getenv("JL_STREAM")) {                                                                                    \
        rp.wfrag = NS == 2 ? wf2 : wf1;                                                                               \
        run_new<NS, false, LSM>(rp, 1);                                                                               \
        CK(hipMemcpy(h0.data(), out1, h0.size() * 4, hipMemcpyDeviceToHost));                                         \
        CK(hipMemset(out1, 0xee, (size_t)M * V * 4));                                                                 \
        const float t_new = run_stream<NS, false, LSM>(rp, NS == 2 ? ws2 : ws1, 5);                                   \
        printf("stream %s: %.1f us (%.2f TB/s, %.3f of 8 TB/s)\n", NAME, t_new, bytes / t_new / 1e6, bytes / t_new / 8e6); \
        compare("stream " NAME, LSM, NS == 2);                                                                        \
        double dm = 0; size_t nbad = 0;                                                                               \
        for (size_t k = 0; k < h0.size(); ++k) { const double d = fabs((double)h0[k] - h1[k]); if (!(d <= 1e-3)) ++nbad; if (!(d <= dm)) dm = d; } \
        printf("  stream vs rows kernel, all %zu values: max diff %.3g, %zu beyond 1e-3\n", h0.size(), dm, nbad);      \
    }
    STREAM(2, false, "bf16x3 logits")
    STREAM(2, true, "bf16x3 log-softmax")
    STREAM(1, false, "bf16 logits")
    STREAM(1, true, "bf16 log-softmax")
#if JR_TRACE
    {
        const char* nm[8] = {"mfma+lds+dma issue", "dma wait", "barrier", "epilogue math", "store issue", "operand formation", "total (last store issued)", "total (stores drained)"};
        rp.wfrag = wf2; run_new<2, false, true>(rp, 1);
        for (int mode = 0; mode < 2; ++mode) {
            if (mode) { rp.wfrag = wf1; run_new<1, false, true>(rp, 1); }
            std::vector<long long> tr(2048 * 8);
            CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(jr_trace), tr.size() * 8));
            const int nw = 4 * std::min(512, std::min(rp.ntiles, g_wgs < 0 ? -g_wgs : (g_wgs ? g_wgs : JR_WGS_PER_CU) * g_cus));
            printf("phase cycles per wave (mean over %d waves), %s log-softmax:\n", nw, mode ? "bf16" : "bf16x3");
            for (int k = 0; k < 8; ++k) { double sm = 0; for (int w = 0; w < nw; ++w) sm += tr[w * 8 + k]; printf("  %-28s %10.0f\n", nm[k], sm / nw); }
        }
    }
#endif
    // run-to-run reproducibility of the new kernel
    {
        rp.wfrag = wf2;
        run_new<2, false, true>(rp, 1);
        CK(hipMemcpy(h0.data(), out1, h0.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int r = 0; r < 3; ++r) { run_new<2, false, true>(rp, 1); CK(hipMemcpy(h1.data(), out1, h1.size() * 4, hipMemcpyDeviceToHost)); bad += memcmp(h0.data(), h1.data(), h0.size() * 4) != 0; }
        printf("reproducibility: %d of 3 reruns differ\n", bad);
    }
    return 0;
}
