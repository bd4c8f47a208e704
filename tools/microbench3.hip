// One grouped-GEMM variant (ffn1 shape, 12 groups) for PMC collection, plus per-workgroup phase time stamps of
// gemm_ns_body (NS_TRACE): where does a workgroup's lifetime go?
#define NS_TRACE 1
#include "../ctc-vr_amd/csrc/rnnt_kernels.hip.h"
#include <climits>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int MT, int NT, int BK, int PD>
void run(const char* name, int G, int M, int N, int K, bool ln) {
    float *x, *y, *w, *bias;
    CK(hipMalloc(&x, (size_t)G * M * K * 4)); CK(hipMalloc(&y, (size_t)G * M * N * 4)); CK(hipMalloc(&w, (size_t)G * N * K * 4)); CK(hipMalloc(&bias, 4096 * 4));
    std::vector<float> hx((size_t)G * M * K, 0.5f), hw((size_t)G * N * K, 0.25f);
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 4096 * 4));
    std::vector<GemmP> t;
    for (int i = 0; i < G; ++i) {
        GemmP p; memset(&p, 0, sizeof(p));
        p.A = x + (size_t)i * M * K; p.W = w + (size_t)i * N * K; p.bias = bias; p.C = y + (size_t)i * M * N; p.M = M; p.N = N; p.K = K;
        p.a_n1 = INT_MAX; p.a_n2 = INT_MAX; p.a_s2 = K; p.a_seg = INT_MAX; p.ldw = K; p.c_n = INT_MAX; p.c_mod = INT_MAX; p.c_s1 = N;
        p.epi = EPI_SILU; p.alpha = 1.f; p.x_n = 1; p.a_plain = 1; p.c_plain = 1;
        if (ln) { p.ln_g = bias; p.ln_b = bias; }
        t.push_back(p);
    }
    GemmP* tab; CK(hipMalloc(&tab, G * sizeof(GemmP)));
    CK(hipMemcpy(tab, t.data(), G * sizeof(GemmP), hipMemcpyHostToDevice));
    const int ntn = N / (32 * NT), ntm = M / (32 * MT);
    const int nwg = (G * ntn + 7) / 8 * 8 * ntm;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((gemm_ns_tab<MT, NT, BK, PD>), dim3(nwg), dim3(256), 0, 0, tab, G, ntn, ntm, 8);
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL((gemm_ns_tab<MT, NT, BK, PD>), dim3(nwg), dim3(256), 0, 0, tab, G, ntn, ntm, 8);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> tr((size_t)nwg * 8);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(ns_trace), tr.size() * 8));
    long long t0 = LLONG_MAX, t1 = 0;
    double ph[4] = {0, 0, 0, 0};
    std::vector<double> starts;
    for (int g = 0; g < nwg; ++g) {
        t0 = std::min(t0, tr[g * 8]); t1 = std::max(t1, tr[g * 8 + 4]);
        for (int k = 0; k < 4; ++k) ph[k] += (tr[g * 8 + k + 1] - tr[g * 8 + k]) / 100.0;
    }
    for (int g = 0; g < nwg; ++g) starts.push_back((tr[g * 8] - t0) / 100.0);
    std::sort(starts.begin(), starts.end());
    printf("%-28s %4d WGs: launch %.2f us (events), first start -> last end %.2f us; per-WG mean us: LN stats %.2f, first tile staged %.2f, K loop %.2f, epilogue %.2f; "
           "WG start times (us after the first): median %.2f, p90 %.2f, max %.2f\n", name, nwg, ms * 1e3 / 20, (t1 - t0) / 100.0,
           ph[0] / nwg, ph[1] / nwg, ph[2] / nwg, ph[3] / nwg, starts[nwg / 2], starts[nwg * 9 / 10], starts[nwg - 1]);
    CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(w)); CK(hipFree(bias)); CK(hipFree(tab));
}
int main(int argc, char** argv) {
    // 6 pairs per launch = one layer group of the two-stream wavefront
    run<1, 2, 32, 2>("ffn1 6p <1,2,32> +LN", 6, 192, 1024, 256, true);
    run<2, 2, 32, 2>("ffn1 6p <2,2,32> +LN", 6, 192, 1024, 256, true);
    run<2, 4, 32, 2>("ffn1 6p <2,4,32> +LN", 6, 192, 1024, 256, true);
    run<1, 4, 32, 2>("ffn1 6p <1,4,32> +LN", 6, 192, 1024, 256, true);
    run<1, 1, 64, 2>("ffn2 6p <1,1,64>", 6, 192, 256, 1024, false);
    run<1, 2, 64, 2>("ffn2 6p <1,2,64>", 6, 192, 256, 1024, false);
    run<2, 2, 32, 2>("ffn2 6p <2,2,32>", 6, 192, 256, 1024, false);
    run<2, 1, 64, 2>("ffn2 6p <2,1,64>", 6, 192, 256, 1024, false);
    run<1, 1, 32, 2>("qkv 18p <1,1,32> +LN", 18, 192, 256, 256, true);
    run<1, 2, 32, 2>("qkv 18p <1,2,32> +LN", 18, 192, 256, 256, true);
    run<2, 2, 32, 2>("qkv 18p <2,2,32> +LN", 18, 192, 256, 256, true);
    run<1, 1, 32, 2>("out 6p <1,1,32>", 6, 192, 256, 256, false);
    run<1, 2, 32, 2>("out 6p <1,2,32>", 6, 192, 256, 256, false);
    run<1, 2, 32, 2>("pw1 6p <1,2,32> +LN", 6, 192, 512, 256, true);
    run<1, 1, 32, 2>("pw1 6p <1,1,32> +LN", 6, 192, 512, 256, true);
    printf("done\n");
    return 0;
}
