// one variant only, for PMC collection: ffn1 12 groups NS<2,2>
#include "../ctc-vr_amd/csrc/rnnt_kernels.hip.h"
#include <climits>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int G = 12, M = 192, N = 1024, K = 256;
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
        t.push_back(p);
    }
    GemmP* tab; CK(hipMalloc(&tab, G * sizeof(GemmP)));
    CK(hipMemcpy(tab, t.data(), G * sizeof(GemmP), hipMemcpyHostToDevice));
    const int ntn = N / 64, ntm = M / 32;
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL((gemm_ns_tab<1, 2>), dim3((G * ntn + 7) / 8 * 8 * ntm), dim3(256), 0, 0, tab, G, ntn, ntm);
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
