// Grouped (table-driven) GEMM variants on the wavefront shapes: 12 groups x (M=192), distinct weights per group.
#include "../ctc-vr_amd/csrc/rnnt_kernels.hip.h"
#include <climits>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static GemmP plain(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M, int N, int K, int epi) {
    GemmP p; memset(&p, 0, sizeof(p));
    p.A = A; p.W = W; p.bias = bias; p.C = C; p.M = M; p.N = N; p.K = K;
    p.a_n1 = INT_MAX; p.a_n2 = INT_MAX; p.a_s2 = lda; p.a_seg = INT_MAX; p.ldw = ldw;
    p.c_n = INT_MAX; p.c_mod = INT_MAX; p.c_s1 = ldc; p.epi = epi; p.alpha = 1.f; p.x_n = 1; p.a_plain = 1; p.c_plain = 1;
    return p;
}
template <int WK, int MT, int NT> void launch_tab(hipStream_t s, const GemmP* tab, int n, int M, int N) {
    dim3 grid((N + 16 * NT - 1) / (16 * NT), (M + 16 * MT - 1) / (16 * MT), n);
    hipLaunchKernelGGL((gemm16_tab<WK, MT, NT>), grid, dim3(64 * WK), 0, s, tab);
}
template <int MT, int NT, int BK = 32> void launch_ns(hipStream_t s, const GemmP* tab, int n, int M, int N) {
    const int ntn = (N + 32 * NT - 1) / (32 * NT), ntm = (M + 32 * MT - 1) / (32 * MT);
    hipLaunchKernelGGL((gemm_ns_tab<MT, NT, BK, 1>), dim3((n * ntn + 7) / 8 * 8 * ntm), dim3(256), 0, s, tab, n, ntn, ntm, 8);
}
static double time_eager(hipStream_t s, int iters, const std::function<void(hipStream_t)>& f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 10; ++i) f(s);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) f(s);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / iters;
}
int main() {
    const int G = 12, M = 192;
    float *x, *h, *y, *w, *bias, *g;
    CK(hipMalloc(&x, (size_t)G * M * 1024 * 4)); CK(hipMalloc(&h, (size_t)G * M * 1024 * 4)); CK(hipMalloc(&y, (size_t)G * M * 1024 * 4));
    CK(hipMalloc(&w, (size_t)G * 1024 * 1024 * 4)); CK(hipMalloc(&bias, 4096 * 4)); CK(hipMalloc(&g, 4096 * 4));
    std::vector<float> hx((size_t)G * M * 1024);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(h, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hw((size_t)G * 1024 * 1024);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) % 997) / 997.f - 0.5f;
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 4096 * 4)); CK(hipMemset(g, 0, 4096 * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    GemmP* tab; CK(hipMalloc(&tab, 64 * sizeof(GemmP)));
    int DBG = 0;
    auto fill = [&](int N, int K, bool ln, int epi, int groups) {
        std::vector<GemmP> t;
        for (int i = 0; i < groups; ++i) {
            GemmP p = plain((K == 1024 ? h : x) + (size_t)i * M * 1024, K, w + (size_t)i * 1024 * 1024, K, bias, y + (size_t)i * M * 1024, N, M, N, K, epi);
            if (ln) { p.ln_g = g; p.ln_b = bias; }
            if (epi == EPI_RESID) p.R = p.C;
            p.dbg = DBG;
            t.push_back(p);
        }
        CK(hipMemcpy(tab, t.data(), t.size() * sizeof(GemmP), hipMemcpyHostToDevice));
    };
    struct V { const char* name; std::function<void(hipStream_t)> f; };
    printf("%-44s %10s %12s\n", "variant (12 groups, M=192)", "us", "TFLOP/s");
    auto run = [&](const char* name, double flop, const std::function<void(hipStream_t)>& f) {
        double us = time_eager(s, 200, f);
        printf("%-44s %10.2f %12.1f\n", name, us, flop / us / 1e6);
    };
    const double F1 = 2.0 * G * M * 1024 * 256;
    for (int ln = 0; ln < 2; ++ln) {
        fill(1024, 256, ln, EPI_SILU, G);
        char nm[64];
        snprintf(nm, 64, "ffn1 N1024 K256 ln=%d <4,4,4>", ln); run(nm, F1, [&](hipStream_t st) { launch_tab<4, 4, 4>(st, tab, G, M, 1024); });
        snprintf(nm, 64, "ffn1 N1024 K256 ln=%d <4,2,4>", ln); run(nm, F1, [&](hipStream_t st) { launch_tab<4, 2, 4>(st, tab, G, M, 1024); });
        snprintf(nm, 64, "ffn1 N1024 K256 ln=%d <4,1,2>", ln); run(nm, F1, [&](hipStream_t st) { launch_tab<4, 1, 2>(st, tab, G, M, 1024); });
        snprintf(nm, 64, "ffn1 N1024 K256 ln=%d <4,1,1>", ln); run(nm, F1, [&](hipStream_t st) { launch_tab<4, 1, 1>(st, tab, G, M, 1024); });
    }
    for (int ln = 0; ln < 2; ++ln) {
        fill(1024, 256, ln, EPI_SILU, G);
        char nm[64];
        snprintf(nm, 64, "ffn1 ln=%d NS<2,2> (64x64)", ln); run(nm, F1, [&](hipStream_t st) { launch_ns<2, 2>(st, tab, G, M, 1024); });
        snprintf(nm, 64, "ffn1 ln=%d NS<1,2> (32x64)", ln); run(nm, F1, [&](hipStream_t st) { launch_ns<1, 2>(st, tab, G, M, 1024); });
        snprintf(nm, 64, "ffn1 ln=%d NS<2,4> (64x128)", ln); run(nm, F1, [&](hipStream_t st) { launch_ns<2, 4>(st, tab, G, M, 1024); });
        snprintf(nm, 64, "ffn1 ln=%d NS<1,2> BK64", ln); run(nm, F1, [&](hipStream_t st) { launch_ns<1, 2, 64>(st, tab, G, M, 1024); });
        snprintf(nm, 64, "ffn1 ln=%d NS<2,2> BK64", ln); run(nm, F1, [&](hipStream_t st) { launch_ns<2, 2, 64>(st, tab, G, M, 1024); });
    }
    fill(256, 1024, false, EPI_RESID, G);
    run("ffn2 N256 K1024 NS<1,2> (32x64)", F1, [&](hipStream_t st) { launch_ns<1, 2>(st, tab, G, M, 256); });
    run("ffn2 N256 K1024 NS<2,2> (64x64)", F1, [&](hipStream_t st) { launch_ns<2, 2>(st, tab, G, M, 256); });
    run("ffn2 N256 K1024 NS<1,2> BK64", F1, [&](hipStream_t st) { launch_ns<1, 2, 64>(st, tab, G, M, 256); });
    run("ffn2 N256 K1024 NS<1,1> BK64", F1, [&](hipStream_t st) { launch_ns<1, 1, 64>(st, tab, G, M, 256); });
    run("ffn2 N256 K1024 NS<1,1> (32x32)", F1, [&](hipStream_t st) { launch_ns<1, 1>(st, tab, G, M, 256); });
    fill(256, 256, false, EPI_RESID, G);
    run("out N256 K256 NS<1,2> (32x64)", 2.0 * G * M * 256 * 256, [&](hipStream_t st) { launch_ns<1, 2>(st, tab, G, M, 256); });
    run("out N256 K256 NS<1,1> BK64", 2.0 * G * M * 256 * 256, [&](hipStream_t st) { launch_ns<1, 1, 64>(st, tab, G, M, 256); });
    run("out N256 K256 NS<1,1> (32x32)", 2.0 * G * M * 256 * 256, [&](hipStream_t st) { launch_ns<1, 1>(st, tab, G, M, 256); });
    for (int dbg : {7}) {
        DBG = dbg;
        fill(1024, 256, 0, EPI_SILU, G);
        char nm[64];
        snprintf(nm, 64, "ffn1 ln=0 <4,2,4> dbg=%d (1=noload 2=nomfma 4=noepi)", dbg); run(nm, F1, [&](hipStream_t st) { launch_tab<4, 2, 4>(st, tab, G, M, 1024); });
        snprintf(nm, 64, "ffn1 ln=0 <4,1,2> dbg=%d", dbg); run(nm, F1, [&](hipStream_t st) { launch_tab<4, 1, 2>(st, tab, G, M, 1024); });
    }
    DBG = 0;
    fill(256, 1024, false, EPI_RESID, G);
    run("ffn2 N256 K1024 <8,2,4>", F1, [&](hipStream_t st) { launch_tab<8, 2, 4>(st, tab, G, M, 256); });
    run("ffn2 N256 K1024 <8,1,2>", F1, [&](hipStream_t st) { launch_tab<8, 1, 2>(st, tab, G, M, 256); });
    run("ffn2 N256 K1024 <8,1,1>", F1, [&](hipStream_t st) { launch_tab<8, 1, 1>(st, tab, G, M, 256); });
    run("ffn2 N256 K1024 <4,2,4>", F1, [&](hipStream_t st) { launch_tab<4, 2, 4>(st, tab, G, M, 256); });
    fill(256, 256, false, EPI_RESID, G);
    const double F3 = 2.0 * G * M * 256 * 256;
    run("out N256 K256 <4,2,4>", F3, [&](hipStream_t st) { launch_tab<4, 2, 4>(st, tab, G, M, 256); });
    run("out N256 K256 <4,1,1>", F3, [&](hipStream_t st) { launch_tab<4, 1, 1>(st, tab, G, M, 256); });
    fill(1024, 256, false, EPI_SILU, 1);
    run("ffn1 ONE group <4,1,2> (x1/12 flop)", F1 / 12, [&](hipStream_t st) { launch_tab<4, 1, 2>(st, tab, 1, M, 1024); });
    return 0;
}
