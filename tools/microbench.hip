// Micro-benchmarks of the path's kernels: back-to-back launches, eager on the null stream, eager on a
// non-blocking stream and as a hipGraph replay.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/microbench tools/microbench.hip
#include "../ctc-vr_amd/csrc/rnnt_kernels.hip.h"
#include <climits>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void empty_kernel() {}

static GemmP plain(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M, int N, int K, int epi) {
    GemmP p; memset(&p, 0, sizeof(p));
    p.A = A; p.W = W; p.bias = bias; p.C = C; p.M = M; p.N = N; p.K = K;
    p.a_n1 = INT_MAX; p.a_n2 = INT_MAX; p.a_s2 = lda; p.a_seg = INT_MAX; p.ldw = ldw;
    p.c_n = INT_MAX; p.c_mod = INT_MAX; p.c_s1 = ldc; p.epi = epi; p.alpha = 1.f; p.x_n = 1;
    return p;
}
template <int WK, int NT> void launch16(hipStream_t s, const GemmP& g) {
    GemmBatch gb; memset(&gb, 0, sizeof(gb)); gb.g[0] = g; gb.g[0].a_plain = 1; gb.g[0].c_plain = 1;
    dim3 grid((g.N + 16 * NT - 1) / (16 * NT), (g.M + 15) / 16, 1);
    hipLaunchKernelGGL((gemm16<WK, 1, NT>), grid, dim3(64 * WK), 0, s, gb);
}
template <int WK> void launch(hipStream_t s, const GemmP& g) {
    GemmBatch gb; memset(&gb, 0, sizeof(gb)); gb.g[0] = g;
    dim3 grid((g.N + 31) / 32, (g.M + 31) / 32, 1);
    hipLaunchKernelGGL(gemm32<WK>, grid, dim3(64 * WK), (WK * 1024 + 64) * sizeof(float), s, gb);
}

static double time_eager(hipStream_t s, int iters, const std::function<void(hipStream_t)>& f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) f(s);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) f(s);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / iters;
}
static double time_graph(hipStream_t s, int per_graph, int replays, const std::function<void(hipStream_t)>& f) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < per_graph; ++i) f(s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / (replays * per_graph);
}

int main() {
    const int M = 192;
    float *x, *w, *bias, *h, *y, *g, *bt;
    CK(hipMalloc(&x, (size_t)M * 1024 * 4)); CK(hipMalloc(&h, (size_t)M * 1024 * 4)); CK(hipMalloc(&y, (size_t)M * 1024 * 4));
    CK(hipMalloc(&w, (size_t)1024 * 1024 * 4 * 4)); CK(hipMalloc(&bias, 4096 * 4)); CK(hipMalloc(&g, 4096 * 4)); CK(hipMalloc(&bt, 4096 * 4));
    CK(hipMemset(x, 0, (size_t)M * 1024 * 4)); CK(hipMemset(w, 0, (size_t)1024 * 1024 * 16)); CK(hipMemset(bias, 0, 4096 * 4));
    CK(hipMemset(g, 0, 4096 * 4)); CK(hipMemset(bt, 0, 4096 * 4)); CK(hipMemset(h, 0, (size_t)M * 1024 * 4));
    hipStream_t nb; CK(hipStreamCreateWithFlags(&nb, hipStreamNonBlocking));
    struct Case { const char* name; std::function<void(hipStream_t)> f; };
    GemmP ffn1 = plain(x, 256, w, 256, bias, h, 1024, M, 1024, 256, EPI_SILU);
    GemmP ffn1ln = ffn1; ffn1ln.ln_g = g; ffn1ln.ln_b = bt;
    GemmP ffn2 = plain(h, 1024, w, 1024, bias, y, 256, M, 256, 1024, EPI_RESID); ffn2.R = y;
    GemmP out = plain(x, 256, w, 256, bias, y, 256, M, 256, 256, EPI_RESID); out.R = y;
    GemmP dec = plain(x, 256, w, 256, bias, y, 256, 64, 256, 256, EPI_BIAS);
    std::vector<Case> cases = {
        {"empty<<<1,64>>>", [&](hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); }},
        {"empty<<<256,256>>>", [&](hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s); }},
        {"layer_norm M=192", [&](hipStream_t s) { hipLaunchKernelGGL(layer_norm, dim3(48), dim3(256), 0, s, LnP{x, g, bt, y, M, INT_MAX, 0, 0LL, 256LL}); }},
        {"gemm32<8> ffn1 192x1024x256", [&](hipStream_t s) { launch<8>(s, ffn1); }},
        {"gemm32<8> ffn1+LN", [&](hipStream_t s) { launch<8>(s, ffn1ln); }},
        {"gemm32<16> ffn2 192x256x1024", [&](hipStream_t s) { launch<16>(s, ffn2); }},
        {"gemm32<8> ffn2 192x256x1024", [&](hipStream_t s) { launch<8>(s, ffn2); }},
        {"gemm32<8> out 192x256x256", [&](hipStream_t s) { launch<8>(s, out); }},
        {"gemm32<4> out 192x256x256", [&](hipStream_t s) { launch<4>(s, out); }},
        {"gemm16<4,2> ffn1 192x1024x256", [&](hipStream_t s) { launch16<4, 2>(s, ffn1); }},
        {"gemm16<4,2> ffn1+LN", [&](hipStream_t s) { launch16<4, 2>(s, ffn1ln); }},
        {"gemm16<4,1> ffn1+LN", [&](hipStream_t s) { launch16<4, 1>(s, ffn1ln); }},
        {"gemm16<8,1> ffn2 192x256x1024", [&](hipStream_t s) { launch16<8, 1>(s, ffn2); }},
        {"gemm16<4,1> ffn2 192x256x1024", [&](hipStream_t s) { launch16<4, 1>(s, ffn2); }},
        {"gemm16<4,1> out 192x256x256", [&](hipStream_t s) { launch16<4, 1>(s, out); }},
        {"gemm16<4,1> dec 64x256x256", [&](hipStream_t s) { launch16<4, 1>(s, dec); }},
        {"gemm32<8> dec 64x256x256", [&](hipStream_t s) { launch<8>(s, dec); }},
        {"gemm32<2> dec 64x256x256", [&](hipStream_t s) { launch<2>(s, dec); }},
    };
    printf("%-34s %10s %10s %10s\n", "kernel", "null us", "nonblk us", "graph us");
    for (auto& c : cases) {
        double a = time_eager(nullptr, 2000, c.f);
        double b = time_eager(nb, 2000, c.f);
        double d = time_graph(nb, 100, 20, c.f);
        printf("%-34s %10.2f %10.2f %10.2f\n", c.name, a, b, d);
    }
    return 0;
}
