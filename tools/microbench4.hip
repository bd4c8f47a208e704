// Do two HIP streams overlap on this system?  Stream A: 300 "big" kernels (~25 us, 1152 workgroups);
// stream B: 3000 tiny dependent kernels.  Report A alone, B alone, both together.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void big(float* p, int iters) {
    float v = p[threadIdx.x + blockIdx.x * blockDim.x];
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
    p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
__global__ void tiny(float* p) { p[threadIdx.x] += 1.f; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float *a, *b;
    CK(hipMalloc(&a, 1152 * 256 * 4)); CK(hipMalloc(&b, 4096));
    CK(hipMemset(a, 0, 1152 * 256 * 4)); CK(hipMemset(b, 0, 4096));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    auto runA = [&](hipStream_t s) { for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(big, dim3(1152), dim3(256), 0, s, a, 4000); };
    auto runB = [&](hipStream_t s, int wg) { for (int i = 0; i < 3000; ++i) hipLaunchKernelGGL(tiny, dim3(wg), dim3(64), 0, s, b); };
    for (int wg : {1, 64}) {
        runA(sa); runB(sb, wg); CK(hipDeviceSynchronize());
        double t0 = now(); runA(sa); CK(hipDeviceSynchronize()); double tA = now() - t0;
        t0 = now(); runB(sb, wg); CK(hipDeviceSynchronize()); double tB = now() - t0;
        t0 = now(); runA(sa); runB(sb, wg); CK(hipDeviceSynchronize()); double tAB = now() - t0;
        t0 = now(); runB(sb, wg); runA(sa); CK(hipDeviceSynchronize()); double tBA = now() - t0;
        // interleaved enqueue
        t0 = now();
        for (int i = 0; i < 300; ++i) { hipLaunchKernelGGL(big, dim3(1152), dim3(256), 0, sa, a, 4000); for (int j = 0; j < 10; ++j) hipLaunchKernelGGL(tiny, dim3(wg), dim3(64), 0, sb, b); }
        CK(hipDeviceSynchronize()); double tI = now() - t0;
        printf("tiny grid %d: A alone %.2f ms, B alone %.2f ms, A then B enqueued %.2f ms, B then A %.2f ms, interleaved enqueue %.2f ms\n", wg, tA, tB, tAB, tBA, tI);
    }
    return 0;
}
