// Write-bandwidth ceiling of the joint lattice's output (735 MB of f32): plain / non-temporal 16-byte stores, grid-stride and
// per-wave-contiguous (each wave writes runs of 26 KiB as the lattice kernel does).
//   hipcc -O3 --offload-arch=gfx950 -o tools/fill_check tools/fill_check.hip && tools/fill_check
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void fill_stride(f4* out, long long n) {
    const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
template <bool NT>
__global__ void fill_runs(f4* out, long long nruns, int run16) {   // run16 = 16-byte units per run; one wave per run, grid-stride over runs
    const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave; r < nruns; r += nw)
        for (int u = lane; u < run16; u += 64) { if (NT) __builtin_nontemporal_store(v, out + r * run16 + u); else out[r * run16 + u] = v; }
}
int main() {
    const long long bytes = 446208LL * 412 * 4, n = bytes / 16;
    f4* out; CK(hipMalloc(&out, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        launch();
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 5; ++r) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %7.1f us  %.2f TB/s\n", name, ms * 200.f, bytes / (ms * 200.f) / 1e6);
    };
    for (int g : {32, 64, 128, 256}) {   // few waves per CU: what one wave's store stream sustains
        char nm[96];
        snprintf(nm, sizeof nm, "26 KiB run per wave plain, %d blocks", g); time(nm, [&] { hipLaunchKernelGGL(fill_runs<false>, dim3(g), dim3(256), 0, 0, out, n / 1648, 1648); });
        snprintf(nm, sizeof nm, "26 KiB run per wave plain, %d blocks x 64 thr", g); time(nm, [&] { hipLaunchKernelGGL(fill_runs<false>, dim3(g), dim3(64), 0, 0, out, n / 1648, 1648); });
    }
    for (int g : {512, 2048, 8192}) {
        char nm[96];
        snprintf(nm, sizeof nm, "grid-stride plain, %d blocks", g); time(nm, [&] { hipLaunchKernelGGL(fill_stride<false>, dim3(g), dim3(256), 0, 0, out, n); });
        snprintf(nm, sizeof nm, "grid-stride nt,    %d blocks", g); time(nm, [&] { hipLaunchKernelGGL(fill_stride<true>, dim3(g), dim3(256), 0, 0, out, n); });
        snprintf(nm, sizeof nm, "26 KiB run per wave plain, %d blocks", g); time(nm, [&] { hipLaunchKernelGGL(fill_runs<false>, dim3(g), dim3(256), 0, 0, out, n / 1648, 1648); });
        snprintf(nm, sizeof nm, "26 KiB run per wave nt,    %d blocks", g); time(nm, [&] { hipLaunchKernelGGL(fill_runs<true>, dim3(g), dim3(256), 0, 0, out, n / 1648, 1648); });
    }
    CK(hipMemsetAsync(out, 0, bytes, 0));
    CK(hipEventRecord(e0, 0)); for (int r = 0; r < 5; ++r) CK(hipMemsetAsync(out, 0, bytes, 0)); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-44s %7.1f us  %.2f TB/s\n", "hipMemsetAsync", ms * 200.f, bytes / (ms * 200.f) / 1e6);
    return 0;
}
