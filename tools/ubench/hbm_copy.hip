// Micro-benchmark: achievable HBM bandwidth for the access shapes the elementwise / depthwise kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy4(const float4* a, float4* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void copy1(const float* a, float* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// one block per 256-float segment (no grid stride): like the per-tile kernels
__global__ void copy1_tile(const float* a, float* b, size_t n) {
    size_t i = blockIdx.x * (size_t)1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * 256 < n) b[i + k * 256] = a[i + k * 256];
}
template <typename F> void run(const char* name, F launch, double bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.3f ms  %6.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
}
int main() {
    const size_t n = (size_t)160 * 16 * 46128 + 3;   // one 160-channel activation tensor of the bench (472 MB)
    float *a, *b; hipMalloc(&a, (n + 64) * 4); hipMalloc(&b, (n + 64) * 4);
    hipMemset(a, 1, (n + 64) * 4);
    const double bytes = 2.0 * n * 4;
    run("float4 grid-stride (2048 blocks)", [&] { hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4); }, bytes);
    run("float4 grid-stride (8192 blocks)", [&] { hipLaunchKernelGGL(copy4, dim3(8192), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4); }, bytes);
    run("float  grid-stride (2048 blocks)", [&] { hipLaunchKernelGGL(copy1, dim3(2048), dim3(256), 0, 0, a, b, n); }, bytes);
    run("float  grid-stride (16384 blocks)", [&] { hipLaunchKernelGGL(copy1, dim3(16384), dim3(256), 0, 0, a, b, n); }, bytes);
    run("float  unaligned (+1) grid-stride", [&] { hipLaunchKernelGGL(copy1, dim3(4096), dim3(256), 0, 0, a + 1, b + 1, n); }, bytes);
    run("float  one 1024-float tile per block", [&] { hipLaunchKernelGGL(copy1_tile, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, a, b, n); }, bytes);
    return 0;
}
