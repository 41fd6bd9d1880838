// Micro-benchmark: sustained v_mfma_f32_16x16x4_f32 / 32x32x2 issue rate for the register shapes the
// conv kernel uses (NACC independent accumulators per wave, W waves per workgroup, 1-2 workgroups/CU).
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    extern __shared__ float pad[];
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + pad[0] * 0.f;
}

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    extern __shared__ float pad[];
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + pad[0] * 0.f;
}

template <typename K>
void run(const char* name, K kern, int threads, size_t lds, int blocks, int iters, double flop_per_mfma, int nacc) {
    float* out;
    hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, 10, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)blocks * (threads / 64) * iters * 4.0 * nacc;
    printf("%-44s blocks=%5d thr=%3d lds=%6zu  %8.3f ms  %7.1f TF/s\n", name, blocks, threads, lds, ms, mf * flop_per_mfma / ms / 1e9);
    hipFree(out);
}

int main() {
    const int it = 4000;
    const double f16 = 2.0 * 16 * 16 * 4, f32 = 2.0 * 32 * 32 * 2;
    // LDS request controls workgroups per CU: 70 KB -> 2/CU, 100 KB -> 1/CU, 20 KB -> up to 8
    run("16x16x4 acc=20, 4 waves, 2 WG/CU", k16<20>, 256, 70 * 1024, 512, it, f16, 20);
    run("16x16x4 acc=20, 4 waves, 1 WG/CU", k16<20>, 256, 100 * 1024, 256, it, f16, 20);
    run("16x16x4 acc=10, 4 waves, 2 WG/CU", k16<10>, 256, 70 * 1024, 512, it, f16, 10);
    run("16x16x4 acc=10, 4 waves, 3 WG/CU", k16<10>, 256, 50 * 1024, 768, it, f16, 10);
    run("16x16x4 acc=4,  4 waves, 1 WG/CU", k16<4>, 256, 100 * 1024, 256, it, f16, 4);
    run("16x16x4 acc=4,  4 waves, 2 WG/CU", k16<4>, 256, 70 * 1024, 512, it, f16, 4);
    run("16x16x4 acc=2,  4 waves, 1 WG/CU", k16<2>, 256, 100 * 1024, 256, it, f16, 2);
    run("32x32x2 acc=5,  4 waves, 2 WG/CU", k32<5>, 256, 70 * 1024, 512, it, f32, 5);
    run("32x32x2 acc=5,  4 waves, 1 WG/CU", k32<5>, 256, 100 * 1024, 256, it, f32, 5);
    run("32x32x2 acc=2,  4 waves, 1 WG/CU", k32<2>, 256, 100 * 1024, 256, it, f32, 2);
    run("32x32x2 acc=4,  4 waves, 2 WG/CU", k32<4>, 256, 70 * 1024, 512, it, f32, 4);
    return 0;
}
