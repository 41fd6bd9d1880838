// Micro-benchmark: do fp32 VALU instructions of a co-resident wave slow down fp32 MFMAs on the same SIMD?
// 512-thread workgroups (2 waves per SIMD), one per CU: waves 0-3 run a register-only v_mfma_f32_16x16x4_f32 loop (20
// accumulators), waves 4-7 run `valu_per_iter` independent v_fma_f32 per loop iteration (0 = idle partner).  Reports the
// MFMA waves' rate.  hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o /tmp/mv && /tmp/mv
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NV, int KIND>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, float a0) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        f32x4 acc[20];
#pragma unroll
        for (int i = 0; i < 20; ++i) acc[i] = f32x4{0, 0, 0, 0};
        float a = a0 + threadIdx.x, b = a0 * 0.5f + threadIdx.x;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 20; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
#pragma unroll
        for (int i = 0; i < 20; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
    } else if (NV > 0) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = a0 + i + threadIdx.x;
        const float m = 1.0001f, c = 0.5f;
        // roughly as long as the MFMA waves run: 20 MFMAs = 640 cycles per iteration
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < NV / 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (KIND == 0) v[i] = fmaf(v[i], m, c);
                    else if (KIND == 1) v[i] = __expf(v[i]) * 1e-3f;
                    else asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(v[(i + 1) & 7]));
                }
            if (NV < 80) __builtin_amdgcn_s_sleep(1);
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

template <int NV, int KIND>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(512), 0, 0, out, cyc, 10, 1.0f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
    printf("%-58s cycles per MFMA (ideal 32): %.1f\n", name, s / 1024 / iters / 20);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0>("partner wave idle");
    run<8, 0>("partner: 8 v_fma_f32 + s_sleep per 20 MFMAs");
    run<40, 0>("partner: 40 v_fma_f32 + s_sleep per 20 MFMAs");
    run<80, 0>("partner: 80 v_fma_f32 per 20 MFMAs (continuous)");
    run<160, 0>("partner: 160 v_fma_f32 per 20 MFMAs (continuous)");
    run<80, 1>("partner: 80 x (v_exp_f32 + v_mul) (continuous)");
    run<80, 2>("partner: 80 v_mov_b32 (continuous)");
    return 0;
}
