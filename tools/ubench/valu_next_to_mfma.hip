// Micro-benchmark: how fast does a wave issue NON-MFMA instructions while the other wave of its SIMD streams fp32 MFMAs?
// 512-thread workgroups (2 waves per SIMD), one per CU: waves 0-3 run a register-only v_mfma_f32_16x16x4_f32 loop (or
// idle), waves 4-7 run a loop of one instruction kind and report cycles per instruction.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_next_to_mfma.hip -o /tmp/vm && /tmp/vm
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

// KIND 0: independent v_fma_f32   1: dependent v_fma_f32 chain   2: s_add_u32 (SALU)   3: ds_read_b32   4: v_mfma too (both stream)
template <int KIND, int MFMA_ON, int PRIO = 0>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, float a0) {
    __shared__ float sm[4096];
    const int wave = threadIdx.x >> 6;
    sm[threadIdx.x] = a0; sm[threadIdx.x + 512] = a0;
    __syncthreads();
    if (wave < 4) {
        if (!MFMA_ON) return;
        f32x4 acc[20];
#pragma unroll
        for (int i = 0; i < 20; ++i) acc[i] = f32x4{0, 0, 0, 0};
        float a = a0 + threadIdx.x, b = a0 * 0.5f + threadIdx.x;
        for (int it = 0; it < iters * 2; ++it) {        // runs longer than the measured waves
#pragma unroll
            for (int i = 0; i < 20; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 20; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = a0 + i + threadIdx.x;
        const float m = 1.0001f, c = 0.5f;
        unsigned sa = 1, sb = 3;
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        unsigned sc[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 10; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (KIND == 0) v[i] = fmaf(v[i], m, c);
                    else if (KIND == 1) v[0] = fmaf(v[0], m, c);
                    else if (KIND == 2) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sa) : "s"(sb));
                    else if (KIND == 5) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sc[i]) : "s"(sb));
                    else if (KIND == 6) asm volatile("s_nop 0");
                    else if (KIND == 3) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((threadIdx.x & 63) * 4 + i * 256)); v[i] = t; }
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[0], v[1], acc[i], 0, 0, 0);
                }
            if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_nop 0" ::: "memory");
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float s = (float)sa;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += (float)sc[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i] + acc[i][0];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave - 4] = t1 - t0;
    }
}

template <int KIND, int MFMA_ON, int PRIO = 0>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, MFMA_ON, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, 10, 1.0f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<KIND, MFMA_ON, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
    printf("%-44s partner %-12s s_memtime ticks per instruction: %.2f\n", name, MFMA_ON ? "MFMA stream" : "idle", s / 1024 / iters / 80);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0>("independent v_fma_f32"); run<0, 1>("independent v_fma_f32");
    run<1, 0>("dependent v_fma_f32 chain"); run<1, 1>("dependent v_fma_f32 chain");
    run<2, 0>("s_add_u32"); run<2, 1>("s_add_u32");
    run<3, 0>("ds_read_b32 (8 in flight)"); run<3, 1>("ds_read_b32 (8 in flight)");
    run<4, 0>("v_mfma_f32_16x16x4_f32"); run<4, 1>("v_mfma_f32_16x16x4_f32");
    run<5, 0>("s_add_u32 (8 independent chains)"); run<5, 1>("s_add_u32 (8 independent chains)");
    run<6, 0>("s_nop 0"); run<6, 1>("s_nop 0");
    printf("measured waves at s_setprio 3:\n");
    run<0, 1, 1>("independent v_fma_f32"); run<2, 1, 1>("s_add_u32"); run<3, 1, 1>("ds_read_b32 (8 in flight)"); run<6, 1, 1>("s_nop 0");
    run<4, 1, 1>("v_mfma_f32_16x16x4_f32");
    return 0;
}
