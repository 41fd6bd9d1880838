// Micro-benchmark: what does ONE extra instruction of kind K cost inside the MFMA stream of a wave that is alone on its
// SIMD?  256-thread workgroups, one per CU, __launch_bounds__(256, 1): each wave runs 60 v_mfma_f32_16x16x4_f32 per
// iteration on 60 distinct accumulator tiles (AGPRs a0..a239 by number, as conv_wino4.h does) with F fillers of kind K
// behind every MFMA (or behind every 8th for the memory kinds).  Reports cycles per MFMA (the matrix pipe needs 32).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_fillers.hip -o /tmp/mf && /tmp/mf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

template <int T>
__device__ __forceinline__ void mfma(float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(4 * T), "n"(4 * T + 3));
}

// KIND: 0 none, 1 v_fma_f32, 2 v_mov_b32, 3 s_nop 0, 4 ds_read_b32, 5 buffer_load_dwordx4 (every EVERY-th slot),
//       6 v_add_u32, 7 s_waitcnt lgkmcnt(15) (never waits), 8 ds_read_b128, 9 ds_write_b32, 10 v_fmac literal,
//       11 s_add_u32, 12 v_cndmask, 13 v_accvgpr_write of an idle AGPR (a252)
template <int KIND, int F, int EVERY>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, const float* buf, int iters, float a0) {
    __shared__ float sm[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 256) sm[i] = a0 + i;
    __syncthreads();
    asm volatile("" ::: "a0", "a255");
    sfor<240>([&](auto R) { asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"n"(decltype(R)::value)); });
    float a = a0 + tid, b = a0 * 0.5f + tid;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a0 + i + tid;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), 0, 1 << 20, 0x00020000);
    f32x4 ld[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ld[i] = f32x4{0, 0, 0, 0};
    const float* sp = sm + lane + wave * 64;
    float dsr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dsr[i] = 0;
    f32x4 dsq[4];
    int sacc = 0;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    f32x2 pk[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) pk[i] = f32x2{a0 + i, a0 - i};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        sfor<60>([&](auto S) {
            constexpr int s = decltype(S)::value;
            mfma<s>(a, b);
            if constexpr (s % EVERY == EVERY - 1) {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const int r = (s / EVERY * F + f) & 7;
                    if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(a), "v"(b));
                    if constexpr (KIND == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(v[r]) : "v"(a));
                    if constexpr (KIND == 3) asm volatile("s_nop 0");
                    if constexpr (KIND == 4) dsr[r] = sp[r * 256];
                    if constexpr (KIND == 5) ld[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + wave * 1024, (r + it * 8) * 4096 & 0xFFFFF, 0));
                    if constexpr (KIND == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[r]) : "v"(a));
                    if constexpr (KIND == 7) asm volatile("s_waitcnt lgkmcnt(15)");
                    if constexpr (KIND == 8) dsq[r & 3] = *reinterpret_cast<const f32x4*>(sm + (lane + wave * 64) * 4 + (r & 3) * 1024);
                    if constexpr (KIND == 9) sm[lane + wave * 64 + r * 256] = v[r];
                    if constexpr (KIND == 10) asm volatile("v_fmac_f32 %0, 0xc0a00000, %1" : "+v"(v[r]) : "v"(a));
                    if constexpr (KIND == 11) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                    if constexpr (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[r]) : "v"(a));
                    if constexpr (KIND == 13) asm volatile("v_accvgpr_write_b32 a252, 0");
                    if constexpr (KIND == 14) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[r & 3]) : "v"(pk[4]), "v"(pk[5]));
                    if constexpr (KIND == 15) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[r & 3]) : "v"(pk[4]));
                    if constexpr (KIND == 16) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[r & 3]) : "v"(pk[4]));
                    if constexpr (KIND == 17) asm volatile("v_accvgpr_read_b32 %0, a252" : "=v"(v[r]));
                    if constexpr (KIND == 18) asm volatile("v_exp_f32 %0, %1" : "=v"(v[r]) : "v"(a));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (KIND == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(dsr[i]));
        }
        if (KIND == 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(dsq[i]));
        }
        if (KIND == 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(ld[i]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + dsr[i] + ld[i][0] + ld[i][3];
    for (int i = 0; i < 4; ++i) s += dsq[i][1] + pk[i][0] + pk[i][1];
    float r0;
    asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a17" : "=v"(r0));
    out[blockIdx.x * 256 + tid] = s + r0 + sacc;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int KIND, int F, int EVERY>
void run(const char* name, float* out, unsigned long long* cyc, float* buf) {
    const int iters = 400;
    hipLaunchKernelGGL((k<KIND, F, EVERY>), dim3(256), dim3(256), 0, 0, out, cyc, buf, 4, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, F, EVERY>), dim3(256), dim3(256), 0, 0, out, cyc, buf, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += h[i];
    const double per = s / 1024 / iters / 60;
    const double tf = 1024.0 * iters * 60 * 2048 / (ms * 1e-3) / 1e12;
    printf("%-44s F=%d every %d: %6.1f cycles/MFMA (32 = pipe), %6.1f TF/s, clock %.2f GHz\n", name, F, EVERY, per, tf,
           per * iters * 60 / (ms * 1e-3) / 1e9);
}

#define RUNF(K, E, NAME)                      \
    run<K, 1, E>(NAME, out, cyc, buf);        \
    run<K, 2, E>(NAME, out, cyc, buf);        \
    run<K, 4, E>(NAME, out, cyc, buf);

int main() {
    float *out, *buf;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8); hipMalloc(&buf, 2 << 20);
    hipMemset(buf, 0, 2 << 20);
    run<0, 1, 1>("no fillers", out, cyc, buf);
    RUNF(1, 1, "v_fma_f32")
    run<1, 6, 1>("v_fma_f32", out, cyc, buf);
    RUNF(2, 1, "v_mov_b32")
    RUNF(6, 1, "v_add_u32")
    RUNF(10, 1, "v_fmac_f32 literal")
    RUNF(12, 1, "v_cndmask_b32")
    RUNF(13, 1, "v_accvgpr_write")
    RUNF(14, 1, "v_pk_fma_f32")
    RUNF(15, 1, "v_pk_add_f32")
    RUNF(16, 1, "v_pk_mul_f32")
    RUNF(17, 1, "v_accvgpr_read")
    RUNF(18, 1, "v_exp_f32")
    run<1, 8, 10>("v_fma_f32 bursts", out, cyc, buf);
    run<1, 12, 20>("v_fma_f32 bursts", out, cyc, buf);
    run<1, 18, 30>("v_fma_f32 bursts", out, cyc, buf);
    run<1, 36, 60>("v_fma_f32 bursts", out, cyc, buf);
    run<14, 18, 60>("v_pk_fma_f32 bursts", out, cyc, buf);
    RUNF(3, 1, "s_nop 0")
    RUNF(7, 1, "s_waitcnt lgkmcnt(15)")
    RUNF(11, 1, "s_add_u32")
    RUNF(4, 1, "ds_read_b32")
    RUNF(8, 1, "ds_read_b128")
    RUNF(9, 1, "ds_write_b32")
    run<4, 1, 2>("ds_read_b32", out, cyc, buf);
    run<5, 1, 8>("buffer_load_dwordx4 (L2)", out, cyc, buf);
    run<5, 1, 4>("buffer_load_dwordx4 (L2)", out, cyc, buf);
    run<5, 1, 2>("buffer_load_dwordx4 (L2)", out, cyc, buf);
    run<5, 1, 1>("buffer_load_dwordx4 (L2)", out, cyc, buf);
    return 0;
}
