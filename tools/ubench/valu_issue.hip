// Issue cost of the instructions conv_wh's input transform is made of, for ONE wave alone on its SIMD (shader cycles per
// instruction from s_memtime around 32 x 64 independent copies) and for a dependent chain.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int K>
__global__ void k(unsigned long long* out, float* sink) {
    f32x2 a{1.f, 2.f}, b{1.0001f, 0.9999f}, c{0.5f, 0.25f}, d{0.f, 0.f}, e{3.f, 4.f}, f{5.f, 6.f};
    float s = threadIdx.x, t = 1.5f, u = 0.f, w = 2.f;
    unsigned h = 0, h2 = 0;
    extern __shared__ float sm[];
    unsigned lds = threadIdx.x * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 32; ++i) {
        if (K == 0) asm volatile(REP16("v_pk_fma_f32 %0, %4, %5, %6\n v_pk_fma_f32 %1, %4, %5, %6\n v_pk_fma_f32 %2, %4, %5, %6\n v_pk_fma_f32 %3, %4, %5, %6\n") : "=v"(d), "=v"(e), "=v"(f), "=v"(c) : "v"(a), "v"(b), "v"(a));
        if (K == 1) asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(d) : "v"(b), "v"(a));
        if (K == 2) asm volatile(REP16("v_fma_f32 %0, %4, %5, %6\n v_fma_f32 %1, %4, %5, %6\n v_fma_f32 %2, %4, %5, %6\n v_fma_f32 %3, %4, %5, %6\n") : "=v"(s), "=v"(t), "=v"(u), "=v"(w) : "v"(a.x), "v"(b.x), "v"(a.y));
        if (K == 3) asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(s) : "v"(b.x), "v"(a.x));
        if (K == 4) asm volatile(REP16("v_cvt_pk_f16_f32 %0, %2, %3\n v_cvt_pk_f16_f32 %1, %3, %2\n v_cvt_pk_f16_f32 %0, %2, %3\n v_cvt_pk_f16_f32 %1, %3, %2\n") : "=v"(h), "=v"(h2) : "v"(a.x), "v"(b.x));
        if (K == 5) asm volatile(REP16("v_fma_mixlo_f16 %0, %2, %3, %4\n v_fma_mixhi_f16 %0, %3, %2, %4\n v_fma_mixlo_f16 %1, %2, %3, %4\n v_fma_mixhi_f16 %1, %3, %2, %4\n") : "+v"(h), "+v"(h2) : "v"(a.x), "v"(b.x), "v"(a.y));
        if (K == 6) asm volatile(REP16("v_mov_b32_dpp %0, %4 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_shl:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %5 row_shl:2 row_mask:0xf bank_mask:0xf\n") : "+v"(s), "+v"(t), "+v"(u), "+v"(w) : "v"(a.x), "v"(b.x));
        if (K == 7) asm volatile(REP64("ds_write2st64_b64 %0, %1, %2 offset0:1 offset1:2\n") "s_waitcnt lgkmcnt(0)\n" :: "v"(lds), "v"(a), "v"(b) : "memory");
        if (K == 8) asm volatile(REP64("ds_write_b64 %0, %1 offset:512\n") "s_waitcnt lgkmcnt(0)\n" :: "v"(lds), "v"(a) : "memory");
        if (K == 9) asm volatile(REP16("v_pk_mul_f32 %0, %4, %5\n v_pk_add_f32 %1, %4, %5\n v_pk_mul_f32 %2, %4, %5\n v_pk_add_f32 %3, %4, %5\n") : "=v"(d), "=v"(e), "=v"(f), "=v"(c) : "v"(a), "v"(b));
        if (K == 10) asm volatile(REP16("v_fma_mix_f32 %0, %4, %5, %6\n v_fma_mix_f32 %1, %4, %5, %6\n v_fma_mix_f32 %2, %4, %5, %6\n v_fma_mix_f32 %3, %4, %5, %6\n") : "=v"(s), "=v"(t), "=v"(u), "=v"(w) : "v"(a.x), "v"(b.x), "v"(a.y));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[threadIdx.x] = d.x + e.x + f.x + c.x + s + t + u + w + (float)h + (float)h2 + sm[threadIdx.x];
}
int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 8 * 8); hipMalloc(&sink, 4 * 64);
    const char* names[] = {"v_pk_fma_f32 independent", "v_pk_fma_f32 dependent chain", "v_fma_f32 independent", "v_fma_f32 dependent chain",
                           "v_cvt_pk_f16_f32", "v_fma_mixlo/hi_f16 (hi after lo on one register)", "v_mov_b32_dpp row_shr/shl:2", "ds_write2st64_b64",
                           "ds_write_b64", "v_pk_mul/add_f32 independent", "v_fma_mix_f32"};
#define RUN(K) { hipLaunchKernelGGL(k<K>, dim3(1), dim3(64), 8192, 0, out, sink); hipLaunchKernelGGL(k<K>, dim3(1), dim3(64), 8192, 0, out, sink); hipDeviceSynchronize(); unsigned long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); printf("%-50s %6.2f cycles per instruction (one wave)\n", names[K], (double)h / (32.0 * 64.0)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10)
    // s_memtime: is it the shader clock?  time 2^20 dependent v_fma against the wall clock
    return 0;
}
