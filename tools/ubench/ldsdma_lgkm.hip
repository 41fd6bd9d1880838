// Micro-benchmark: does an in-flight LDS-DMA (buffer_load ... lds) hold up `s_waitcnt lgkmcnt(0)` / a ds_read of OTHER
// LDS data issued by the same wave?  One wave per workgroup; per iteration: [optional DMA of 1 KiB from a cold
// (HBM) address] ; ds_read + s_waitcnt lgkmcnt(0) ; s_memtime.  Prints cycles of the ds_read round trip with and without
// the DMA in flight, and the DMA's own landing time (s_waitcnt vmcnt(0)).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/ldsdma_lgkm.hip -o /tmp/ldsdma && /tmp/ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ __launch_bounds__(64) void k(const float* src, unsigned long long* out, int mode, int iters, size_t stride) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x;
    lds[lane] = (float)lane;
    lds[2048 + lane] = 1.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x3FFFFF00, 0x00020000);
    unsigned long long tsum_read = 0, tsum_land = 0;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        const unsigned off = (unsigned)(((size_t)(blockIdx.x * iters + i) * stride) & 0x1FFFFFF0) + lane * 16;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (mode == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 1024), 16, (int)off, 0, 0, 0);
        if (mode >= 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 1024 + q * 256), 16, (int)(off + q * 65536), 0, 0, 0);
            // probe the LDS read latency every ~130 cycles while the four fills arrive
            unsigned long long worst = 0, tq0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int r = 0; r < 24; ++r) {
                unsigned long long ta = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float vv;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(vv) : "v"(lane * 4) : "memory");
                unsigned long long tb = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                acc += vv;
                if (tb - ta > worst) worst = tb - ta;
                if (mode == 3) __builtin_amdgcn_s_sleep(1);
            }
            (void)tq0;
            tsum_read += worst;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned long long t2b = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tsum_land += t2b - t0;
            continue;
        }
        float v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lane * 4) : "memory");
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += v;
        tsum_read += t1 - t0;
        tsum_land += t2 - t0;
    }
    if (lane == 0) { out[blockIdx.x * 2] = tsum_read; out[blockIdx.x * 2 + 1] = tsum_land; }
    if (acc == 12345.f) out[0] = 0;
}

int main() {
    const size_t bytes = 1ull << 30;
    float* src; hipMalloc(&src, bytes); hipMemset(src, 0, bytes);
    unsigned long long* out; hipMalloc(&out, 256 * 2 * sizeof(unsigned long long));
    const int iters = 200;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, src, out, mode, iters, (size_t)4096 * 17);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(512);
        hipMemcpy(h.data(), out, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double r = 0, l = 0;
        for (int b = 0; b < 256; ++b) { r += h[2 * b]; l += h[2 * b + 1]; }
        printf("mode %d (%s): ds_read round trip %.0f cycles, vmcnt(0) reached after %.0f cycles\n", mode,
               mode == 0 ? "no DMA" : mode == 1 ? "1 KiB LDS-DMA from a cold address in flight" : "4 x 1 KiB LDS-DMA, WORST of 24 probes", r / 256 / iters, l / 256 / iters);
    }
    return 0;
}
