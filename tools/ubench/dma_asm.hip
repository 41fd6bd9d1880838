// LDS-DMA through inline asm (s_mov_b32 m0 + buffer_load_dword ... offen lds): does it behave like the builtin?
// hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_asm.hip -o /tmp/dma_asm && /tmp/dma_asm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* in, float* out, int n, int mode) {
    extern __shared__ float smem[];
    typedef __attribute__((address_space(3))) float lds_f;
    const unsigned lds0 = (unsigned)(size_t)(lds_f*)smem;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, n * 4, 0x00020000);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned voff = (lane % 7 == 3) ? 0x40000000u : (unsigned)(lane * 4);          // some lanes out of range -> zero
    for (int i = 0; i < 4; ++i) {
        unsigned ldsaddr = lds0 + 4u * (wave * 512 + i * 64) + (mode ? 32768u : 0u);
        int soff = i * n / 16;          // (n = 4096: 256 bytes per step, kept out of the compiler's sight: a literal is no valid soffset)
        asm volatile("s_mov_b32 m0, %1\n\tbuffer_load_dword %0, %2, %3 offen lds" ::"v"(voff), "s"(ldsaddr), "s"(rs), "s"(soff) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[(wave * 4 + i) * 64 + lane] = smem[(mode ? 8192 : 0) + wave * 512 + i * 64 + lane];
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *din, *dout;
    hipMalloc(&din, n * 4); hipMalloc(&dout, 4 * 4 * 64 * 4);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 163840, 0, din, dout, n, mode);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> o(1024);
        hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < 4; ++w) for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; ++l) {
            float want = (l % 7 == 3) ? 0.f : (float)(l + i * 64);
            if (o[(w * 4 + i) * 64 + l] != want) ++bad;
        }
        printf("mode %d: %s, mismatches %d (first values %g %g %g %g)\n", mode, hipGetErrorString(e), bad, o[0], o[1], o[2], o[3]);
    }
    return 0;
}
