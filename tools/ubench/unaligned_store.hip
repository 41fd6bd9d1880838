// Micro-test: 16-byte buffer stores / loads at 4-byte-aligned (not 16-byte-aligned) addresses, with the offset split
// between the VGPR offset and a SCALAR offset in all combinations of alignment.  (conv_wino4's epilogue keeps the channel
// in the scalar offset; its 16-byte stores went wrong at image widths with W % 4 != 0.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void probe(float* dst, int n, int voff_mis, int soff_mis) {
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, n * 4, 0x00020000);
    f32x4 w = {1000.f + lane, 2000.f + lane, 3000.f + lane, 4000.f + lane};
    const unsigned voff = (unsigned)lane * 16u + 4u * voff_mis;
    const int soff = 4096 + 4 * soff_mis;
    const u32x4 u = __builtin_bit_cast(u32x4, w);
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 7" ::"v"(u), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

int main() {
    const int n = 4096;
    float* d; hipMalloc(&d, n * 4);
    for (int vm = 0; vm < 4; ++vm) for (int sm = 0; sm < 4; ++sm) {
        hipMemset(d, 0, n * 4);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n, vm, sm);
        std::vector<float> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
            const int idx = 1024 + sm + lane * 4 + vm + e;
            const float want = (e + 1) * 1000.f + lane;
            if (h[idx] != want) { if (++bad <= 3) printf("  voff+%d soff+%d: lane %d elem %d at %d = %g, want %g\n", 4 * vm, 4 * sm, lane, e, idx, h[idx], want); }
        }
        printf("voffset misaligned by %2d B, soffset by %2d B (total %2d mod 16): %d wrong of 256\n", 4 * vm, 4 * sm, (4 * (vm + sm)) % 16, bad);
    }
    return 0;
}
