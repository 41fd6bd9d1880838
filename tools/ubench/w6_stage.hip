// conv_wino6's raw-tile staging in isolation: 28 LDS-DMA instructions per wave fill four channel planes of the 10 x 40 halo
// tile (row R at R * 41 + 2 (R >> 2), plane stride 449); compared with the tile gathered on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int RS = 41, PS = 449;
__host__ __device__ constexpr int rowoff(int R) { return R * RS + 2 * (R >> 2); }
__global__ void k(const float* in, float* out, int Cin, int H, int W, int b, int y0, int x0) {
    extern __shared__ float smem[];
    typedef __attribute__((address_space(3))) float lds_f;
    const unsigned lds0 = (unsigned)(size_t)(lds_f*)smem;
    const int lane = threadIdx.x & 63, wi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int HW = H * W;
    const unsigned HW4 = HW * 4u;
    unsigned goffd[7];
    for (int i = 0; i < 7; ++i) {
        const int P = 64 * i + lane;
        int row = 0;
        for (int r = 1; r < 10; ++r) row = P >= rowoff(r) ? r : row;
        const int col = P - rowoff(row);
        const int gy = y0 + row - 1, gx = x0 - 4 + col;
        const bool ok = col < 40 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goffd[i] = ok ? (unsigned)(gy * W + gx) * 4u : 0x40000000u;
    }
    const float* base = in + ((size_t)b * Cin + wi * 4) * HW;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 4 * (int)HW4, 0x00020000);
    for (int g = 0; g < 4; ++g)
        for (int i = 0; i < 7; ++i) {
            unsigned ldsaddr = lds0 + 4u * ((wi * 4 + g) * PS + 64 * i);
            int soff = g * (int)HW4;
            asm volatile("s_mov_b32 m0, %1\n\tbuffer_load_dword %0, %2, %3 offen lds" ::"v"(goffd[i]), "s"(ldsaddr), "s"(rs), "s"(soff) : "memory");
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = threadIdx.x; j < 16 * PS; j += 256) out[j] = smem[j];
}
int main() {
    const int B = 2, Cin = 16, H = 21, W = 72;
    std::vector<float> h((size_t)B * Cin * H * W);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 100003) * 0.5f + 1.f;
    float *din, *dout;
    (void)hipMalloc(&din, h.size() * 4); (void)hipMalloc(&dout, 16 * PS * 4);
    (void)hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int tcase = 0; tcase < 3; ++tcase) {
        const int b = tcase == 2 ? 1 : 0, y0 = tcase == 0 ? 0 : 16, x0 = tcase == 0 ? 0 : 64;
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 163840, 0, din, dout, Cin, H, W, b, y0, x0);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> o(16 * PS);
        (void)hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int c = 0; c < 16; ++c) for (int r = 0; r < 10; ++r) for (int x = 0; x < 40; ++x) {
            const int gy = y0 + r - 1, gx = x0 - 4 + x;
            const float want = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? h[(((size_t)b * Cin + c) * H + gy) * W + gx] : 0.f;
            if (o[c * PS + rowoff(r) + x] != want) ++bad;
        }
        printf("case %d: %s, mismatches %d of 6400\n", tcase, hipGetErrorString(e), bad);
    }
    return 0;
}
