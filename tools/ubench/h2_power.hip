// What do the gfx950 matrix pipes sustain UNDER THE SOCKET'S POWER LIMIT, for seconds, not for a burst?
// Eight waves per CU (two per SIMD), 256 workgroups, ~2 s per variant; a side thread samples the amdgpu hwmon files.
//   0: v_mfma_f32_16x16x4_f32 only          1: v_mfma_f32_32x32x16_f16 only        2: ... + ds_read_b128 at conv_h2's ratio
//   3: v_mfma_f32_16x16x32_f16 only         4: the ds_read_b128 stream alone       5: v_mfma_f32_32x32x16_bf16 only
//   6: f16 32x32x16 with ZERO operands (data-dependent power?)
// build: hipcc --offload-arch=gfx950 -O3 -o h2_power tools/ubench/h2_power.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#include <glob.h>

using h16x8 = __attribute__((ext_vector_type(8))) _Float16;
using b16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int V>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* __restrict__ dst, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = src[i];
    __syncthreads();
    if constexpr (V == 0) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        float a = src[lane], b = src[lane + 64];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        if (s == 12345.f) dst[threadIdx.x] = s;
    } else if constexpr (V == 3) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        h16x8 a = *reinterpret_cast<const h16x8*>(&lds[lane * 4]), b = *reinterpret_cast<const h16x8*>(&lds[256 + lane * 4]);
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        if (s == 12345.f) dst[threadIdx.x] = s;
    } else if constexpr (V == 4) {
        f32x4 s4{0, 0, 0, 0};
        const float* p = lds + lane * 4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 14; ++j) {
                f32x4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)(p) ), "n"(j * 1024));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                s4 += v;
            }
        }
        if (s4[0] == 12345.f) dst[threadIdx.x] = s4[1];
    } else {
        f32x16 acc[10];
        for (int i = 0; i < 10; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const float* p = lds + lane * 4;
        h16x8 a0 = *reinterpret_cast<const h16x8*>(p), a1 = *reinterpret_cast<const h16x8*>(p + 256);
        h16x8 b0 = *reinterpret_cast<const h16x8*>(p + 512), b1 = *reinterpret_cast<const h16x8*>(p + 768);
        if constexpr (V == 6) { for (int e = 0; e < 8; ++e) { a0[e] = a1[e] = b0[e] = b1[e] = (_Float16)0.f; } }
        for (int it = 0; it < iters; ++it) {
            if constexpr (V == 2) {
                // conv_h2's operand stream: 4 A fragments, then per column tile 2 B fragments, 6 MFMAs each
                h16x8 ah[2], al[2];
                ah[0] = *reinterpret_cast<const h16x8*>(p + ((it * 4 + 0) & 15) * 256);
                ah[1] = *reinterpret_cast<const h16x8*>(p + ((it * 4 + 1) & 15) * 256);
                al[0] = *reinterpret_cast<const h16x8*>(p + ((it * 4 + 2) & 15) * 256);
                al[1] = *reinterpret_cast<const h16x8*>(p + ((it * 4 + 3) & 15) * 256);
#pragma unroll
                for (int n = 0; n < 5; ++n) {
                    const h16x8 bh = *reinterpret_cast<const h16x8*>(p + (16 + ((it + 2 * n) & 15)) * 256);
                    const h16x8 bl = *reinterpret_cast<const h16x8*>(p + (32 + ((it + 2 * n + 1) & 15)) * 256);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0], bh, acc[n], 0, 0, 0);
                    acc[5 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[1], bh, acc[5 + n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bl, acc[n], 0, 0, 0);
                    acc[5 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bl, acc[5 + n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh, acc[n], 0, 0, 0);
                    acc[5 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh, acc[5 + n], 0, 0, 0);
                }
            } else if constexpr (V == 5) {
                const b16x8 c0 = __builtin_bit_cast(b16x8, a0), c1 = __builtin_bit_cast(b16x8, b0);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int i = 0; i < 10; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, c1, acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 10; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 10; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 10; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[i], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][15];
        if (s == 12345.f) dst[threadIdx.x] = s;
    }
}

static double rd(const std::string& p) {
    FILE* f = fopen(p.c_str(), "r");
    if (!f) return -1;
    double v = -1;
    if (fscanf(f, "%lf", &v) != 1) v = -1;
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    std::vector<std::string> hw;
    glob_t g;
    if (glob("/sys/class/drm/card*/device/hwmon/hwmon*", 0, nullptr, &g) == 0)
        for (size_t i = 0; i < g.gl_pathc; ++i) hw.push_back(g.gl_pathv[i]);
    float *src, *dst;
    hipMalloc(&src, 16384 * 4);
    hipMalloc(&dst, 4096);
    std::vector<float> h(16384);
    // half-precision-safe random-ish payload (pairs of f16 around 1)
    for (int i = 0; i < 16384; ++i) { unsigned short a = 0x3c00 + (i * 37 % 512), b = 0xbc00 + (i * 91 % 512); unsigned u = a | (unsigned)b << 16; h[i] = *reinterpret_cast<float*>(&u); }
    hipMemcpy(src, h.data(), 16384 * 4, hipMemcpyHostToDevice);
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    for (int v = 0; v <= 6; ++v) {
        const int iters = 4096;
        double flop_per_wave_iter = 0, lds_bytes_per_wave_iter = 0;
        switch (v) {
            case 0: flop_per_wave_iter = 64 * 2048.0; break;
            case 3: flop_per_wave_iter = 64 * 16384.0; break;
            case 4: lds_bytes_per_wave_iter = 14 * 1024.0; break;
            case 2: flop_per_wave_iter = 30 * 32768.0; lds_bytes_per_wave_iter = 14 * 1024.0; break;
            default: flop_per_wave_iter = 30 * 32768.0;
        }
        auto launch = [&]() {
            switch (v) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, src, dst, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, src, dst, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, src, dst, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, src, dst, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, src, dst, iters); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, src, dst, iters); break;
                default: hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, src, dst, iters);
            }
        };
        launch();
        hipDeviceSynchronize();
        std::atomic<bool> stop{false};
        std::vector<double> pw, ck;
        std::thread th([&]() {
            while (!stop.load()) {
                double bw = -1, bc = -1;
                for (auto& p : hw) {
                    double w = rd(p + "/power1_average");
                    if (w < 0) w = rd(p + "/power1_input");
                    if (w > bw) { bw = w; bc = rd(p + "/freq1_input"); }
                }
                if (bw > 0) { pw.push_back(bw * 1e-6); ck.push_back(bc * 1e-6); }
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
        });
        const auto t0 = std::chrono::steady_clock::now();
        int n = 0;
        double el = 0;
        do {
            for (int j = 0; j < 8; ++j) launch();
            hipDeviceSynchronize();
            n += 8;
            el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        } while (el < secs);
        stop = true;
        th.join();
        // second half of the samples (the power average lags)
        double w = 0, c = 0; int m = 0;
        for (size_t i = pw.size() / 2; i < pw.size(); ++i) { w += pw[i]; c += ck[i]; ++m; }
        if (m) { w /= m; c /= m; }
        const double waves = 256.0 * 8, tot_it = (double)n * iters;
        printf("variant %d: %.1f TFLOP/s  LDS %.1f TB/s  %.0f W  %.0f MHz  (%d launches, %.2f s)\n", v,
               flop_per_wave_iter * waves * tot_it / el / 1e12, lds_bytes_per_wave_iter * waves * tot_it / el / 1e12, w, c, n, el);
        fflush(stdout);
    }
    return 0;
}
