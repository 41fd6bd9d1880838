// Is the rounding of v_mfma_f32_16x16x32_f16 biased?  Random binary16 operands, fp32 accumulator input, result against the
// exact sum (double holds it exactly enough): mean and rms of (mfma - exact) in units of the result's ulp, per scenario.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_bias.hip -o tools/ubench/mfma_bias
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using f4 = __attribute__((ext_vector_type(4))) float;

// A[t][16][32], B[t][16][32] (row-major over k), C[t][16][16] -> D[t][16][16]
__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D, int chain) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const int r = lane & 15, kg = lane >> 4;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[(t * 16 + r) * 32 + kg * 8 + e]; b[e] = B[(t * 16 + r) * 32 + kg * 8 + e]; }
    f4 acc;
    // D layout of 16x16x32: lane holds column (lane & 15), rows 4 * (lane >> 4) + 0..3
    for (int i = 0; i < 4; ++i) acc[i] = C[(t * 16 + 4 * kg + i) * 16 + r];
    for (int c = 0; c < chain; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(t * 16 + 4 * kg + i) * 16 + r] = acc[i];
}

// the fp32 pipe for comparison: v_mfma_f32_16x16x4_f32, A[t][16][4], B[t][16][4] (lane: row lane & 15, k = lane >> 4)
__global__ void k32(const float* A, const float* B, const float* C, float* D) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const int r = lane & 15, kg = lane >> 4;
    f4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = C[(t * 16 + 4 * kg + i) * 16 + r];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(t * 16 + r) * 4 + kg], B[(t * 16 + r) * 4 + kg], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(t * 16 + 4 * kg + i) * 16 + r] = acc[i];
}

static double rnd() { return (double)rand() / RAND_MAX; }
static double gauss() { return std::sqrt(-2 * std::log(rnd() + 1e-300)) * std::cos(6.283185307179586 * rnd()); }

int main() {
    const int T = 4096;
    std::vector<_Float16> A(T * 16 * 32), B(T * 16 * 32);
    std::vector<float> C(T * 256), D(T * 256);
    _Float16 *dA, *dB; float *dC, *dD;
    (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dC, C.size() * 4); (void)hipMalloc(&dD, D.size() * 4);
    const char* names[] = {"products ~N(0,1), C = 0", "C ~ +64 sigma", "C ~ -64 sigma", "C ~ N(0, 8 sigma)",
                           "16 big + 16 small (2^-11) products, C ~ N(0, 8 sigma)", "all 32 small (2^-11), C ~ N(0, 8 sigma) (the [U_lo] MFMA)",
                           "all 32 small (2^-11), C ~ +8 sigma", "all 32 small (2^-11), C ~ -8 sigma"};
    for (int sc = 0; sc < 8; ++sc) {
        for (size_t i = 0; i < A.size(); ++i) {
            const int kk = (int)(i % 32);
            double sa = 1.0;
            if (sc == 4 && kk >= 16) sa = 1.0 / 2048;
            if (sc >= 5) sa = 1.0 / 2048;
            A[i] = (_Float16)(gauss() * 37.0 * sa);
            B[i] = (_Float16)(gauss() * 51.0);
        }
        const double sig = 37.0 * 51.0 * std::sqrt(32.0);
        for (size_t i = 0; i < C.size(); ++i) {
            double c = 0;
            if (sc == 1) c = 64 * sig * (1 + 0.3 * rnd());
            if (sc == 2) c = -64 * sig * (1 + 0.3 * rnd());
            if (sc == 3 || sc == 4 || sc == 5) c = gauss() * 8 * sig;
            if (sc == 6) c = 8 * sig * (1 + 0.3 * rnd());
            if (sc == 7) c = -8 * sig * (1 + 0.3 * rnd());
            C[i] = (float)c;
        }
        (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(T), dim3(64), 0, 0, dA, dB, dC, dD, 1);
        (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        double m = 0, m2 = 0, mrn = 0, m2rn = 0, msgn = 0; long n = 0;
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double ex = C[t * 256 + i * 16 + j];
                    for (int q = 0; q < 32; ++q) ex += (double)(float)A[(t * 16 + i) * 32 + q] * (double)(float)B[(t * 16 + j) * 32 + q];
                    const float rn = (float)ex;
                    int e; std::frexp(std::fabs((double)rn) > 0 ? (double)rn : 1.0, &e);
                    const double ulp = std::ldexp(1.0, e - 24);
                    const double err = ((double)D[t * 256 + i * 16 + j] - ex) / ulp, errn = ((double)rn - ex) / ulp;
                    m += err; m2 += err * err; mrn += errn; m2rn += errn * errn; msgn += err * (ex > 0 ? 1 : -1); ++n;
                }
        printf("%-75s mfma: mean %+8.4f  (x sign(result): %+8.4f)  rms %7.4f ulp   | ideal RN: mean %+8.4f rms %7.4f\n", names[sc], m / n, msgn / n,
               std::sqrt(m2 / n), mrn / n, std::sqrt(m2rn / n));
    }
    // fp32 pipe: 4 products of fp32 operands (24 x 24 bits: not exact in fp32) + C
    {
        std::vector<float> A32(T * 64), B32(T * 64);
        float *dA32, *dB32;
        (void)hipMalloc(&dA32, A32.size() * 4); (void)hipMalloc(&dB32, B32.size() * 4);
        for (int sc = 0; sc < 3; ++sc) {
            for (size_t i = 0; i < A32.size(); ++i) { A32[i] = (float)(gauss() * 37.0); B32[i] = (float)(gauss() * 51.0); }
            const double sig = 37.0 * 51.0 * 2.0;
            for (size_t i = 0; i < C.size(); ++i) C[i] = (float)(sc == 0 ? gauss() * 8 * sig : (sc == 1 ? 1 : -1) * 8 * sig * (1 + 0.3 * rnd()));
            (void)hipMemcpy(dA32, A32.data(), A32.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB32, B32.data(), B32.size() * 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k32, dim3(T), dim3(64), 0, 0, dA32, dB32, dC, dD);
            (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            double m = 0, m2 = 0; long n = 0;
            for (int t = 0; t < T; ++t)
                for (int i = 0; i < 16; ++i)
                    for (int j = 0; j < 16; ++j) {
                        long double ex = C[t * 256 + i * 16 + j];
                        for (int q = 0; q < 4; ++q) ex += (long double)A32[(t * 16 + i) * 4 + q] * (long double)B32[(t * 16 + j) * 4 + q];
                        const float rn = (float)ex;
                        int e; std::frexp(std::fabs((double)rn) > 0 ? (double)rn : 1.0, &e);
                        const double ulp = std::ldexp(1.0, e - 24);
                        const double err = (double)(((long double)D[t * 256 + i * 16 + j] - ex) / ulp);
                        m += err; m2 += err * err; ++n;
                    }
            printf("v_mfma_f32_16x16x4_f32, C %-50s mean %+8.4f  rms %7.4f ulp\n", sc == 0 ? "~ N(0, 8 sigma)" : (sc == 1 ? "~ +8 sigma" : "~ -8 sigma"), m / n, std::sqrt(m2 / n));
        }
    }
    return 0;
}
