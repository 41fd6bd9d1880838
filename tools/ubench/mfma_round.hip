// How does v_mfma_f32_16x16x32_f16 round?  One wave, A = per-row constants, B = per-column constants, C = a big constant:
// D[i][j] = C + sum_k a[k] * b[k].  Cases probe (1) round-to-nearest vs truncation of the final sum, (2) whether small
// addends are truncated one by one when aligned to a big accumulator, (3) sign symmetry.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_round.hip -o /tmp/mfma_round
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using f4 = __attribute__((ext_vector_type(4))) float;

__global__ void k(const float* av, const float* bv, const float* cv, float* out, int ncase) {
    for (int c = 0; c < ncase; ++c) {
        h8 a, b;
        // lane holds 8 consecutive k of row (lane & 15), k group lane >> 4; all rows / columns alike
        for (int e = 0; e < 8; ++e) {
            const int kk = (threadIdx.x >> 4) * 8 + e;
            a[e] = (_Float16)av[c * 32 + kk];
            b[e] = (_Float16)bv[c * 32 + kk];
        }
        f4 acc = {cv[c], cv[c], cv[c], cv[c]};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        if (threadIdx.x == 0) out[c] = acc[0];
    }
}

int main() {
    const int N = 24;
    float a[N][32] = {}, b[N][32] = {}, c[N] = {};
    double exact[N];
    const char* what[N];
    auto fill = [&](int i, float C, int n, float av, float bvv, const char* w) {
        c[i] = C; for (int q = 0; q < n; ++q) { a[i][q] = av; b[i][q] = bvv; }
        exact[i] = (double)C + (double)n * av * bvv; what[i] = w;
    };
    fill(0, 16777216.f, 1, 1.f, 1.5f, "2^24 + 1.5 (RN: +2, trunc: +0)");
    fill(1, -16777216.f, 1, 1.f, -1.5f, "-2^24 - 1.5 (RN: -2, RZ: -0, floor: -2)");
    fill(2, 16777216.f, 1, 1.f, -0.5f, "2^24 - 0.5 (RN: 2^24, floor: 2^24 - 1)");
    fill(3, -16777216.f, 1, 1.f, 0.5f, "-2^24 + 0.5 (RN: -2^24, RZ: -2^24 + 1, floor: -2^24)");
    fill(4, 16777216.f, 32, 1.f, 0.75f, "2^24 + 32 x 0.75 = +24 (per-addend truncation: +0)");
    fill(5, 16777216.f, 32, 1.f, 0.0625f, "2^24 + 32 x 1/16 = +2");
    fill(6, 16777216.f, 3, 1.f, 0.5f, "2^24 + 3 x 0.5 = +1.5 (RN: +2)");
    fill(7, 16777216.f, 2, 1.f, 0.5f, "2^24 + 1 (tie: RNE +0)");
    fill(8, 16777218.f, 2, 1.f, 0.5f, "2^24 + 2 + 1 (tie: RNE +4)");
    fill(9, 1.f, 1, 0.000244140625f, 0.000244140625f, "1 + 2^-24 (tie -> 1)");
    fill(10, 1.f, 3, 0.000244140625f, 0.000244140625f, "1 + 3 x 2^-24 (RN: 1 + 2^-23 x ... )");
    fill(11, -1.f, 3, 0.000244140625f, -0.000244140625f, "-1 - 3 x 2^-24");
    fill(12, 16777216.f, 32, 1.f, -0.0625f, "2^24 + 32 x -1/16 = -2 (toward zero: 0; floor per addend: -8)");
    fill(13, -16777216.f, 32, 1.f, 0.0625f, "-2^24 + 32 x 1/16");
    fill(14, -16777216.f, 32, 1.f, -0.0625f, "-2^24 + 32 x -1/16");
    fill(15, 16777216.f, 32, 1.f, -0.125f, "2^24 + 32 x -1/8 = -4");
    fill(16, 16777216.f, 32, 1.f, 0.125f, "2^24 + 32 x 1/8 = +4");
    fill(17, 16777216.f, 32, 1.f, -0.25f, "2^24 + 32 x -1/4 = -8");
    fill(18, 16777216.f, 32, 1.f, 0.25f, "2^24 + 32 x 1/4 = +8");
    fill(19, 16777216.f, 32, 1.f, -0.375f, "2^24 + 32 x -3/8 = -12");
    fill(20, 16777216.f, 32, 1.f, 0.375f, "2^24 + 32 x 3/8 = +12");
    fill(21, 16777216.f, 32, 1.f, -0.001f, "2^24 + 32 x -0.001");
    fill(22, 0.f, 32, 1.f, 0.0625f, "0 + 32/16 = 2"); a[22][0] = 4096.f; b[22][0] = 4096.f; exact[22] += 16777216.0 - 0.0625;
    fill(23, 0.f, 32, 1.f, -0.0625f, "2^24 (as a product) + 31 x -1/16"); a[23][0] = 4096.f; b[23][0] = 4096.f; exact[23] += 16777216.0 + 0.0625;
    float *da, *db, *dc, *dout, out[N];
    hipMalloc(&da, sizeof a); hipMalloc(&db, sizeof b); hipMalloc(&dc, sizeof c); hipMalloc(&dout, sizeof out);
    hipMemcpy(da, a, sizeof a, hipMemcpyHostToDevice); hipMemcpy(db, b, sizeof b, hipMemcpyHostToDevice);
    hipMemcpy(dc, c, sizeof c, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dout, N);
    hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
    for (int i = 0; i < N; ++i)
        printf("%2d  %-60s exact %.10g  RN(fp32) %.10g  mfma %.10g  (mfma - C = %.10g)\n", i, what[i], exact[i], (double)(float)exact[i],
               (double)out[i], (double)out[i] - (double)c[i]);
    return 0;
}
