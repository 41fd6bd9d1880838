// Shared pieces of the binary16 hi/lo scheme (conv_wh.h, wgrad_wh.h): an fp32 operand a is carried as two binary16 pieces
// a_hi = rn16(2^s a), a_lo = rn16(2^s a - a_hi) (22 significand bits) with an EXACT power-of-two scale 2^s that the epilogue
// removes again:
//   * weights: per output channel, chosen when the weights are packed;
//   * activations: per SAMPLE, from the running max |x| of the sample's tensor (`amax_in[b * AMAX_STRIDE]`, a device scalar
//     its producer kernel maintains with one guarded atomicMax per wave).  Per sample, so that a chain's numbers do not
//     depend on the rest of its batch (shards of a multi-GPU job reproduce the single-GPU run).
// Pieces that fall below the binary16 normal range lose at most 2^-25 absolute = 2^-35 of the tensor's max (measured on
// channels 2^16 apart: tests/test_gpu_h2.py::test_gate_heavy_tailed_channel_gains).
// (The direct implicit-GEMM kernel these helpers were first written for, conv_h2.h, is archived in tools/variants/.)
#pragma once
#include "conv_mfma.h"
#include "conv_wino2.h"

namespace sinddm {

using h16x8 = __attribute__((ext_vector_type(8))) _Float16;
using h16x2 = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int H2_TARGET_EXP = 13;                     // scaled max |w| in [2^13, 2^14)

// exact power of two 2^s as a float, s in [-126, 127]
__host__ __device__ __forceinline__ float h2_pow2(int s) {
    union { unsigned u; float f; } c;
    c.u = (unsigned)(s + 127) << 23;
    return c.f;
}
// shift s such that m * 2^s lies in [2^13, 2^14); 0 for m = 0 / non-finite
__device__ __forceinline__ int h2_shift_for(float m) {
    const unsigned bits = __float_as_uint(m) & 0x7fffffffu;
    const int e = (int)(bits >> 23);
    if (e == 0 || e == 255) return 0;
    int s = H2_TARGET_EXP - (e - 127);
    return s > 100 ? 100 : (s < -100 ? -100 : s);
}

// running max |x| of a tensor: one guarded atomic per wave (non-negative floats order like their bit patterns)
__device__ __forceinline__ void amax_publish(float m, float* slot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) {
        const unsigned bits = __float_as_uint(m);
        unsigned* s = reinterpret_cast<unsigned*>(slot);
        if (bits > __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(s, bits);
    }
}

}  // namespace sinddm
