#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "plan.h"
#include "../../include/sinddm_hip.h"
#include "../../include/sinddm_hip_debug.h"

namespace sinddm {

// Kernel-selection switches are COMPILE-TIME macros (A/B builds: tools/build_variant.sh <name> -D...): the shipped
// library reads no environment variable, so nothing outside the caller's arguments can change which kernel a call runs.
#ifndef SINDDM_WINO_V3        // 1: inference 3x3 convs of big launches on the F(2x4,3x3) kernel (conv_wino3.h)
#define SINDDM_WINO_V3 1
#endif
#ifndef SINDDM_V3_MIN_ITEMS_PER_CU   // launches with at least this many (tile, 80-channel block) items per CU take conv_wino3.h
#define SINDDM_V3_MIN_ITEMS_PER_CU 1
#endif
#ifndef SINDDM_WINO_V4        // 1: launches with several 8x32 items per CU on the one-wave-per-SIMD F(2x4,3x3) kernel (conv_wino4.h)
#define SINDDM_WINO_V4 1
#endif
#ifndef SINDDM_CONV_WINO      // 1: 3x3 convs (C_in >= 8) on the Winograd kernel, 0: direct implicit-GEMM kernel
#define SINDDM_CONV_WINO 1
#endif
#ifndef SINDDM_CONV_VAR       // -1: pick 4- or 8-wave direct-conv workgroups per launch; 4 / 8: force
#define SINDDM_CONV_VAR (-1)
#endif
#ifndef SINDDM_CONV_C3        // 1: dedicated VALU kernel for the C_in = 3 conv
#define SINDDM_CONV_C3 1
#endif
#ifndef SINDDM_WGRAD_WINO     // 1: Winograd-domain 3x3 weight gradient
#define SINDDM_WGRAD_WINO 1
#endif
#ifndef SINDDM_DWG_ROWS       // 1: depthwise weight gradient of wide aligned images on the register-window kernel
#define SINDDM_DWG_ROWS 1
#endif
#ifndef SINDDM_WGRAD_WH       // 1: ... and on the binary16 matrix pipe (wgrad_wh.h) when the operands' running maxima exist (training on conv_wh)
#define SINDDM_WGRAD_WH 1
#endif
#ifndef SINDDM_WGRAD_WIDE     // 1: W % 4 == 0 launches of the Winograd weight gradient on wgrad_wino_wide_kernel (16-byte DMA)
#define SINDDM_WGRAD_WIDE 1
#endif
#ifndef SINDDM_WGRAD_W3       // 1: 80x80-slab direct 3x3 weight gradient where the Winograd one does not apply
#define SINDDM_WGRAD_W3 1
#endif
#ifndef SINDDM_WGRAD_W1       // 1: 80x80-slab 1x1 weight gradient
#define SINDDM_WGRAD_W1 1
#endif
#ifndef SINDDM_WGRAD_STAGE    // 1: coalesced [co][tap][ci] staging slab for the 3x3 weight-gradient atomics
#define SINDDM_WGRAD_STAGE 1
#endif
#ifndef SINDDM_WGRAD_ABL      // timing ablation bits of the weight-gradient kernels (results WRONG when != 0)
#define SINDDM_WGRAD_ABL 0
#endif

// Barrier that also publishes this wave's LDS-DMA (buffer_load ... lds).  __syncthreads() is a WORKGROUP-scope fence, for
// which the gfx950 memory model waits on lgkmcnt only: an LDS-DMA counts on vmcnt and may still be in flight when the
// other waves pass the barrier and read its destination.  (The compiler does wait before the issuing wave's OWN ds_reads
// of the destination, which hid the hole wherever a DMA had a whole tile of slack; wgrad_wino_wide_kernel's 16-channel
// slabs, whose tiles are shorter than an HBM round trip, showed it as run-to-run differences of 1e-3.)
__device__ __forceinline__ void dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// split16.h / conv_wh.h: running max |x| of the tensors the binary16 kernel reads, PER SAMPLE ([B][AMAX_STRIDE] floats, slot 2 l + i =
// input of conv i of block l): a sample's scale -- and so its result, bit for bit -- does not depend on what else is in the batch
constexpr int AMAX_STRIDE = 8;

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

#define SINDDM_LAUNCH_CHECK()                          \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)

// erf for the GELU epilogues: branch-free Abramowitz-Stegun 7.1.26 evaluated in fp32
// (|error| <= 6e-7 absolute, measured against double precision; the resulting GELU is within 5e-7 of the
// double-precision GELU, tighter than an fp32 evaluation of 0.5*x*(1+erf(x/sqrt2)) with a 1-ulp erff).
// A dozen instructions without divergence instead of libm's two-range erff (~50 instructions, both
// ranges executed under divergence) -- the conv epilogues are VALU-bound.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    // v_rcp_f32 (1 ulp): the correctly-rounded __frcp_rn expands to a 10-instruction division sequence, which was a
    // quarter of the VALU work of the GELU epilogues; the extra ulp is far below the approximation's 6e-7
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float r = fmaf(-poly, __expf(-ax * ax), 1.0f);
    return copysignf(r, x);
}

// exact-erf GELU (nn.GELU() default), reference SinDDM/models.py:55,64,108, two values at a time on the packed fp32 VALU:
// 0.5 v (1 + erf(v / sqrt 2)) with erf(x) = x P(x^2) / Q(x^2) (the odd 13 / even 8 rational of Eigen's and XLA's fp32 erf,
// argument clamped to +-4, where fp32 erf is +-1), rescaled to v and to Q(0) = 1:
//     gelu(v) = 0.5 v (Q(s) + c P(s)) / Q(s),   c = clamp(v, +-4 sqrt 2),  s = c^2
// One reciprocal, no exponential: per value 7.5 packed-rate instructions + v_med3 + v_rcp instead of the 13 + v_rcp +
// v_exp of the Abramowitz-Stegun form above (the GELU epilogues of the 3x3 convs are VALU-bound: 47 -> ~30 cycles per
// value).  |error| <= 1.7e-6 absolute (at |v| ~ 5; 1.1e-6 inside |v| < 4), 5e-8 rel-L2 over [-12, 12] against float64
// (tools/gelu_fit.py); beyond the clamp the quotient is 2 - 2e-7 / 5e-7, i.e. gelu = v / a relative 2.5e-7 of |v|.
// Every kernel takes THIS sequence (the scalar form is the packed one's first half): the 3x3 kernel families stay
// bit-identical to each other.
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
    constexpr float CL = 5.656854249f;
    const f32x2 c{__builtin_amdgcn_fmed3f(v.x, -CL, CL), __builtin_amdgcn_fmed3f(v.y, -CL, CL)};
    const f32x2 s = c * c;
    f32x2 p = s * 2.111493341e-10f + -4.291980815e-08f;
    p = p * s + 6.509268587e-06f;
    p = p * s + 3.527237568e-04f;
    p = p * s + 9.108418599e-03f;
    p = p * s + 7.323013246e-02f;
    p = p * s + 7.978845239e-01f;
    f32x2 q = s * 6.382026913e-05f + 1.869768254e-03f;
    q = q * s + 2.949277498e-02f;
    q = q * s + 2.584459782e-01f;
    q = q * s + 1.0f;
    const f32x2 n = p * c + q;
    const f32x2 r{__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};
    // (the multiplier is v itself above the lower clamp and the clamp below it: for v < -4 sqrt 2 the quotient is 5e-7, not 0,
    // and 0.5 v x 5e-7 would grow with |v|; held at the clamp the tail is the constant -1.4e-6, inside the bound above, as
    // nn.GELU's tail decays to 0 -- ADVICE r4.  gelu_erf_grad keeps the Abramowitz-Stegun form: the two differ by <= 1e-6.)
    const f32x2 h = 0.5f * __builtin_elementwise_max(v, f32x2{-CL, -CL});
    return (h * n) * r;
}
__device__ __forceinline__ float gelu_erf(float v) { return gelu_erf2(f32x2{v, v}).x; }
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
    const f32x2 a = gelu_erf2(f32x2{v.x, v.y}), b = gelu_erf2(f32x2{v.z, v.w});
    return f32x4{a.x, a.y, b.x, b.y};
}

__device__ __forceinline__ float gelu_erf_grad(float v) {
    // d/dv [0.5 v (1+erf(v/sqrt2))] = 0.5(1+erf(v/sqrt2)) + v * exp(-v^2/2)/sqrt(2 pi).
    // erf_fast's exp(-x^2) at x = v/sqrt2 IS the exp(-v^2/2) of the density: evaluated once for both terms.
    const float ax = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __expf(-0.5f * v * v);
    const float erfv = copysignf(fmaf(-poly, e, 1.0f), v);
    return fmaf(v, 0.39894228040143267794f * e, 0.5f * (1.0f + erfv));
}

// ---- geometry of the implicit-GEMM 3x3 conv tile (shared by forward / dgrad) ----
constexpr int CONV_THREADS = 256;   // 4 waves, one per SIMD; 2-3 workgroups per CU
constexpr int CONV_NT = 4;          // 16-pixel N tiles per wave
constexpr int CONV_WN = 4;          // waves along N
constexpr int CONV_TW = 32;         // tile width  (pixels)
constexpr int CONV_TPR = CONV_TW / 16;                       // N tiles per tile row
constexpr int CONV_TH = CONV_WN * CONV_NT / CONV_TPR;        // tile height = 8
constexpr int CONV_RS = CONV_TW + 2;                         // LDS row stride incl. halo
constexpr int CONV_HR = CONV_TH + 2;                         // LDS rows incl. halo
constexpr int CONV_PS = ((CONV_HR * CONV_RS - 16 + 31) / 32) * 32 + 16;  // plane stride == 16 mod 32
constexpr int CONV_IN_ELEMS = KC * CONV_HR * CONV_RS;
constexpr int CONV_IREGS = (CONV_IN_ELEMS + CONV_THREADS - 1) / CONV_THREADS;

}  // namespace sinddm
