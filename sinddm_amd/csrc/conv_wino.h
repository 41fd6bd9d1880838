// Winograd F(2x2, 3x3) 3x3 convolution on the gfx950 fp32 matrix cores.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, d = its 4x4 input patch
//
// 16 independent "frequency" GEMMs  M_xi[co][tile] = sum_ci U_xi[co][ci] * V_xi[ci][tile]  replace the
// 9-tap implicit GEMM: 16 multiplies per 4 outputs instead of 36 (2.25x fewer MFMAs), all in fp32.
// Mapping: a workgroup = 16 waves owns an 8x32 pixel tile (4x16 = 64 output 2x2-tiles) for MT*16 output
// channels; WAVE xi owns frequency xi = (i,j): MT x 4 accumulator tiles (M = co, N = 16 tiles of one
// tile-row, K = ci, v_mfma_f32_16x16x4_f32).
//   * U (pre-transformed weights, pack kernel) is laid out per (co-block, chunk, xi) in exactly the MFMA
//     A-operand register image, so a wave streams its 20 registers per 16-channel chunk straight from
//     L2 with fully coalesced loads -- weights never touch LDS (no wave shares them).
//   * the RAW input halo tile (16 channels x 10 x 34) is LDS-DMA'd (double buffered, one barrier per chunk);
//     every wave builds its own V_xi operand on the fly: 4 LDS reads + 3 FMAs per value (B^T has two +-1
//     entries per row).  Plane stride 341 (odd) keeps the stride-2 lane pattern bank-conflict-free.
//   * the output transform A^T M A needs all 16 frequencies of a (co, tile): one 16-channel M tile at a time
//     goes through LDS ([xi][co][tile]), then thread (co, tile) produces its 2x2 pixels, applies the fused
//     epilogue (bias, GELU / GELU'(u), residual, pre-activation save) and stores.
// Same ConvArgs / epilogue contract as conv_mfma.h; 1x1 residual projections are not fused here (the
// caller runs them through the direct kernel first and passes the result as `resid`).
#pragma once
#include "conv_mfma.h"
#include "conv1x1.h"
// conv_wino_launch() runs the second-generation kernel of conv_wino2.h (two 4-wave workgroups per CU, wave = frequency row).
// (The first-generation 16-wave kernel and its -DSINDDM_WINO_V2=0 build are archived: tools/variants/conv_wino_gen1.h and
// `git show 6674de8:sinddm_amd/csrc/conv_wino.h` -- product sources include nothing from tools/.)
#define SINDDM_WINO_V2 1
#include "conv_wino2.h"

namespace sinddm {

inline int device_cu_count() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

inline int conv_wino_launch(const ConvArgs& a_in, int mt, hipStream_t st) {
    // (the kernel stages a wave's four channel planes of a chunk through one descriptor: callers route convs with
    // C_in % 4 != 0 -- dim = 10, 20, 28 ... -- to the direct kernel)
    if (a_in.Cin % 4 != 0) return SINDDM_E_BADSHAPE;
    return conv_wino2_launch(a_in, mt, st);
}

inline bool wino_enabled() { return SINDDM_CONV_WINO != 0; }

}  // namespace sinddm
