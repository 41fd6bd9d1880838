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
// SINDDM_WINO_V2 = 1 (default): conv_wino_launch() runs the second-generation kernel of conv_wino2.h (two 4-wave
// workgroups per CU, wave = frequency row); 0 keeps this file's 16-wave kernel (A/B builds).  The packed weight
// layout (pack kind 3) follows the same switch.
#ifndef SINDDM_WINO_V2
#define SINDDM_WINO_V2 1
#endif
#include "conv_wino2.h"

namespace sinddm {

#if !SINDDM_WINO_V2
#include "../../tools/variants/conv_wino_gen1.h"     // (archived first-generation kernel: variant builds only)
#endif

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
#if SINDDM_WINO_V2
    // (the second-generation kernel stages a wave's four channel planes of a chunk through one descriptor; the packed
    // Winograd image has ITS layout, so the first-generation kernel below must not see it: callers route convs with
    // C_in % 4 != 0 -- dim = 10, 20, 28 ... -- to the direct kernel)
    if (a_in.Cin % 4 != 0) return SINDDM_E_BADSHAPE;
    return conv_wino2_launch(a_in, mt, st);
#else
    ConvArgs a = a_in;
    const int ntr = wino_ntr();
    ConvProfiler& prof = conv_profiler();
    const bool rec = prof.on && prof.used < ConvProfiler::MAXREC;
    if (rec) {
        while (prof.created <= prof.used) {
            (void)hipEventCreate(&prof.ev[2 * prof.created]);
            (void)hipEventCreate(&prof.ev[2 * prof.created + 1]);
            ++prof.created;
        }
        (void)hipEventRecord(prof.ev[2 * prof.used], st);
    }
    const int TH = 2 * ntr;
    a.tilesX = (a.W + WN_TW - 1) / WN_TW;
    a.tilesY = (a.H + TH - 1) / TH;
    a.ntiles = a.B * a.tilesX * a.tilesY;
    a.tiles_per_xcd = (a.ntiles + 7) / 8;
    // persistent launch: one 16-wave workgroup per CU (its registers and LDS fill the CU), each walking its
    // share of the XCD's work items
    const int ipx = a.tiles_per_xcd * a.coblks;                  // work items per XCD
    int wpx = device_cu_count() / 8;                             // workgroups per XCD
    if (wpx < 1) wpx = 1;
    if (wpx > ipx) wpx = ipx;
    const unsigned grid = (unsigned)(wpx * 8);
    switch (mt) {
        case 5: conv_wino_launch_t<5>(a, grid, ipx, wpx, ntr, st); break;
        case 2: conv_wino_launch_t<2>(a, grid, ipx, wpx, ntr, st); break;
        case 1: conv_wino_launch_t<1>(a, grid, ipx, wpx, ntr, st); break;
        default: return SINDDM_E_BADSHAPE;
    }
    if (rec) {
        (void)hipEventRecord(prof.ev[2 * prof.used + 1], st);
        const double fl = 2.0 * a.B * a.H * a.W * (double)a.Cout * 9.0 * a.Cin;   // algorithmic (direct-conv) FLOPs
        prof.note(1, fl, fl * (16.0 / 36.0), 1);
    }
    SINDDM_LAUNCH_CHECK();
    return 0;
#endif
}

inline bool wino_enabled() { return SINDDM_CONV_WINO != 0; }

}  // namespace sinddm
