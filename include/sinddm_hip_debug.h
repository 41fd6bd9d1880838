/*
 * sinddm_hip_debug.h -- measurement hooks of libsinddm_hip.so.  NOT part of the drop-in boundary (sinddm_hip.h):
 * these are the only entry points that keep process-global mutable state (a table of hipEvents), they are off
 * unless sinddm_prof_begin() was called, they are not thread-safe, and a production caller never needs them.
 * bench.py uses them for the `roofline` object (HIP events around every MFMA conv launch, on the launch stream).
 * The reference has no counterpart (it has no profiling).
 */
#ifndef SINDDM_HIP_DEBUG_H
#define SINDDM_HIP_DEBUG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*  * Between prof_begin and prof_end every MFMA conv launch is bracketed by hipEvents on the stream it
 * is launched on; prof_end synchronises those events and returns the summed kernel time (ms), the
 * number of launches and their algorithmic FLOPs (2*B*H*W*Cout*(9*Cin + Cin2)).  Process-global,
 * not thread-safe; off by default.  No reference counterpart (the reference has no profiling). */
int sinddm_prof_begin(void);
int sinddm_prof_end(double* conv_ms_total, int64_t* conv_launches, double* conv_flops_total);
/* same, plus the FLOPs the matrix cores actually executed (Winograd F(2x4,3x3) launches -- conv_wino3/4/5 -- execute
 * 24/72 of their algorithmic FLOPs, F(2x2,3x3) ones -- conv_wino2 -- 16/36) */
int sinddm_prof_end2(double* conv_ms_total, int64_t* conv_launches, double* conv_flops_total,
                     double* conv_exec_flops_total);
/* Same, restricted to one kernel family: kind 0 = all, 1 = Winograd 3x3 (conv_wino2/3/4/5_kernel), 2 = 1x1 convs,
 * 3 = direct 3x3 (conv_mfma_dma_kernel), 4 = Winograd-domain 3x3 weight gradient (wgrad_wino_kernel, F(2x2): 16/36 of
 * 2*B*H*W*Cout*Cin*9 executed).  reset = 0 keeps the records so that several kinds can be queried. */
int sinddm_prof_end3(int kind, double* ms_total, int64_t* launches, double* flops_total,
                     double* exec_flops_total, int reset);

/* Host-only (no device call): the workgroup -> (slab, pixel split) table wgrad_wino_kernel would be launched with for a
 * Cin -> Cout 3x3 weight gradient over `ntiles` 4x16-pixel tiles on `ncu` compute units.  wg_out (>= 512 entries) receives
 * slab << 16 | split per workgroup id, splits_out (>= 256) the number of pixel splits of each (co block, ci block) slab.
 * Returns the number of workgroups (> 0) or a negative SINDDM_E_* code.  For tests of the load balance. */
int sinddm_debug_wgrad_map(int Cin, int Cout, int64_t ntiles, int ncu, uint32_t* wg_out, int32_t* splits_out);

#ifdef __cplusplus
}
#endif
#endif /* SINDDM_HIP_DEBUG_H */
