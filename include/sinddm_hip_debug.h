/*
 * sinddm_hip_debug.h -- measurement hooks of libsinddm_hip.so.  NOT part of the drop-in boundary (sinddm_hip.h):
 * these are the only entry points that keep process-global mutable state (a table of hipEvents -- nothing a result depends on), they are off
 * unless sinddm_prof_begin() was called, they are not thread-safe, and a production caller never needs them.
 * bench.py uses them for the `roofline` object (HIP events around every MFMA conv launch, on the launch stream).
 * The reference has no counterpart (it has no profiling).
 */
#ifndef SINDDM_HIP_DEBUG_H
#define SINDDM_HIP_DEBUG_H

#include <stddef.h>
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
 * 2*B*H*W*Cout*Cin*9 executed); 11 .. 14 = the Winograd 3x3 launches of ONE kernel generation (conv_wino, conv_wino2,
 * conv_wino3, conv_wino4).  reset = 0 keeps the records so that several kinds can be queried. */
int sinddm_prof_end3(int kind, double* ms_total, int64_t* launches, double* flops_total,
                     double* exec_flops_total, int reset);

/* Which kernel generation the dim -> dim 3x3 convolutions of a launch of this shape take on the current device:
 * 4 = conv_wino4 (F(2x4,3x3), one wave per SIMD), 3 = conv_wino3 (F(2x4,3x3)), 2 = F(2x2,3x3) kernels, 0 = direct
 * implicit GEMM; negative = SINDDM_E_*.  Lets a test assert that it exercises the kernel it means to. */
int sinddm_debug_conv_path(int dim, int B, int H, int W);

/* The same question for INFERENCE launches (sinddm_net_forward / sinddm_sample_chain: rows padded to 4 floats inside the
 * workspace): 8 = conv_wh (Winograd F(2x4), binary16 hi/lo frequency GEMMs: >= 12 items of 8x32 pixels x 80 channels per CU, images of >= 12 000 pixels),
 * else the value sinddm_debug_conv_path gives for the padded shape.  `dim` may carry SINDDM_DIM_FP32_CONVS (never 8 then). */
int sinddm_debug_infer_path(int dim, int B, int H, int W);

/* ... and for TRAINING launches (sinddm_net_forward_train / sinddm_net_backward: plain rows, no padding): 8 = the forward
 * 3x3 convs and both data-gradient convs of the dim -> dim blocks take conv_wh (same rule as inference; needs W % 4 == 0),
 * else the value of sinddm_debug_conv_path.  With 8 the 3x3 weight gradients of those convs run on the binary16 pipe too
 * (wgrad_wh.h); the 1x1 / depthwise / first-conv weight gradients stay fp32. */
int sinddm_debug_train_path(int dim, int B, int H, int W);

/* ONE SinDDMConvBlock (l = 0..3 of the plan of SinDDMNet(dim); reference SinDDM/models.py:51-80) forward + backward on
 * its own: x (B, C_in, H, W), cond_bias (B, C_in) = the block's per-sample condition (time_reshape(mlp(cond)),
 * models.py:74-76), grad_y (B, C_out, H, W).  Writes y, grad_x (may be NULL), dcond (B, C_in) = gradient of cond_bias, and
 * ADDS the block's conv / depthwise weight and bias gradients into grad_params (flat layout of sinddm_param_offset).
 * ws: a training workspace (sinddm_train_workspace_bytes).  For block-level parity tests; training never calls it. */
int sinddm_debug_block_train(const float* params, const float* packed, const float* packed_bwd, int dim, int l,
                             const float* x, const float* cond_bias, const float* grad_y, float* y, float* grad_x,
                             float* grad_params, float* dcond, int B, int H, int W, void* ws, size_t ws_bytes,
                             void* stream);

/* Host-only (no device call): the workgroup -> (slab, pixel split) table wgrad_wino_kernel would be launched with for a
 * Cin -> Cout 3x3 weight gradient over `ntiles` 4x16-pixel tiles on `ncu` compute units.  wg_out (wg_cap entries) receives
 * slab << 16 | split per workgroup id, splits_out (splits_cap entries) the number of pixel splits of each (co block, ci
 * block) slab.  Returns the number of workgroups (> 0) or a negative SINDDM_E_* code (BADARG when a buffer is too small,
 * BADSHAPE when the launch table cannot describe the shape -- the library then uses its direct kernels).  For tests. */
int sinddm_debug_wgrad_map(int Cin, int Cout, int64_t ntiles, int ncu, uint32_t* wg_out, int wg_cap, int32_t* splits_out,
                           int splits_cap);

#ifdef __cplusplus
}
#endif
#endif /* SINDDM_HIP_DEBUG_H */
