/*
 * sinddm_hip.h -- C ABI of the MI355X (gfx950) SinDDM hot-path library (libsinddm_hip.so).
 *
 * Drop-in boundary: the reference is pure Python/PyTorch, so the
 * "FFI" a maintainer binds is ctypes.  Each entry point below replaces a chain of
 * PyTorch library calls in the reference; the file:line it replaces is cited.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller
 *     (PyTorch-ROCm tensors) unless marked "host"; the library never allocates or
 *     frees device memory and keeps no mutable global state that a call's result depends
 *     on (the only cached value is the CU count per device id, queried once).  It reads no
 *     environment variable: kernel selection is fixed at compile time.  (The opt-in measurement
 *     hooks live in sinddm_hip_debug.h and are not part of this contract.)
 *   - all tensors are fp32, NCHW, contiguous.  Timesteps are int64.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); no
 *     host synchronisation inside.  Re-entrant across streams.
 *   - return value: 0 = ok; >0 = hipError_t from a launch; <0 = SINDDM_E_* below.
 */
#ifndef SINDDM_HIP_H
#define SINDDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SINDDM_ABI_VERSION 3    /* 3 (round 6): option bits in `dim`; packed-weight layout without the conv_h2 images; sinddm_debug_set_h2 removed */

/* Every `dim` argument below = SinDDMNet's width (reference SinDDM/models.py:86, main.py --dim) in its low 16 bits, plus
 * per-call option bits above them.  Options change which kernels a call launches, never a buffer layout or size. */
#define SINDDM_DIM_FP32_CONVS 0x10000  /* keep every 3x3 convolution on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32 Winograd
                                        * kernels) instead of the binary16 hi/lo kernel (conv_wh.h): for A/B measurements and
                                        * parity tests of the two paths; results agree to fp32 rounding (tests/test_gpu_h2.py) */

#define SINDDM_E_BADARG   (-1)  /* null pointer / non-positive size            */
#define SINDDM_E_BADSHAPE (-2)  /* dim/channels not supported by the kernels   */
#define SINDDM_E_WORKSPACE (-3) /* workspace too small (see *_workspace_bytes) */

/* ---- introspection --------------------------------------------------------------------- */
int sinddm_abi_version(void);

/* Number of fp32 elements of the flat parameter buffer of SinDDMNet(dim, channels=3,
 * multiscale=True) in nn.Module registration order (reference SinDDM/models.py:100-132;
 * key list in SURVEY.md 8(b)), and the offset of the idx-th tensor (0..n_tensors-1). */
int64_t sinddm_param_count(int dim);
int sinddm_param_tensors(int dim);
int64_t sinddm_param_offset(int dim, int idx);

/* Number of fp32 elements of the MFMA-ready packed weight image built by sinddm_pack_weights. */
int64_t sinddm_packed_count(int dim);

/* Bytes of scratch sinddm_net_forward needs for a (B,3,H,W) input. */
size_t sinddm_workspace_bytes(int dim, int B, int H, int W);

/* ---- network ---------------------------------------------------------------------------- */
/* Re-layout the 3x3 / 1x1 conv weights of the flat parameter buffer into the chunked
 * [co-block][ci-chunk][tap][ci][co] image the MFMA kernels stage into LDS.  Must be called
 * after every parameter update (load_state_dict, optimizer step). */
int sinddm_pack_weights(const float* params, float* packed, int dim, void* stream);

/* eps = SinDDMNet.forward(x, t, scale)          reference SinDDM/models.py:134-151
 *   params  flat parameter buffer (sinddm_param_count floats)
 *   packed  image made by sinddm_pack_weights from the same params
 *   x       (B,3,H,W)     t_dev (B,) int64 or NULL -> every sample uses t_host
 *   scale   the pyramid scale s (python int / 1-elem tensor in the reference, models.py:137)
 *   out     (B,3,H,W)     ws/ws_bytes scratch >= sinddm_workspace_bytes()               */
int sinddm_net_forward(const float* params, const float* packed, const float* x,
                       const int64_t* t_dev, int t_host, float scale, float* out,
                       int dim, int B, int H, int W, void* ws, size_t ws_bytes, void* stream);

/* The conditioning path alone: SinusoidalPosEmb(32) of t and of the scale, time_mlp, and every block's
 * time_reshape(mlp(GELU(cond)))            reference SinDDM/models.py:39-46,106-110,136-141 and :54-60,74-76.
 *   emb_out        (B,64)  [sin(t f)|cos(t f)|sin(s f)|cos(s f)]   or NULL
 *   cond_vec_out   (B,32)  time_mlp output                         or NULL
 *   block_bias_out (B,sinddm_cond_stride(dim)) per-sample bias each block adds after its depthwise conv
 *                  (l1: 3 | l2: dim/2 | l3: dim | l4: dim channels, concatenated)                          */
int sinddm_cond_embed(const float* params, const int64_t* t_dev, int t_host, float scale, int dim, int B,
                      float* emb_out, float* cond_vec_out, float* block_bias_out, void* stream);
int sinddm_cond_stride(int dim);

/* ---- diffusion elementwise --------------------------------------------------------------- */
/* out = sqrt_ac[t]*x0 + sqrt_1m_ac[t]*noise      reference SinDDM/models.py:570-576 (+extract,
 * functions.py:105-108).  If x_orig != NULL the training-time blur mix of models.py:583-585 is
 * fused in front:  x0 := gamma_row[t]*x0 + (1-gamma_row[t])*x_orig.
 * t_dev (B,) int64 or NULL -> t_host for all samples.  n = C*H*W elements per sample.        */
int sinddm_q_sample(const float* x0, const float* x_orig, const float* noise, float* out,
                    const float* tab_sqrt_ac, const float* tab_sqrt_1m_ac, const float* gamma_row,
                    const int64_t* t_dev, int t_host, int B, int64_t n, void* stream);

/* Per-step scalars of one reverse diffusion step (all samples share t, models.py:481,541). */
typedef struct sinddm_step_coefs {
    int mode;            /* 0: s==0 or !reblurring (DDPM posterior, models.py:322-330)
                            1: s>0, t>0  (re-blur mix, models.py:331-345,434-436)
                            2: s>0, t==0 (models.py:347-350)                                */
    int clip;            /* clip_denoised (models.py:440-442) */
    float sqrt_recip_ac_t, sqrt_recipm1_ac_t;      /* models.py:308-309 */
    float coef1_t, coef2_t;                        /* mode 0: posterior_mean_coef1/2 */
    float gamma_t, gamma_tm1;                      /* clamp(gammas[s-1],0,0.55)[t], [t-1] */
    float sqrt_ac_tm1, sqrt_ac_t, sqrt_1m_ac_t;    /* mode 1 */
    float sqrt_1m_ac_tm1_mvar;                     /* sqrt(1 - ac[t-1] - var)  (models.py:343) */
    float sigma;                                   /* [t!=0]*exp(0.5*logvar)   (models.py:459) */
} sinddm_step_coefs;

/* x_{t-1} = p_sample tail: predict_start_from_noise + p_mean_variance(normal branch) +
 * q_posterior + noise add.   reference SinDDM/models.py:306-352,433-459.
 * x_tilde = img_prev_upsample (NULL in mode 0).  n = total elements B*C*H*W.                 */
int sinddm_reverse_step(const float* x_t, const float* eps, const float* x_tilde,
                        const float* noise, float* out, const sinddm_step_coefs* coefs /*host*/,
                        int64_t n, void* stream);

/* A run of reverse steps of one scale WITHOUT returning to the host between them: for i in [0, n_steps):
 *   eps = SinDDMNet(x_i, t_list[i], scale);  x_{i+1} = p_sample tail(x_i, eps, x_tilde, z_i; coefs[i])
 * = the body of p_sample_loop / p_sample_via_scale_loop (reference SinDDM/models.py:462-487,501-547).  The N(0,1)
 * draws z_i of models.py:455 are generated INSIDE the step kernel (Philox4x32-10 + Box-Muller; stream = (seed,
 * stream_id0 + i, element index)) -- the reference never seeds its generator, so only the distribution is contract;
 * callers that must inject recorded noise use sinddm_net_forward + sinddm_reverse_step per step instead.
 *   x       (B,3,H,W) state in; x_alt same-size scratch: the states ping-pong, *result_in_alt tells where x_n is
 *   eps     (B,3,H,W) scratch;  coefs / t_list: HOST arrays of n_steps entries;  ws as for sinddm_net_forward  */
int sinddm_sample_chain(const float* params, const float* packed, float* x, float* x_alt, float* eps,
                        const float* x_tilde, const sinddm_step_coefs* coefs /*host*/, const int* t_list /*host*/,
                        int n_steps, float scale, uint64_t seed, uint64_t stream_id0, int dim, int B, int H, int W,
                        void* ws, size_t ws_bytes, void* stream, int* result_in_alt /*host*/);
/* The same with a second, caller-owned stream: runs whose launches carry only a few work items per CU (coarse pyramid
 * scales) are executed as TWO half-batches -- the chains of a batch are independent -- whose launches overlap on `stream`
 * and `aux_stream` (ordered against each other with events inside the call; on return both streams' work is ordered
 * before anything enqueued on `stream` afterwards).  Results are identical to sinddm_sample_chain: the noise is keyed on
 * the element's index inside the whole batch.  aux_stream = NULL: plain sinddm_sample_chain. */
int sinddm_sample_chain2(const float* params, const float* packed, float* x, float* x_alt, float* eps,
                         const float* x_tilde, const sinddm_step_coefs* coefs /*host*/, const int* t_list /*host*/,
                         int n_steps, float scale, uint64_t seed, uint64_t stream_id0, int dim, int B, int H, int W,
                         void* ws, size_t ws_bytes, void* stream, void* aux_stream, int* result_in_alt /*host*/);

/* out[i] ~ N(0,1) from the same counter-based generator (the sampler's initial / re-noise draws, models.py:467,518) */
int sinddm_normal_fill(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream);

/* Same step with the reference's ROI guidance folded in (roi_patch_modification, models.py:291-298, applied at
 * :430-431 when roi_guided_sampling and s < n_scales-1): the predicted clean image x_recon becomes
 * edit_w[p] * x_recon + edit_c[ch][p] before the re-blur mix and the clamps.  edit_w: HW floats, edit_c: C*HW
 * floats (shared by all B samples, like the reference's broadcast target patch). */
int sinddm_reverse_step_edit(const float* x_t, const float* eps, const float* x_tilde,
                             const float* noise, float* out, const sinddm_step_coefs* coefs /*host*/,
                             const float* edit_w, const float* edit_c, int B, int C, int HW, void* stream);

/* F.interpolate(in, size=(H,W), mode='bilinear') (align_corners=False)  models.py:567 */
int sinddm_upsample_bilinear(const float* in, float* out, int BC, int h, int w, int H, int W,
                             void* stream);

/* ---- training ------------------------------------------------------------------------------ */
/* Scratch for one training forward+backward of a (B,3,H,W) batch: saved activations (about
 * 1843 floats per pixel per sample at dim=160) + backward scratch. */
size_t sinddm_train_workspace_bytes(int dim, int B, int H, int W);

/* Transposed / tap-flipped weight images for the data-gradient convolutions (rebuild after every
 * parameter update, like sinddm_pack_weights). */
int64_t sinddm_packed_bwd_count(int dim);
int sinddm_pack_weights_bwd(const float* params, float* packed_bwd, int dim, void* stream);

/* Same as sinddm_net_forward, but keeps every activation the backward needs inside `ws`
 * (layout private to the library).  `ws` must stay untouched until sinddm_net_backward ran.
 * Replaces the autograd-recording forward of reference SinDDM/models.py:587,591. */
int sinddm_net_forward_train(const float* params, const float* packed, const float* x,
                             const int64_t* t_dev, int t_host, float scale, float* out,
                             int dim, int B, int H, int W, void* ws, size_t ws_bytes, void* stream);

/* Backward of SinDDMNet: given grad_out = dL/d eps (B,3,H,W), ACCUMULATES (+=) dL/d params into
 * grad_params (same flat layout as params) and, if grad_x != NULL, writes dL/dx.
 * Replaces loss.backward() through the net (reference SinDDM/functions.py:97-102, trainer.py:202). */
int sinddm_net_backward(const float* params, const float* packed, const float* packed_bwd,
                        const float* x, const float* grad_out, float* grad_params, float* grad_x,
                        int dim, int B, int H, int W, void* ws, size_t ws_bytes, void* stream);

/* loss_out[0] += mean(|noise - eps|)  (caller zeroes loss_out);  if grad_out != NULL:
 * grad_out = -sign(noise - eps)/n * grad_scale.           reference SinDDM/models.py:594 */
int sinddm_l1_loss_fwd_bwd(const float* noise, const float* eps, float* loss_out, float* grad_out,
                           int64_t n, float grad_scale, void* stream);

/* Fused optimizer / EMA over flat buffers of n floats.  mode bits: 1 = Adam update of p from g
 * (torch.optim.Adam defaults, reference trainer.py:134,208; step_size = lr/(1-beta1^k),
 * bc2_sqrt = sqrt(1-beta2^k)); 2 = ema := p (copy phase, trainer.py:156-158);
 * 4 = ema := ema_decay*ema + (1-ema_decay)*p (models.py:28-31). */
int sinddm_adam_ema_step(float* p, const float* g, float* m, float* v, float* ema, float step_size,
                         float beta1, float beta2, float eps, float bc2_sqrt, float ema_decay,
                         float reserved, int mode, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SINDDM_HIP_H */
