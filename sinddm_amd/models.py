"""SinDDMNet and MultiScaleGaussianDiffusion backed by the gfx950 HIP library.

Same Python surface as the reference (SinDDM/models.py): constructor signatures, method names,
state-dict keys and buffer names are kept so `main.py` / `MultiscaleTrainer` and existing
checkpoints work unchanged, while every per-step tensor op runs in libsinddm_hip.so:

  SinDDMNet.forward          -> sinddm_net_forward        (models.py:134-151)
  q_sample / p_losses mix    -> sinddm_q_sample           (models.py:570-590)
  p_sample tail              -> sinddm_reverse_step       (models.py:306-352,433-459)
  inter-scale upsample       -> sinddm_upsample_bilinear  (models.py:567)

There is no CPU / eager-PyTorch fallback: tensors must live on a ROCm device and the shared
library must be built, otherwise calls raise.
"""
from __future__ import annotations

import ctypes as C
import math
from functools import partial
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .functions import cosine_beta_schedule, default, exists, extract, noise_like
from .synth import net_param_shapes


# --------------------------------------------------------------------------------------------
# EMA (reference models.py:18-31) -- the trainer uses the fused kernel; this stays for API parity
# --------------------------------------------------------------------------------------------
class EMA:
    def __init__(self, beta):
        self.beta = beta

    def update_model_average(self, ma_model, current_model):
        for cur, ma in zip(current_model.parameters(), ma_model.parameters()):
            ma.data = self.update_average(ma.data, cur.data)

    def update_average(self, old, new):
        if old is None:
            return new
        return old * self.beta + (1 - self.beta) * new


class SinusoidalPosEmb(nn.Module):
    """[sin(x f_i) | cos(x f_i)] (models.py:34-46).  Exposed for API parity; SinDDMNet computes the
    embedding inside its conditioning kernel."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        half = self.dim // 2
        k = math.log(10000) / (half - 1)
        f = torch.exp(torch.arange(half, device=x.device) * -k)
        arg = x[:, None] * f[None, :]
        return torch.cat((arg.sin(), arg.cos()), dim=-1)


# --------------------------------------------------------------------------------------------
# scratch memory shared by all nets on a device (PyTorch = allocator only)
# --------------------------------------------------------------------------------------------
_WS: Dict[Tuple[str, int], torch.Tensor] = {}


def _workspace(device: torch.device, nbytes: int, tag: str = "fwd") -> torch.Tensor:
    device = torch.device(device)
    key = (tag, device.index if device.index is not None else torch.cuda.current_device())
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = None
        _WS.pop(key, None)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


_AUX: Dict[int, "torch.cuda.Stream"] = {}


def _aux_stream(device: torch.device) -> int:
    """A second stream per device for sinddm_sample_chain2 (coarse scales run as two overlapping half-batches)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _AUX.get(idx)
    if st is None:
        st = torch.cuda.Stream(device=idx)
        _AUX[idx] = st
    return st.cuda_stream


class _Leaf(nn.Module):
    """Holds one (weight, bias) pair under the reference's key names."""

    def __init__(self, wshape, fan_in):
        super().__init__()
        w = torch.empty(wshape)
        # torch default init of nn.Conv2d / nn.Linear (kaiming_uniform a=sqrt(5); bias U(+-1/sqrt(fan_in)))
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        b = torch.empty(wshape[0]).uniform_(-bound, bound)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(b)


def _container(children: Dict[str, nn.Module]) -> nn.Module:
    m = nn.Module()
    for k, v in children.items():
        m.add_module(k, v)
    return m


class SinDDMNet(nn.Module):
    """4-block fully-convolutional eps-predictor conditioned on (t, s) -- reference
    SinDDM/models.py:85-151 -- executed by hand-written gfx950 kernels.

    All 52 parameter tensors are views into ONE flat fp32 buffer laid out in nn.Module
    registration order (what the C ABI expects); gradients likewise, so the fused Adam/EMA
    kernel walks a single array."""

    # per-call kernel option of the C ABI (include/sinddm_hip.h, SINDDM_DIM_FP32_CONVS): True keeps every 3x3 conv of this net's
    # launches on the fp32 matrix pipe -- A/B measurements and parity tests (set it on an instance, or on the class for every
    # net a test builds); no reference counterpart
    fp32_convs = False

    def __init__(self, dim, out_dim=None, channels=3, with_time_emb=True, multiscale=False, device=None):
        super().__init__()
        if not with_time_emb or not multiscale:
            # (not a gap against the reference: its forward tests `exists(self.multiscale)` -- true for False as well -- and then
            # dereferences SinEmbTime / SinEmbScale, which only the multiscale=True, with_time_emb=True constructor creates
            # (models.py:99-118,136-141): multiscale=False raises AttributeError at the first forward there, with_time_emb=False a TypeError in the constructor -- checked against the reference)
            raise NotImplementedError("the MI355X build implements the configuration main.py uses: "
                                      "with_time_emb=True, multiscale=True (reference main.py:77-81); the reference's own forward "
                                      "cannot run any other combination (models.py:136-141)")
        if channels != 3 or default(out_dim, channels) != 3:
            raise NotImplementedError("channels=3 / out_dim=3 only (reference main.py:77-81, models.py:129)")
        self.device = device
        self.channels = channels
        self.multiscale = multiscale
        self.dim = int(dim)
        time_dim = 32
        half = int(dim / 2)
        self.SinEmbTime = SinusoidalPosEmb(time_dim)
        self.SinEmbScale = SinusoidalPosEmb(time_dim)
        self.time_mlp = _container({"0": _Leaf((time_dim * 4, time_dim * 2), time_dim * 2),
                                    "2": _Leaf((time_dim, time_dim * 4), time_dim * 4)})
        for name, (cin, cout) in zip(("l1", "l2", "l3", "l4"),
                                     ((channels, half), (half, dim), (dim, dim), (dim, half))):
            kids = {
                "mlp": _container({"1": _Leaf((time_dim, time_dim), time_dim)}),
                "time_reshape": _Leaf((cin, time_dim, 1, 1), time_dim),
                "ds_conv": _Leaf((cin, 1, 5, 5), 25),
                "net": _container({"0": _Leaf((cout, cin, 3, 3), cin * 9), "2": _Leaf((cout, cout, 3, 3), cout * 9)}),
            }
            if cin != cout:
                kids["res_conv"] = _Leaf((cout, cin, 1, 1), cin)
            self.add_module(name, _container(kids))
        self.final_conv = _container({"0": _Leaf((channels, half, 1, 1), half)})

        self._flat: Optional[torch.Tensor] = None
        self._flat_grad: Optional[torch.Tensor] = None
        self._packed: Optional[torch.Tensor] = None
        self._packed_bwd: Optional[torch.Tensor] = None
        self._packed_version = -1
        self._packed_bwd_version = -1
        self._dirty = 0
        expected = net_param_shapes(self.dim, channels)
        got = {k: tuple(v.shape) for k, v in self.named_parameters()}
        assert list(got.items()) == list(expected.items()), "parameter layout drifted from the reference key order"
        self._flatten()

    # ---- flat parameter storage ---------------------------------------------------------
    def _flatten(self):
        params = list(self.parameters())
        dev = params[0].device
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1).to(torch.float32))
                p.data = flat[off:off + n].view(p.shape)
                p.grad = None
                off += n
        self._flat, self._flat_grad = flat, grad
        self._packed = None
        self._packed_bwd = None
        self._packed_version = self._packed_bwd_version = -1

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._flatten()
        return out

    def bind_grads(self):
        """Point every p.grad at its slice of the flat gradient buffer (idempotent)."""
        off = 0
        for p in self.parameters():
            n = p.numel()
            g = self._flat_grad[off:off + n].view(p.shape)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g
            off += n

    @property
    def flat_params(self) -> torch.Tensor:
        return self._flat

    @property
    def flat_grads(self) -> torch.Tensor:
        return self._flat_grad

    def _autograd_anchor(self) -> torch.Tensor:
        a = getattr(self, "_anchor", None)
        if a is None or a.device != self._flat.device:
            a = torch.zeros((), device=self._flat.device, requires_grad=True)
            self._anchor = a
        return a

    def mark_dirty(self):
        """Call after the parameters were changed through raw pointers (fused optimizer)."""
        self._dirty += 1

    def _param_version(self) -> int:
        # in-place updates through the nn.Parameter views (load_state_dict, torch optimizers) bump the
        # parameters' own version counters; raw-pointer updates call mark_dirty()
        return sum(p._version for p in self.parameters()) + (self._dirty << 32)

    def __deepcopy__(self, memo):
        # the EMA copy (trainer.py:100 in the reference) must own its own flat buffer
        new = type(self)(dim=self.dim, channels=self.channels, multiscale=True, device=self.device)
        new.to(self._flat.device)
        with torch.no_grad():
            new._flat.copy_(self._flat)
        for p_new, p_old in zip(new.parameters(), self.parameters()):
            p_new.requires_grad_(p_old.requires_grad)
        new.train(self.training)
        memo[id(self)] = new
        return new

    def _check_lib_layout(self):
        lib = _lib.load()
        n = lib.sinddm_param_count(self.dim)
        if n != self._flat.numel():
            raise _lib.SinddmError(f"parameter count mismatch: python {self._flat.numel()} vs library {n}")

    def packed_weights(self) -> torch.Tensor:
        """MFMA-ready weight image; rebuilt on device whenever the parameters changed."""
        lib = _lib.load()
        v = self._param_version()
        if self._packed is None or self._packed_version != v:
            if self._packed is None:
                self._check_lib_layout()
                self._packed = torch.empty(lib.sinddm_packed_count(self.dim), dtype=torch.float32,
                                           device=self._flat.device)
            _lib.check(lib.sinddm_pack_weights(_lib.ptr(self._flat), _lib.ptr(self._packed), self.dim,
                                               _lib.stream_ptr(self._flat.device)), "sinddm_pack_weights")
            self._packed_version = v
        return self._packed

    def packed_weights_bwd(self) -> torch.Tensor:
        lib = _lib.load()
        v = self._param_version()
        if self._packed_bwd is None or self._packed_bwd_version != v:
            if self._packed_bwd is None:
                self._packed_bwd = torch.empty(lib.sinddm_packed_bwd_count(self.dim), dtype=torch.float32,
                                               device=self._flat.device)
            _lib.check(lib.sinddm_pack_weights_bwd(_lib.ptr(self._flat), _lib.ptr(self._packed_bwd), self.dim,
                                                   _lib.stream_ptr(self._flat.device)), "sinddm_pack_weights_bwd")
            self._packed_bwd_version = v
        return self._packed_bwd

    # ---- forward ------------------------------------------------------------------------
    def infer(self, x: torch.Tensor, t_dev: Optional[torch.Tensor], t_host: int, scale: float,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Inference forward (no saved activations).  `t_dev` (B,) int64 or None -> all samples
        use the host integer `t_host` (the sampler's case, models.py:481,541)."""
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.SinddmError("SinDDMNet needs a ROCm device tensor: there is no CPU fallback")
        x = x.contiguous()
        if x.dtype != torch.float32:
            raise _lib.SinddmError("fp32 only")
        B, Cc, H, W = x.shape
        assert Cc == self.channels
        if out is None:
            out = torch.empty_like(x)
        packed = self.packed_weights()
        nbytes = lib.sinddm_workspace_bytes(self.dim, B, H, W)
        ws = _workspace(x.device, nbytes)
        if t_dev is not None:
            t_dev = t_dev.to(device=x.device, dtype=torch.int64).contiguous()
        _lib.check(lib.sinddm_net_forward(_lib.ptr(self._flat), _lib.ptr(packed), _lib.ptr(x),
                                          _lib.ptr(t_dev) if t_dev is not None else None, int(t_host),
                                          float(scale), _lib.ptr(out), self.dim_arg, B, H, W, ws.data_ptr(),
                                          ws.numel(), _lib.stream_ptr(x.device)), "sinddm_net_forward")
        return out

    @property
    def dim_arg(self) -> int:
        """The `dim` argument of the launching C-ABI calls: the width + this net's option bits."""
        return self.dim | (_lib.DIM_FP32_CONVS if self.fp32_convs else 0)

    def forward(self, x, time, scale=None):
        """eps = net(x, time, scale) -- same call signature as the reference (models.py:134)."""
        s = float(scale) if scale is not None else 0.0
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if needs_grad:
            from .autograd import net_forward_train
            return net_forward_train(self, x, time, s)
        return self.infer(x, time, 0, s)


# --------------------------------------------------------------------------------------------
# diffusion process
# --------------------------------------------------------------------------------------------
class MultiScaleGaussianDiffusion(nn.Module):
    """Multi-scale DDPM with SinDDM's re-blurring (reference SinDDM/models.py:155-631)."""

    def __init__(self, denoise_fn, *, save_interm=False, results_folder='/Results', n_scales, scale_factor,
                 image_sizes, scale_mul=(1, 1), channels=3, timesteps=100, train_full_t=False, scale_losses=None,
                 loss_factor=1, loss_type='l1', betas=None, device=None, reblurring=True, sample_limited_t=False,
                 omega=0):
        super().__init__()
        self.device = device
        self.save_interm = save_interm
        self.results_folder = Path(results_folder)
        self.channels = channels
        self.n_scales = n_scales
        self.scale_factor = scale_factor
        self.scale_mul = scale_mul
        self.sample_limited_t = sample_limited_t
        self.reblurring = reblurring
        self.img_prev_upsample = None

        # guided-sampling state of the reference (models.py:193-220).  CLIP itself (clip/, text2live_util/) is outside
        # the hot-path build; the guidance BRANCH of p_mean_variance (models.py:367-431) is implemented against the
        # interface the reference uses: `clip_model` is any object with zero_grad() and a differentiable
        # calculate_clip_loss(image in [0,1], text_embedds) -> scalar (stock PyTorch-ROCm autograd).
        self.clip_guided_sampling = False
        self.guidance_sub_iters = None
        self.stop_guidance = None
        self.quantile = 0.8
        self.clip_model = None
        self.clip_strength = None
        self.clip_text = ''
        self.text_embedds = None
        self.text_embedds_hr = None
        self.text_embedds_lr = None
        self.clip_text_features = None
        self.clip_score = []
        self.clip_mask = None
        self.llambda = 0
        self.x_recon_prev = None
        self.clip_roi_bb = []
        self.omega = omega
        self.roi_guided_sampling = False
        self.roi_bbs = []
        self.roi_bbs_stat = []
        self.roi_target_patch = []
        self._roi_cache = {}

        # (W,H) -> (H,W)   models.py:222-223
        self.image_sizes = tuple((image_sizes[i][1], image_sizes[i][0]) for i in range(n_scales))
        self.denoise_fn = denoise_fn

        if exists(betas):
            betas = betas.detach().cpu().numpy() if isinstance(betas, torch.Tensor) else betas
        else:
            betas = cosine_beta_schedule(timesteps)
        alphas = 1. - betas
        abar = np.cumprod(alphas, axis=0)
        abar_prev = np.append(1., abar[:-1])
        timesteps, = betas.shape
        self.num_timesteps = int(timesteps)
        self.loss_type = loss_type
        f32 = partial(torch.tensor, dtype=torch.float32)
        reg = self.register_buffer
        reg('betas', f32(betas))
        reg('alphas_cumprod', f32(abar))
        reg('alphas_cumprod_prev', f32(abar_prev))
        reg('sqrt_alphas_cumprod', f32(np.sqrt(abar)))
        reg('sqrt_one_minus_alphas_cumprod', f32(np.sqrt(1. - abar)))
        reg('log_one_minus_alphas_cumprod', f32(np.log(1. - abar)))
        reg('sqrt_recip_alphas_cumprod', f32(np.sqrt(1. / abar)))
        reg('sqrt_recipm1_alphas_cumprod', f32(np.sqrt(1. / abar - 1)))
        post_var = betas * (1. - abar_prev) / (1. - abar)
        reg('posterior_variance', f32(post_var))
        reg('posterior_log_variance_clipped', f32(np.log(np.maximum(post_var, 1e-20))))
        reg('posterior_mean_coef1', f32(betas * np.sqrt(abar_prev) / (1. - abar)))
        reg('posterior_mean_coef2', f32((1. - abar_prev) * np.sqrt(alphas) / (1. - abar)))

        # per-scale starting timesteps (bit-exact integer bookkeeping, models.py:269-280)
        sigma_t = np.sqrt(1. - abar) / np.sqrt(abar)
        self.num_timesteps_trained = [self.num_timesteps]
        self.num_timesteps_ideal = [self.num_timesteps]
        if scale_losses is not None:
            for i in range(n_scales - 1):
                self.num_timesteps_ideal.append(int(np.argmax(sigma_t > loss_factor * scale_losses[i])))
                self.num_timesteps_trained.append(int(timesteps) if train_full_t else self.num_timesteps_ideal[i + 1])
        # gamma blur schedule (models.py:283-287): float64 ratio, clamped, stored fp32
        gammas = torch.zeros((n_scales - 1, self.num_timesteps), dtype=torch.float32)
        for i in range(n_scales - 1):
            gammas[i, :] = (torch.tensor(sigma_t) / (loss_factor * scale_losses[i])).clamp(min=0, max=1)
        reg('gammas', gammas)
        if device is not None:
            # the reference builds `gammas` directly on `device`; mirror that placement
            self.gammas = self.gammas.to(device)

        self._host_tabs: Optional[dict] = None
        self._host_ver = None
        # optional noise hook for parity tests: fn(kind, shape, s, t, device) -> tensor
        self.noise_fn = None
        # replay aid: when set to a list, every host-side draw ('init' / 'renoise', the tensor itself) and every fused
        # run of reverse steps (('chain', s, seed, [t...]): the in-kernel draws are sinddm_normal_fill(seed, i)) is logged
        self.draw_log = None
        self.two_streams = True       # coarse scales as two half-batches on two streams (sinddm_sample_chain2)

    # ---- host copies of the per-t tables (scalar kernel arguments; no device sync per step) ----
    _TABS = ('alphas_cumprod', 'sqrt_alphas_cumprod', 'sqrt_one_minus_alphas_cumprod',
             'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod', 'posterior_log_variance_clipped',
             'posterior_mean_coef1', 'posterior_mean_coef2', 'gammas')

    def _host(self) -> dict:
        ver = tuple(getattr(self, n)._version for n in self._TABS) + tuple(getattr(self, n).data_ptr() for n in self._TABS)
        if self._host_tabs is None or ver != self._host_ver:
            h = {n: getattr(self, n).detach().cpu().numpy().astype(np.float32) for n in self._TABS}
            h['gammas_clamped'] = np.clip(h['gammas'], np.float32(0), np.float32(0.55))     # models.py:314,358
            h['sigma_plain'] = np.exp(np.float32(0.5) * h['posterior_log_variance_clipped']).astype(np.float32)
            self._host_tabs, self._host_ver = h, ver
        return self._host_tabs

    def _draw(self, kind: str, shape, s: int, t: int, device) -> torch.Tensor:
        if self.noise_fn is not None:
            return self.noise_fn(kind, tuple(shape), int(s), int(t), device).contiguous()
        z = torch.randn(tuple(shape), device=device)
        if self.draw_log is not None:
            self.draw_log.append((kind, int(s), int(t), z.clone()))
        return z

    def step_coefs(self, t: int, s: int, clip_denoised: bool = True) -> _lib.StepCoefs:
        """Host scalars of one reverse step (everything `extract` gathers in models.py:306-352)."""
        h = self._host()
        t = int(t)
        k = _lib.StepCoefs()
        k.clip = 1 if clip_denoised else 0
        k.sqrt_recip_ac_t = float(h['sqrt_recip_alphas_cumprod'][t])
        k.sqrt_recipm1_ac_t = float(h['sqrt_recipm1_alphas_cumprod'][t])
        one = np.float32(1)
        if (not self.reblurring) or int(s) == 0:
            k.mode = 0
            k.coef1_t = float(h['posterior_mean_coef1'][t])
            k.coef2_t = float(h['posterior_mean_coef2'][t])
            k.sigma = float(h['sigma_plain'][t]) if t != 0 else 0.0
            return k
        g = h['gammas_clamped'][int(s) - 1]
        k.gamma_t = float(g[t])
        if t > 0:
            k.mode = 1
            k.gamma_tm1 = float(g[t - 1])
            ac_tm1 = h['alphas_cumprod'][t - 1]
            var = np.float32(np.float32(self.omega) * (one - ac_tm1))                       # models.py:335-337
            logvar = np.log(np.maximum(var, np.float32(1e-20)))
            k.sqrt_ac_tm1 = float(h['sqrt_alphas_cumprod'][t - 1])
            k.sqrt_ac_t = float(h['sqrt_alphas_cumprod'][t])
            k.sqrt_1m_ac_t = float(h['sqrt_one_minus_alphas_cumprod'][t])
            k.sqrt_1m_ac_tm1_mvar = float(np.sqrt(np.float32(one - ac_tm1 - var)))
            k.sigma = float(np.exp(np.float32(0.5) * np.float32(logvar)))
        else:
            k.mode = 2
            k.sigma = 0.0
        return k

    # ---- reference API: fine-grained pieces (off the hot path; kept for callers / app modes) ----
    def q_mean_variance(self, x_start, t):                                  # models.py:300-304
        mean = extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
        variance = extract(1. - self.alphas_cumprod, t, x_start.shape)
        log_variance = extract(self.log_one_minus_alphas_cumprod, t, x_start.shape)
        return mean, variance, log_variance

    def predict_start_from_noise(self, x_t, t, s, noise):                   # models.py:306-318
        x0 = extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - extract(
            self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise
        if not self.reblurring or s == 0:
            return x0, x0
        g = extract(self.gammas[s - 1].reshape(-1).clamp(0, 0.55), t, x0.shape)
        return (x0 - g * self.img_prev_upsample) / (1 - g), x0

    def q_posterior(self, x_start, x_t_mix, x_t, t, s):                     # models.py:321-352
        if not self.reblurring or s == 0:
            mean = extract(self.posterior_mean_coef1, t, x_t.shape) * x_start + extract(
                self.posterior_mean_coef2, t, x_t.shape) * x_t
            var = extract(self.posterior_variance, t, x_t.shape)
            logvar = extract(self.posterior_log_variance_clipped, t, x_t.shape)
        elif t[0] > 0:
            var = self.omega * (1 - extract(self.alphas_cumprod, t - 1, x_t.shape)) + torch.zeros_like(x_t)
            logvar = torch.log(var.clamp(1e-20, None))
            mean = extract(self.sqrt_alphas_cumprod, t - 1, x_t.shape) * x_start + torch.sqrt(
                1 - extract(self.alphas_cumprod, t - 1, x_t.shape) - var) * (
                x_t - extract(self.sqrt_alphas_cumprod, t, x_t.shape) * x_t_mix) / extract(
                self.sqrt_one_minus_alphas_cumprod, t, x_t.shape)
        else:
            mean = x_start
            var = extract(self.posterior_variance, t, x_t.shape)
            logvar = extract(self.posterior_log_variance_clipped, t, x_t.shape)
        return mean, var, logvar

    def roi_patch_modification(self, x_recon, scale=0, eta=0.8):            # models.py:291-298 (in place, like it)
        w, c = self.roi_edit_maps(scale, x_recon.shape[-2], x_recon.shape[-1], x_recon.device, eta)
        x_recon.mul_(w[None, None]).add_(c[None])
        return x_recon

    def roi_edit_maps(self, scale: int, H: int, W: int, device, eta: float = 0.8):
        """The reference's sequential ROI blends `x[box] = eta * patch + (1 - eta) * x[box]` over all boxes
        (models.py:291-298) composed into ONE per-pixel affine map x -> w * x + c (w: (H,W), c: (3,H,W)); this is
        what the fused reverse-step kernel applies (`sinddm_reverse_step_edit`).  Cached per (scale, size, boxes)."""
        key = (int(scale), int(H), int(W), float(eta), tuple(tuple(int(v) for v in bb) for bb in self.roi_bbs),
               id(self.roi_target_patch[scale]))
        if self._roi_cache.get("key") == key:
            return self._roi_cache["w"], self._roi_cache["c"]
        w = torch.ones((H, W), device=device, dtype=torch.float32)
        c = torch.zeros((self.channels, H, W), device=device, dtype=torch.float32)
        for bb in self.roi_bbs:                                             # bounding box is [y, x, h, w]
            bb = [int(bb_i / np.power(self.scale_factor, self.n_scales - scale - 1)) for bb_i in bb]
            y, x, h, ww = bb
            patch = F.interpolate(self.roi_target_patch[scale].to(device=device, dtype=torch.float32), size=(h, ww))[0]
            c[:, y:y + h, x:x + ww] = eta * patch + (1 - eta) * c[:, y:y + h, x:x + ww]
            w[y:y + h, x:x + ww] *= (1 - eta)
        self._roi_cache = {"key": key, "w": w.contiguous(), "c": c.contiguous()}
        return self._roi_cache["w"], self._roi_cache["c"]

    def _clip_guidance(self, x_recon, t0: int, s: int, clip_denoised: bool):
        """The CLIP-guidance block of p_mean_variance (models.py:367-421): the score's gradient is taken with respect to
        x_recon itself (a leaf -- nothing flows back through the network), soft-thresholded into a mask at the first
        call, normalised to the image's norm inside the mask and added; changes of the previous step are blended in
        with `llambda`.  Runs on stock PyTorch autograd: `clip_model` is external (see __init__)."""
        from .functions import thresholded_grad
        if clip_denoised:
            x_recon = x_recon.clamp(-1., 1.)
        if self.clip_mask is not None:
            x_recon = x_recon * (1 - self.clip_mask) + (
                (1 - self.llambda) * self.x_recon_prev + self.llambda * x_recon) * self.clip_mask
        x_recon = x_recon.detach().requires_grad_(True)
        emb = self.text_embedds_hr if s > 0 else self.text_embedds_lr
        with torch.enable_grad():
            for i in range(self.guidance_sub_iters[s]):
                self.clip_model.zero_grad()
                score = -self.clip_model.calculate_clip_loss((x_recon + 1) * 0.5, emb)
                clip_grad = torch.autograd.grad(score, x_recon, create_graph=False)[0]
                if self.clip_mask is None:
                    clip_grad, clip_mask = thresholded_grad(grad=clip_grad, quantile=self.quantile)
                    self.clip_mask = clip_mask.float()
                if self.save_interm:                                        # models.py:394-404
                    self._dump_interm(self.clip_mask.to(torch.float64), s, f'clip_mask_s-{s}.png', renorm=False)
                    self._dump_interm(x_recon.detach().clamp(-1., 1.), s, f'clip_out_s-{s}_t-{t0}_subiter_{i}.png')
                with torch.no_grad():
                    division_norm = torch.linalg.vector_norm(x_recon * self.clip_mask, dim=(1, 2, 3), keepdim=True) / \
                        torch.linalg.vector_norm(clip_grad * self.clip_mask, dim=(1, 2, 3), keepdim=True)
                    x_recon += self.clip_strength * division_norm * clip_grad * self.clip_mask
                    x_recon.clamp_(-1., 1.)
                self.clip_score.append(score.detach().cpu())
        self.x_recon_prev = x_recon.detach()
        return self.x_recon_prev.clone()

    def _clip_active(self, t0: int, s: int) -> bool:                        # the condition of models.py:368
        return bool(self.clip_guided_sampling and (self.stop_guidance <= t0 or s < self.n_scales - 1)
                    and self.guidance_sub_iters[s] > 0)

    def p_mean_variance(self, x, t, s, clip_denoised: bool):               # models.py:354-447
        with torch.no_grad():
            eps = self._eps(x, t, int(t[0]), s)
            x_recon, x_t_mix = self.predict_start_from_noise(x, t=t, s=s, noise=eps)
            self._dump_x_recon(x_recon, int(t[0]), int(s))                  # models.py:360-366
        if self._clip_active(int(t[0]), int(s)):                            # models.py:367-421
            x_recon = self._clip_guidance(x_recon, int(t[0]), int(s), clip_denoised)
        elif self.roi_guided_sampling and (s < self.n_scales - 1):         # models.py:430-431
            x_recon = self.roi_patch_modification(x_recon, scale=s)
        with torch.no_grad():
            if int(s) > 0 and t[0] > 0 and self.reblurring:
                g = extract(self.gammas[s - 1].reshape(-1).clamp(0, 0.55), t - 1, x_recon.shape)
                x_tm1_mix = g * self.img_prev_upsample + (1 - g) * x_recon
            else:
                x_tm1_mix = x_recon
            if clip_denoised:
                x_tm1_mix = x_tm1_mix.clamp(-1., 1.)
                x_t_mix = x_tm1_mix if ((not self.reblurring) or s == 0) else x_t_mix.clamp(-1., 1.)
            return self.q_posterior(x_start=x_tm1_mix, x_t_mix=x_t_mix, x_t=x, t=t, s=s)

    def _p_sample_guided(self, x, t_host: int, s: int, clip_denoised: bool, repeat_noise: bool):
        """p_sample (models.py:449-459) through p_mean_variance in eager torch ops: the path of CLIP-guided steps (the
        fused reverse-step kernel has no place for an external autograd call between x_recon and the posterior)."""
        t = torch.full((x.shape[0],), int(t_host), device=x.device, dtype=torch.long)
        mean, _, logvar = self.p_mean_variance(x=x, t=t, s=s, clip_denoised=clip_denoised)
        if repeat_noise:
            z = noise_like(x.shape, x.device, True)
        else:
            z = self._draw("step", x.shape, s, t_host, x.device)
        nonzero = 0.0 if t_host == 0 else 1.0
        return mean + nonzero * (0.5 * logvar).exp() * z

    # ---- hot path -----------------------------------------------------------------------------
    def _coef_table(self, s: int, clip_denoised: bool = True):
        """`step_coefs` of every t of scale s as one ctypes array (built once per schedule / reblurring / omega)."""
        key = (int(s), bool(clip_denoised), bool(self.reblurring), float(self.omega))
        self._host()                                             # refreshes self._host_ver
        cache = getattr(self, "_coef_cache", None)
        if cache is None or cache.get("ver") != self._host_ver:
            cache = {"ver": self._host_ver}
            self._coef_cache = cache
        tab = cache.get(key)
        if tab is None:
            tab = (_lib.StepCoefs * self.num_timesteps)(*[self.step_coefs(t, s, clip_denoised)
                                                          for t in range(self.num_timesteps)])
            cache[key] = tab
        return tab

    def _run_steps(self, img: torch.Tensor, s: int, t_seq) -> torch.Tensor:
        """The reverse steps `t_seq` of scale s (the loop bodies of models.py:477-485,536-546).  Production path: ONE
        library call for the whole run -- the per-step scalars come from a prebuilt table, the states ping-pong between
        two buffers, the N(0,1) draws of models.py:455 are generated inside the step kernel (sinddm_sample_chain), and
        Python is not entered between steps.  Injected noise (`noise_fn`, parity tests), ROI guidance, intermediate
        dumps and foreign denoisers take the step-by-step path."""
        t_seq = [int(t) for t in t_seq]
        s = int(s)
        fast = (self.noise_fn is None and isinstance(self.denoise_fn, SinDDMNet) and not self.save_interm
                and not self.clip_guided_sampling and not (self.roi_guided_sampling and s < self.n_scales - 1)
                and len(t_seq) > 0 and img.is_cuda and img.dtype == torch.float32 and img.dim() == 4
                and img.shape[1] == self.channels == 3)
        if not fast:
            for i in t_seq:
                img = self._p_sample_host_t(img, i, s)
                self._dump_interm(img, s, f'output_t-{i:03}_s-{s}.png')
            return img
        lib = _lib.load()
        net = self.denoise_fn
        x = img.contiguous().clone()
        x_alt = torch.empty_like(x)
        eps = torch.empty_like(x)
        B, Cc, H, W = x.shape
        tab = self._coef_table(s)
        n = len(t_seq)
        coefs = (_lib.StepCoefs * n)(*[tab[t] for t in t_seq])
        tl = (C.c_int * n)(*t_seq)
        xt = None
        if coefs[0].mode != 0 or coefs[n - 1].mode != 0:
            xt = self.img_prev_upsample
            if xt is None:
                raise _lib.SinddmError("img_prev_upsample is not set (call sample_via_scale / p_sample_via_scale_loop)")
            if xt.shape != x.shape or xt.dtype != x.dtype or xt.device != x.device:
                # (the library takes raw pointers: a mismatching x-tilde would be read out of bounds)
                raise _lib.SinddmError(f"img_prev_upsample {tuple(xt.shape)} {xt.dtype} {xt.device} does not match the "
                                       f"running sample {tuple(x.shape)} {x.dtype} {x.device}")
            xt = xt.contiguous()
        packed = net.packed_weights()
        ws = _workspace(x.device, lib.sinddm_workspace_bytes(net.dim, B, H, W))
        # the step noise is keyed on a 62-bit seed drawn from torch's CPU generator: torch.manual_seed() reproduces a
        # sample, seeding only the CUDA generator (torch.cuda.manual_seed) does not
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))
        if self.draw_log is not None:
            self.draw_log.append(("chain", s, seed, list(t_seq)))
        in_alt = C.c_int(0)
        # (the second stream lets the library run coarse scales as two overlapping half-batches; same numbers either way)
        _lib.check(lib.sinddm_sample_chain2(_lib.ptr(net.flat_params), _lib.ptr(packed), _lib.ptr(x), _lib.ptr(x_alt),
                                            _lib.ptr(eps), _lib.ptr(xt), coefs, tl, n, float(s), seed, 0, net.dim_arg, B, H, W,
                                            ws.data_ptr(), ws.numel(), _lib.stream_ptr(x.device),
                                            _aux_stream(x.device) if self.two_streams else None, C.byref(in_alt)),
                   "sinddm_sample_chain2")
        return x_alt if in_alt.value else x

    def _eps(self, x, t_dev, t_host, s):
        if isinstance(self.denoise_fn, SinDDMNet):
            return self.denoise_fn.infer(x, None if t_dev is None else t_dev, int(t_host), float(s))
        if t_dev is None:
            t_dev = torch.full((x.shape[0],), int(t_host), device=x.device, dtype=torch.long)
        return self.denoise_fn(x, t_dev, scale=s)                           # plug point, models.py:356

    def _p_sample_host_t(self, x: torch.Tensor, t: int, s: int, clip_denoised: bool = True,
                         repeat_noise: bool = False) -> torch.Tensor:
        """One reverse step with the timestep known on the host: net forward + ONE fused kernel."""
        if self.clip_guided_sampling:
            if self.clip_model is None or self.guidance_sub_iters is None or self.stop_guidance is None:
                raise RuntimeError("clip_guided_sampling is set but clip_model / guidance_sub_iters / stop_guidance are not: "
                                   "CLIP is outside this build -- assign any object with zero_grad() and a differentiable "
                                   "calculate_clip_loss(image, text_embedds) (reference models.py:193-220, 367-421)")
            return self._p_sample_guided(x.contiguous(), int(t), int(s), clip_denoised, repeat_noise)
        lib = _lib.load()
        x = x.contiguous()
        eps = self._eps(x, None, t, s)
        if self.save_interm:                                                # models.py:360-366 (x_recon lives inside the fused kernel:
            tt = torch.full((x.shape[0],), int(t), device=x.device, dtype=torch.long)   # recomputed here, for the dump only)
            self._dump_x_recon(self.predict_start_from_noise(x, t=tt, s=s, noise=eps)[0], t, s)
        if repeat_noise:
            z = noise_like(x.shape, x.device, True).contiguous()
        else:
            z = self._draw("step", x.shape, s, t, x.device)
        k = self.step_coefs(t, s, clip_denoised)
        xt = None
        if k.mode != 0:
            xt = self.img_prev_upsample
            if xt is None:
                raise _lib.SinddmError("img_prev_upsample is not set (call sample_via_scale / p_sample_via_scale_loop)")
            xt = xt.contiguous()
        out = torch.empty_like(x)
        if self.roi_guided_sampling and s < self.n_scales - 1:             # models.py:430-431
            B_, C_, H_, W_ = x.shape
            ew, ec = self.roi_edit_maps(s, H_, W_, x.device)
            _lib.check(lib.sinddm_reverse_step_edit(_lib.ptr(x), _lib.ptr(eps), _lib.ptr(xt), _lib.ptr(z),
                                                    _lib.ptr(out), C.byref(k), _lib.ptr(ew), _lib.ptr(ec), B_, C_,
                                                    H_ * W_, _lib.stream_ptr(x.device)), "sinddm_reverse_step_edit")
            return out
        _lib.check(lib.sinddm_reverse_step(_lib.ptr(x), _lib.ptr(eps), _lib.ptr(xt), _lib.ptr(z), _lib.ptr(out),
                                           C.byref(k), x.numel(), _lib.stream_ptr(x.device)), "sinddm_reverse_step")
        return out

    @torch.no_grad()
    def p_sample(self, x, t, s, clip_denoised=True, repeat_noise=False):   # models.py:449-459
        t_host = int(t[0]) if isinstance(t, torch.Tensor) else int(t)      # the reference also reads t[0] (:331,:434)
        return self._p_sample_host_t(x, t_host, int(s), clip_denoised, repeat_noise)

    @torch.no_grad()
    def p_sample_loop(self, shape, s):                                     # models.py:462-487
        device = self.betas.device
        img = self._draw("init", shape, s, 0, device)
        self._dump_interm(img, s, f'input_noise_s-{s}.png')
        if self.sample_limited_t and s < (self.n_scales - 1):
            t_min = self.num_timesteps_ideal[s + 1]
        else:
            t_min = 0
        return self._run_steps(img, s, reversed(range(t_min, self.num_timesteps)))

    def _dump_interm(self, img, s, name, renorm=True):
        """save_interm=True (models.py:469-485,520-546): PNG grid of the running sample after every step (debug aid)."""
        if not self.save_interm:
            return
        from .trainer import save_image
        folder = Path(str(self.results_folder / f'interm_samples_scale_{s}'))
        folder.mkdir(parents=True, exist_ok=True)
        save_image((img + 1) * 0.5 if renorm else img, str(folder / name), nrow=4)

    def _dump_x_recon(self, x_recon, t_host: int, s: int):
        """save_interm=True: the denoised estimate of the step, `denoised_t-TTT_s-S.png` (models.py:360-366)."""
        self._dump_interm(x_recon.clamp(-1., 1.), s, f'denoised_t-{int(t_host):03}_s-{int(s)}.png')

    @torch.no_grad()
    def sample(self, batch_size=16, scale_0_size=None, s=0):               # models.py:489-499
        image_size = scale_0_size if scale_0_size is not None else self.image_sizes[0]
        return self.p_sample_loop((batch_size, self.channels, image_size[0], image_size[1]), s=s)

    @torch.no_grad()
    def p_sample_via_scale_loop(self, batch_size, img, s, custom_t=None):  # models.py:501-547
        if custom_t is None:
            total_t = self.num_timesteps_ideal[min(s, self.n_scales - 1)] - 1
        else:
            total_t = custom_t
        total_t = int(total_t)
        self.img_prev_upsample = img                                        # x-tilde of this scale
        noise = self._draw("renoise", img.shape, s, 0, img.device)
        img = self._q_sample_impl(img, None, total_t, noise)                # models.py:518
        self._dump_interm(img, s, f'noisy_input_s_{s}.png')
        if self.clip_mask is not None:                                      # models.py:528-535
            if s > 0:
                mul_size = [int(self.image_sizes[s][0] * self.scale_mul[0]), int(self.image_sizes[s][1] * self.scale_mul[1])]
                self.clip_mask = F.interpolate(self.clip_mask, size=mul_size, mode='bilinear')
                self.x_recon_prev = F.interpolate(self.x_recon_prev, size=mul_size, mode='bilinear')
            else:                                                           # a mask created at scale 0 is too noisy
                self.clip_mask = None
        if self.sample_limited_t and s < (self.n_scales - 1):
            t_min = self.num_timesteps_ideal[s + 1]
        else:
            t_min = 0
        return self._run_steps(img, s, reversed(range(t_min, total_t)))

    def target_size(self, s, scale_mul=(1, 1), custom_sample=False, custom_img_size_idx=0, custom_image_size=None):
        """Size selection of sample_via_scale (models.py:554-565), int() truncation included."""
        if custom_sample:
            if custom_img_size_idx >= self.n_scales:
                size = self.image_sizes[self.n_scales - 1]
                factor = self.scale_factor ** (custom_img_size_idx + 1 - self.n_scales)
                size = (int(size[0] * factor), int(size[1] * factor))
            else:
                size = self.image_sizes[custom_img_size_idx]
        else:
            size = self.image_sizes[s]
        image_size = (int(size[0] * scale_mul[0]), int(size[1] * scale_mul[1]))
        if custom_image_size is not None:
            image_size = custom_image_size
        return image_size

    def upsample(self, img: torch.Tensor, size) -> torch.Tensor:
        """F.interpolate(img, size, mode='bilinear') on the HIP kernel (models.py:567)."""
        lib = _lib.load()
        img = img.contiguous()
        B, Cc, h, w = img.shape
        H, W = int(size[0]), int(size[1])
        out = torch.empty((B, Cc, H, W), dtype=img.dtype, device=img.device)
        _lib.check(lib.sinddm_upsample_bilinear(_lib.ptr(img), _lib.ptr(out), B * Cc, h, w, H, W,
                                                _lib.stream_ptr(img.device)), "sinddm_upsample_bilinear")
        return out

    @torch.no_grad()
    def sample_via_scale(self, batch_size, img, s, scale_mul=(1, 1), custom_sample=False, custom_img_size_idx=0,
                         custom_t=None, custom_image_size=None):           # models.py:549-568
        image_size = self.target_size(s, scale_mul, custom_sample, custom_img_size_idx, custom_image_size)
        img = self.upsample(img, image_size)
        return self.p_sample_via_scale_loop(batch_size, img, s, custom_t=custom_t)

    def _q_sample_impl(self, x0, t_dev, t_host, noise, x_orig=None, gamma_row=None):
        lib = _lib.load()
        x0 = x0.contiguous()
        noise = noise.contiguous()
        out = torch.empty_like(x0)
        B = x0.shape[0]
        n = x0.numel() // B
        if t_dev is not None:
            t_dev = t_dev.to(device=x0.device, dtype=torch.int64).contiguous()
        _lib.check(lib.sinddm_q_sample(_lib.ptr(x0), _lib.ptr(x_orig.contiguous()) if x_orig is not None else None,
                                       _lib.ptr(noise), _lib.ptr(out), _lib.ptr(self.sqrt_alphas_cumprod),
                                       _lib.ptr(self.sqrt_one_minus_alphas_cumprod),
                                       _lib.ptr(gamma_row) if gamma_row is not None else None,
                                       _lib.ptr(t_dev) if t_dev is not None else None, int(t_host), B, n,
                                       _lib.stream_ptr(x0.device)), "sinddm_q_sample")
        return out

    def q_sample(self, x_start, t, noise=None):                            # models.py:570-576
        noise = default(noise, lambda: torch.randn_like(x_start))
        return self._q_sample_impl(x_start, t, 0, noise)

    def p_losses(self, x_start, t, s, noise=None, x_orig=None):            # models.py:578-611
        noise = default(noise, lambda: torch.randn_like(x_start))
        if self.loss_type not in ('l1', 'l2', 'l1_pred_img'):
            raise NotImplementedError()
        if int(s) > 0:
            gamma_row = self.gammas[int(s) - 1].reshape(-1).contiguous()    # NOT clamped to 0.55 in training
            x_noisy = self._q_sample_impl(x_start, t, 0, noise, x_orig=x_orig, gamma_row=gamma_row)
        else:
            x_noisy = self._q_sample_impl(x_start, t, 0, noise)
        x_recon = self.denoise_fn(x_noisy, t, int(s))
        if self.loss_type == 'l1':                                          # what main.py:97 selects: fused loss + seed kernel
            from .autograd import l1_loss
            return l1_loss(noise, x_recon)
        # the two loss types main.py never selects (models.py:595-607): the network's forward / backward are the HIP path
        # (x_recon carries its autograd node), the loss itself is three elementwise torch ops
        if self.loss_type == 'l2':
            return F.mse_loss(noise, x_recon)
        if int(s) > 0:
            if int(t[0]) > 0:                                               # (the reference's host sync, models.py:599)
                g = extract(self.gammas[int(s) - 1].reshape(-1), t - 1, x_start.shape)
                x_mix_prev = g * x_start + (1 - g) * x_orig
            else:
                x_mix_prev = x_orig
        else:
            x_mix_prev = x_start
        return (x_mix_prev - x_recon).abs().mean()

    def forward(self, x, s, *args, **kwargs):                              # models.py:613-631
        s = int(s)
        if s > 0:
            x_orig, x_recon = x[0], x[1]
            b, c, h, w = x_orig.shape
            img_size = self.image_sizes[s]
            assert h == img_size[0] and w == img_size[1], f'height and width of image must be {img_size}'
            t = torch.randint(0, self.num_timesteps_trained[s], (b,), device=x_orig.device).long()
            return self.p_losses(x_recon, t, s, x_orig=x_orig, *args, **kwargs)
        b, c, h, w = x[0].shape
        img_size = self.image_sizes[s]
        assert h == img_size[0] and w == img_size[1], f'height and width of image must be {img_size}'
        t = torch.randint(0, self.num_timesteps_trained[s], (b,), device=x[0].device).long()
        return self.p_losses(x[0], t, s, *args, **kwargs)
