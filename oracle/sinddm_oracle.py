"""CPU oracle for the SinDDM multi-scale diffusion hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain fp32 restatement (torch CPU
functional ops + float64 numpy for the schedule) of the algorithm the reference
implements in /root/reference/SinDDM/{models,functions,trainer}.py.  It is the
checker for the HIP path in ``sinddm_amd``; nothing in the product path may
import it.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
itself in the build container (it cannot travel to the GPU box) and stores its
outputs as fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every function below against those fixtures.

Every function cites the reference file:line it restates.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# (C_in, C_out) of the four conv blocks at width `dim` (models.py:122-127)
def block_channels(dim: int, channels: int = 3) -> List[Tuple[int, int]]:
    half = int(dim / 2)
    return [(channels, half), (half, dim), (dim, dim), (dim, half)]


# ---------------------------------------------------------------------------
# schedule  (functions.py:117-127, models.py:227-287)
# ---------------------------------------------------------------------------
def cosine_beta_schedule(timesteps: int, s: float = 0.008) -> np.ndarray:
    """float64 cosine schedule, functions.py:117-127."""
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return np.clip(betas, a_min=0, a_max=0.999)


SCHEDULE_BUFFERS = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
    "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2",
)


def make_schedule(timesteps: int, n_scales: int, scale_losses: Optional[Sequence[float]],
                  loss_factor: float = 1, train_full_t: bool = False) -> dict:
    """All per-t buffers (f32), num_timesteps_ideal / _trained (ints) and the
    gamma blur schedule.  models.py:227-287."""
    betas = cosine_beta_schedule(timesteps)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    out = {
        "betas": f32(betas),
        "alphas_cumprod": f32(ac),
        "alphas_cumprod_prev": f32(ac_prev),
        "sqrt_alphas_cumprod": f32(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - ac)),
        "log_one_minus_alphas_cumprod": f32(np.log(1.0 - ac)),
        "sqrt_recip_alphas_cumprod": f32(np.sqrt(1.0 / ac)),
        "sqrt_recipm1_alphas_cumprod": f32(np.sqrt(1.0 / ac - 1)),
    }
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    out["posterior_variance"] = f32(post_var)
    out["posterior_log_variance_clipped"] = f32(np.log(np.maximum(post_var, 1e-20)))
    out["posterior_mean_coef1"] = f32(betas * np.sqrt(ac_prev) / (1.0 - ac))
    out["posterior_mean_coef2"] = f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac))

    sigma_t = np.sqrt(1.0 - ac) / np.sqrt(ac)                     # models.py:269
    ideal = [int(timesteps)]
    trained = [int(timesteps)]
    if scale_losses is not None:
        for i in range(n_scales - 1):                             # models.py:272-280
            ideal.append(int(np.argmax(sigma_t > loss_factor * scale_losses[i])))
            trained.append(int(timesteps) if train_full_t else ideal[i + 1])
    gammas = torch.zeros((max(n_scales - 1, 0), timesteps), dtype=torch.float32)
    for i in range(n_scales - 1):                                 # models.py:283-285
        gammas[i, :] = (torch.tensor(sigma_t) / (loss_factor * scale_losses[i])).clamp(min=0, max=1)
    out["gammas"] = gammas
    out["num_timesteps"] = int(timesteps)
    out["num_timesteps_ideal"] = ideal
    out["num_timesteps_trained"] = trained
    return out


# ---------------------------------------------------------------------------
# network  (models.py:34-151)
# ---------------------------------------------------------------------------
def sinusoidal_emb(x: Tensor, dim: int = 32) -> Tensor:
    """models.py:39-46: [sin(x f_i) | cos(x f_i)], f_i = exp(-i ln(1e4)/(half-1))."""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half) * -k)
    arg = x[:, None] * f[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def gelu(x: Tensor) -> Tensor:
    """exact-erf GELU (nn.GELU default), models.py:55,64,108."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def cond_vector(sd: Dict[str, Tensor], t: Tensor, scale) -> Tensor:
    """models.py:136-141: time_mlp([emb(t) | emb(s)]) -> (B, 32)."""
    scale_tensor = torch.ones(t.shape) * float(scale)
    e = torch.cat((sinusoidal_emb(t), sinusoidal_emb(scale_tensor)), dim=1)
    h = F.linear(e, sd["time_mlp.0.weight"], sd["time_mlp.0.bias"])
    h = gelu(h)
    return F.linear(h, sd["time_mlp.2.weight"], sd["time_mlp.2.bias"])


def block_condition(sd: Dict[str, Tensor], name: str, cond: Tensor) -> Tensor:
    """models.py:74-76: time_reshape(Linear(GELU(cond))) -> (B, C_in) per-sample bias."""
    c = F.linear(gelu(cond), sd[f"{name}.mlp.1.weight"], sd[f"{name}.mlp.1.bias"])
    w = sd[f"{name}.time_reshape.weight"]
    return F.linear(c, w.reshape(w.shape[0], w.shape[1]), sd[f"{name}.time_reshape.bias"])


def conv_block(sd: Dict[str, Tensor], name: str, x: Tensor, cond: Tensor,
               return_intermediates: bool = False):
    """models.py:69-80."""
    cin = x.shape[1]
    h = F.conv2d(x, sd[f"{name}.ds_conv.weight"], sd[f"{name}.ds_conv.bias"], padding=2, groups=cin)
    h = h + block_condition(sd, name, cond)[:, :, None, None]
    u = F.conv2d(h, sd[f"{name}.net.0.weight"], sd[f"{name}.net.0.bias"], padding=1)
    g = gelu(u)
    o = F.conv2d(g, sd[f"{name}.net.2.weight"], sd[f"{name}.net.2.bias"], padding=1)
    if f"{name}.res_conv.weight" in sd:
        r = F.conv2d(x, sd[f"{name}.res_conv.weight"], sd[f"{name}.res_conv.bias"])
    else:
        r = x
    out = o + r
    if return_intermediates:
        return out, dict(h=h, u=u, g=g)
    return out


def net_forward(sd: Dict[str, Tensor], x: Tensor, t: Tensor, scale) -> Tensor:
    """SinDDMNet.forward (multiscale=True), models.py:134-151."""
    cond = cond_vector(sd, t, scale)
    for name in ("l1", "l2", "l3", "l4"):
        x = conv_block(sd, name, x, cond)
    return F.conv2d(x, sd["final_conv.0.weight"], sd["final_conv.0.bias"])


# ---------------------------------------------------------------------------
# diffusion process  (models.py:300-631, functions.py:105-108)
# ---------------------------------------------------------------------------
def extract(a: Tensor, t: Tensor, ndim: int = 4) -> Tensor:
    """functions.py:105-108."""
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))


def q_sample(sched: dict, x_start: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """models.py:570-576."""
    return (extract(sched["sqrt_alphas_cumprod"], t) * x_start
            + extract(sched["sqrt_one_minus_alphas_cumprod"], t) * noise)


def p_losses_inputs(sched: dict, x_start: Tensor, t: Tensor, s: int, noise: Tensor,
                    x_orig: Optional[Tensor]) -> Tensor:
    """x_noisy of models.py:582-590 (gammas NOT clamped to 0.55 in training)."""
    if int(s) > 0:
        g = extract(sched["gammas"][s - 1].reshape(-1), t)
        x_mix = g * x_start + (1 - g) * x_orig
        return q_sample(sched, x_mix, t, noise)
    return q_sample(sched, x_start, t, noise)


def p_losses(sched: dict, sd: Dict[str, Tensor], x_start: Tensor, t: Tensor, s: int,
             noise: Tensor, x_orig: Optional[Tensor] = None, loss_type: str = "l1") -> Tensor:
    """models.py:578-611: 'l1' (what main.py:97 selects), 'l2', 'l1_pred_img'."""
    x_noisy = p_losses_inputs(sched, x_start, t, s, noise, x_orig)
    eps = net_forward(sd, x_noisy, t, s)
    if loss_type == "l1":
        return (noise - eps).abs().mean()
    if loss_type == "l2":
        return F.mse_loss(noise, eps)
    if loss_type == "l1_pred_img":                                   # models.py:597-607
        if int(s) > 0:
            if int(t[0]) > 0:
                g = extract(sched["gammas"][s - 1].reshape(-1), t - 1)
                x_mix_prev = g * x_start + (1 - g) * x_orig
            else:
                x_mix_prev = x_orig
        else:
            x_mix_prev = x_start
        return (x_mix_prev - eps).abs().mean()
    raise NotImplementedError(loss_type)


def roi_patch_modification(x_recon: Tensor, roi_bbs, target_patch: Tensor, scale_factor: float, n_scales: int,
                           scale: int, eta: float = 0.8) -> Tensor:
    """models.py:291-298: inside every box (given at finest-scale coordinates, rescaled with int() truncation) the
    predicted clean image is blended with the nearest-resized target patch, sequentially over the boxes."""
    x = x_recon.clone()
    for bb in roi_bbs:
        bb = [int(b / np.power(scale_factor, n_scales - scale - 1)) for b in bb]
        y, xx, h, w = bb
        patch = F.interpolate(target_patch, size=(h, w))
        x[:, :, y:y + h, xx:xx + w] = eta * patch + (1 - eta) * x[:, :, y:y + h, xx:xx + w]
    return x


def reverse_step(sched: dict, x: Tensor, eps: Tensor, t: int, s: int, noise: Tensor,
                 x_tilde: Optional[Tensor], reblurring: bool = True, omega: float = 0.0,
                 clip_denoised: bool = True, x_recon_edit=None) -> Tensor:
    """Everything in p_sample after the net call: predict_start_from_noise
    (models.py:306-318) + the normal-sampling branch of p_mean_variance (:433-447)
    + q_posterior (:321-352) + the noise add of p_sample (:455-459).
    All samples share the integer timestep `t` (models.py:481,541).
    `x_recon_edit`: optional callable applied to x_recon where the reference applies
    roi_patch_modification (models.py:430-431)."""
    B = x.shape[0]
    tt = torch.full((B,), int(t), dtype=torch.long)
    x0 = extract(sched["sqrt_recip_alphas_cumprod"], tt) * x - extract(sched["sqrt_recipm1_alphas_cumprod"], tt) * eps
    plain = (not reblurring) or int(s) == 0
    if plain:
        if x_recon_edit is not None:
            x0 = x_recon_edit(x0)          # x_recon and x_t_mix are one tensor here (models.py:311-312)
        x_tm1_mix = x0
        x_t_mix = x0
    else:
        cur_g = sched["gammas"][s - 1].reshape(-1).clamp(0, 0.55)
        g_t = extract(cur_g, tt)
        x_tm1_mix = (x0 - g_t * x_tilde) / (1 - g_t)
        if x_recon_edit is not None:
            x_tm1_mix = x_recon_edit(x_tm1_mix)
        x_t_mix = x0
    # models.py:434-438
    if int(s) > 0 and t > 0 and reblurring:
        g_tm1 = extract(cur_g, tt - 1)
        x_tm1_mix = g_tm1 * x_tilde + (1 - g_tm1) * x_tm1_mix
    if clip_denoised:
        x_tm1_mix = x_tm1_mix.clamp(-1.0, 1.0)
        # when plain, x_t_mix aliases x_tm1_mix in the reference (same tensor, clamped in place)
        x_t_mix = x_tm1_mix if plain else x_t_mix.clamp(-1.0, 1.0)
    # q_posterior
    if plain:
        mean = extract(sched["posterior_mean_coef1"], tt) * x_tm1_mix + extract(sched["posterior_mean_coef2"], tt) * x
        logvar = extract(sched["posterior_log_variance_clipped"], tt)
    elif t > 0:
        var_low = torch.zeros(x.shape)
        var_high = 1 - extract(sched["alphas_cumprod"], tt - 1)
        var = (1 - omega) * var_low + omega * var_high
        logvar = torch.log(var.clamp(1e-20, None))
        mean = (extract(sched["sqrt_alphas_cumprod"], tt - 1) * x_tm1_mix
                + torch.sqrt(1 - extract(sched["alphas_cumprod"], tt - 1) - var)
                * (x - extract(sched["sqrt_alphas_cumprod"], tt) * x_t_mix)
                / extract(sched["sqrt_one_minus_alphas_cumprod"], tt))
    else:
        mean = x_tm1_mix
        logvar = extract(sched["posterior_log_variance_clipped"], tt)
    nonzero = 0.0 if t == 0 else 1.0
    return mean + nonzero * (0.5 * logvar).exp() * noise


def p_sample(sched: dict, sd: Dict[str, Tensor], x: Tensor, t: int, s: int, noise: Tensor,
             x_tilde: Optional[Tensor], **kw) -> Tensor:
    """models.py:449-459 with the noise supplied by the caller."""
    B = x.shape[0]
    eps = net_forward(sd, x, torch.full((B,), int(t), dtype=torch.long), s)
    return reverse_step(sched, x, eps, t, s, noise, x_tilde, **kw)


def bilinear_upsample(img: Tensor, size: Tuple[int, int]) -> Tensor:
    """F.interpolate(img, size, mode='bilinear') semantics (align_corners=False, no
    antialias) as called at models.py:567, written out in fp32 coordinate arithmetic (source index = one fused
    multiply-add, like ATen's compiled kernels)."""
    B, C, h, w = img.shape
    H, W = int(size[0]), int(size[1])

    def axis(n_in, n_out):
        scale = torch.tensor(n_in / n_out, dtype=torch.float32) if n_out > 0 else torch.tensor(0.0)
        dst = torch.arange(n_out, dtype=torch.float32)
        # ATen evaluates scale*(dst+0.5)-0.5 with ONE rounding (the compiler contracts it to an fma, on CPU and GPU
        # builds alike): pinned by fixture g8:hash_8x776_to_11x1092, where a two-rounding evaluation is 3e-5 off
        src = (scale.double() * (dst.double() + 0.5) - 0.5).float().clamp(min=0.0)
        i0 = src.floor().to(torch.long).clamp(max=n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        lam = src - i0.to(torch.float32)
        return i0, i1, lam

    y0, y1, ly = axis(h, H)
    x0, x1, lx = axis(w, W)
    top = img[:, :, y0, :]
    bot = img[:, :, y1, :]
    ly = ly[None, None, :, None]
    lx = lx[None, None, None, :]
    tl, tr = top[:, :, :, x0], top[:, :, :, x1]
    bl, br = bot[:, :, :, x0], bot[:, :, :, x1]
    return (1 - ly) * ((1 - lx) * tl + lx * tr) + ly * ((1 - lx) * bl + lx * br)


def scale_size(image_sizes_hw, n_scales, scale_factor, s, scale_mul=(1, 1), custom_sample=False,
               custom_img_size_idx=0, custom_image_size=None):
    """Size selection of sample_via_scale, models.py:554-565 (int() truncation)."""
    if custom_sample:
        if custom_img_size_idx >= n_scales:
            size = image_sizes_hw[n_scales - 1]
            factor = scale_factor ** (custom_img_size_idx + 1 - n_scales)
            size = (int(size[0] * factor), int(size[1] * factor))
        else:
            size = image_sizes_hw[custom_img_size_idx]
    else:
        size = image_sizes_hw[s]
    image_size = (int(size[0] * scale_mul[0]), int(size[1] * scale_mul[1]))
    if custom_image_size is not None:
        image_size = custom_image_size
    return image_size


def sample_chain(sched: dict, sd: Dict[str, Tensor], sizes_hw: Sequence[Tuple[int, int]],
                 noises: dict, batch: int, custom_t_list: Optional[Sequence[int]] = None,
                 trace: Optional[list] = None, roi: Optional[dict] = None) -> List[Tensor]:
    """Full multi-scale sampling, trainer.py:226-285 -> models.py:463-568, with every
    random draw supplied through `noises`:
      noises[("init", 0)]           (B,3,H0,W0)  the randn of models.py:467
      noises[("renoise", s)]        (B,3,Hs,Ws)  the q_sample noise of models.py:518
      noises[("step", s, t)]        (B,3,Hs,Ws)  the randn of models.py:455
    Returns the per-scale outputs.  `trace` collects (s, total_t, [t...]) bookkeeping.
    `roi` = dict(bbs, target_patch[per scale], scale_factor) turns on ROI guided sampling."""
    n_scales = len(sizes_hw)
    ideal = sched["num_timesteps_ideal"]
    if custom_t_list is None:
        custom_t_list = ideal[1:]
    def edit(s):
        # ROI guidance (trainer.py:436-454 -> models.py:430-431): every scale but the finest
        if roi is None or not (s < n_scales - 1):
            return None
        return lambda xr: roi_patch_modification(xr, roi["bbs"], roi["target_patch"][s], roi["scale_factor"],
                                                 n_scales, s)

    outs = []
    T = sched["num_timesteps"]
    img = noises[("init", 0)].clone()
    ts = list(reversed(range(0, T)))
    if trace is not None:
        trace.append((0, T, ts))
    for t in ts:
        img = p_sample(sched, sd, img, t, 0, noises[("step", 0, t)], None, x_recon_edit=edit(0))
    outs.append(img)
    for s in range(1, n_scales):
        up = bilinear_upsample(outs[-1], sizes_hw[s])
        total_t = int(custom_t_list[s - 1])                        # models.py:504-507 (no -1)
        tt = torch.full((batch,), total_t, dtype=torch.long)
        img = q_sample(sched, up, tt, noises[("renoise", s)])
        ts = list(reversed(range(0, total_t)))
        if trace is not None:
            trace.append((s, total_t, ts))
        for t in ts:
            img = p_sample(sched, sd, img, t, s, noises[("step", s, t)], up, x_recon_edit=edit(s))
        outs.append(img)
    return outs


# ---------------------------------------------------------------------------
# optimiser pieces  (trainer.py:134-136,155-159,208-213; models.py:23-31)
# ---------------------------------------------------------------------------
def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam defaults (no wd, no amsgrad); `step` is 1-based.  In place."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def multistep_lr(lr0: float, milestones: Sequence[int], n: int, gamma: float = 0.5) -> float:
    """LR used by optimizer step n (1-based) under MultiStepLR stepped after opt.step()
    (trainer.py:136,213)."""
    k = sum(1 for m in milestones if m <= n - 1)
    return lr0 * (gamma ** k)


def ema_update(ema: Tensor, p: Tensor, beta: float = 0.995) -> Tensor:
    """models.py:28-31."""
    return ema * beta + (1 - beta) * p
