#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  The reference is imported as-is from /root/reference with stub
modules for the packages this image lacks (torchvision, skimage) -- recipe of
SURVEY.md 8(c).  Only inputs/outputs are written; no reference source is copied.

    python tests/golden/make_golden.py            # regenerates every fixture

Fixtures (G-numbers follow SURVEY.md 8(c)):
  g1_schedule.npz    schedule buffers, num_timesteps_ideal/_trained, gammas  (C1..C5)
  g2_posemb.npz      SinusoidalPosEmb(32) for t = 0..999 and s = 0..5
  g3_net.npz         SinDDMNet.forward, dim=160 and dim=32, odd sizes, distinct t
  g4_block.npz       SinDDMConvBlock fwd + input grad + weight grads (4 block shapes, dim=32)
  g5_losses.npz      p_losses value + all 52 grads at s=0 and s=2 (dim=32)
  g17_loss_types.npz p_losses with loss_type 'l2' / 'l1_pred_img' (models.py:595-607): values + three gradient tensors
  g6_psample.npz     p_sample single steps (s=0/s>0, t=17/t=0) with recorded noise
  g7_qsample.npz     q_sample
  g8_bilinear.npz    F.interpolate(mode='bilinear') at the pyramid ratios
  g9_chain_c1.npz    full C1 chain (3 scales, T=100, B=1, dim=160), hash noise
  g10_train.npz      20 train() steps with injected (s, t, noise), dim=32
  g11_img_scales.json create_img_scales() integer/float64 bookkeeping for all datasets
  c1_pyramid.npz     the C1 balloons pyramid (uint8 images) the trainer fixtures use
  g12_model-1.pt     a checkpoint WRITTEN BY the reference trainer's save() after 3 train() steps (dim=16)
  g12_ckpt.npz       what the reference computes from that checkpoint (EMA net forward, one p_sample step)
  g14_chain_c2.npz   full C2 chain (5 scales, T=1000, B=1, dim=160: 2 478 chained evaluations), hash noise
  g18_chain_c3.npz   full C3 chain (6 scales, T=1000, finest 411x512: 2 551 evaluations -- the workload bench.py is quoted on), hash noise
                     (`python tests/golden/make_golden.py g18`, ~10 CPU-minutes)
  g21_chain_c4.npz   full C4 chain (starry_night, 6 scales, T=1000: 2 693 evaluations), hash noise (`... g21`, ~5 CPU-minutes)
  g20_skimage.npz    dilate_mask / match_histograms by scikit-image itself -- written by make_golden_skimage.py under /opt/conda's Python 3.9
  g19_chain_c5_mul24.npz  full C5 chain sampled with scale_mul=(2,4) (92x276 ... 364x1092: 2 521 evaluations), hash noise
                     (`... g19`, ~40 CPU-minutes)
  g16_clip_roi.npz    trainer.clip_roi_sampling (trainer.py:412-468) with the synthetic score: ROI ascent + 5 reverse steps
  g15_clip_guided.npz CLIP-guided p_sample steps (models.py:367-431) with a SYNTHETIC differentiable score in place of
                     CLIP (clip/ is out of scope): mask creation, sub-iterations, lambda blending across steps, dim=32
  g13_roi_i2i.npz    ROI-guided p_sample steps (roi_patch_modification) and an image2image (style-transfer path,
                     no mask / no histogram matching: scikit-image is absent) chain, dim=32, hash noise
"""
import contextlib
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from sinddm_amd.synth import closed_form_state_dict, closed_form_tensor, hash_randn, noise_key  # noqa: E402


# ---------------------------------------------------------------------------
# stub modules for what the image lacks, then import the reference
# ---------------------------------------------------------------------------
def _install_stubs():
    from PIL import Image  # noqa: F401

    sk = types.ModuleType("skimage")
    sk.morphology = types.ModuleType("skimage.morphology")
    sk.filters = types.ModuleType("skimage.filters")
    sk.exposure = types.ModuleType("skimage.exposure")
    sk.exposure.match_histograms = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    for n in ("skimage", "skimage.morphology", "skimage.filters", "skimage.exposure"):
        sys.modules[n] = sk if n == "skimage" else getattr(sk, n.split(".")[1])

    tv = types.ModuleType("torchvision")
    tv.utils = types.ModuleType("torchvision.utils")
    tv.utils.save_image = lambda *a, **k: None
    tv.transforms = types.ModuleType("torchvision.transforms")

    class ToTensor:
        def __call__(self, pic):
            a = np.asarray(pic, dtype=np.uint8)
            if a.ndim == 2:
                a = a[:, :, None]
            return torch.from_numpy(a.transpose(2, 0, 1).copy()).to(torch.float32).div(255)

    class Lambda:
        def __init__(self, f):
            self.f = f

        def __call__(self, x):
            return self.f(x)

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    tv.transforms.ToTensor, tv.transforms.Lambda, tv.transforms.Compose = ToTensor, Lambda, Compose
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.utils"] = tv.utils
    sys.modules["torchvision.transforms"] = tv.transforms
    # text2live_util.util is imported by the reference trainer for an unused helper
    t2l = types.ModuleType("text2live_util")
    t2l.util = types.ModuleType("text2live_util.util")
    t2l.util.get_augmentations_template = lambda *a, **k: None
    sys.modules["text2live_util"] = t2l
    sys.modules["text2live_util.util"] = t2l.util
    import matplotlib
    matplotlib.use("Agg")


_install_stubs()
sys.path.insert(0, REF)
from SinDDM import functions as rf   # noqa: E402
from SinDDM import models as rm      # noqa: E402
from SinDDM import trainer as rt     # noqa: E402

torch.set_num_threads(8)
DEV = "cpu"


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: tuple(v.shape) for k, v in out.items()})


def ref_net(dim):
    net = rm.SinDDMNet(dim=dim, multiscale=True, device=DEV)
    net.load_state_dict(closed_form_state_dict(dim))
    return net


# the five configs of BASELINE.json made concrete
CONFIGS = {
    "C1": dict(image="balloons/balloons.png", image_size=(126, 94), auto_scale=None, sf=1.411, T=100),
    "C2": dict(image="balloons/balloons.png", image_size=None, auto_scale=50000, sf=1.411, T=1000),
    "C3": dict(image="seascape/seascape.png", image_size=(512, 411), auto_scale=None, sf=1.5, T=1000),
    "C4": dict(image="starry_night/starry_night.png", image_size=(252, 198), auto_scale=None, sf=1.3, T=1000),
    "C5": dict(image="marinabaysands/marinabaysands.png", image_size=None, auto_scale=50000, sf=1.411, T=1000),
}


def run_create_img_scales(cfg, workdir):
    """Copy the dataset image to a scratch dir (the reference writes scale_i/ next to it)."""
    folder, fname = cfg["image"].split("/")
    dst = os.path.join(workdir, folder) + "/"
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(REF, "datasets", cfg["image"]), dst + fname)
    sizes, losses, sf, n = rf.create_img_scales(dst, fname, scale_factor=cfg["sf"], image_size=cfg["image_size"],
                                                create=True, auto_scale=cfg["auto_scale"])
    return dst, fname, sizes, [float(x) for x in losses], float(sf), int(n)


def make_diffusion(net, sizes, losses, sf, n, T, **kw):
    return rm.MultiScaleGaussianDiffusion(
        denoise_fn=net, n_scales=n, scale_factor=sf, image_sizes=sizes, channels=3, timesteps=T,
        train_full_t=True, scale_losses=losses, loss_factor=1, loss_type="l1", device=DEV,
        reblurring=True, sample_limited_t=False, omega=0, results_folder=tempfile.mkdtemp(), **kw)


# ---------------------------------------------------------------------------
def g1_g11(workdir):
    meta = {}
    arrays = {}
    for name, cfg in CONFIGS.items():
        dst, fname, sizes, losses, sf, n = run_create_img_scales(cfg, os.path.join(workdir, name))
        net = ref_net(32)
        d = make_diffusion(net, sizes, losses, sf, n, cfg["T"])
        meta[name] = dict(sizes=[list(map(int, s)) for s in sizes], rescale_losses=losses, scale_factor=sf,
                          n_scales=n, T=cfg["T"], num_timesteps_ideal=d.num_timesteps_ideal,
                          num_timesteps_trained=d.num_timesteps_trained,
                          image_sizes_hw=[list(map(int, s)) for s in d.image_sizes],
                          image_size_arg=cfg["image_size"], auto_scale=cfg["auto_scale"], sf_in=cfg["sf"],
                          image=cfg["image"])
        arrays[f"{name}_gammas"] = d.gammas
        if name in ("C1", "C2"):
            for b in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                      "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
                      "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                      "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
                arrays[f"T{cfg['T']}_{b}"] = getattr(d, b)
    save("g1_schedule.npz", **arrays)

    # G11: create_img_scales bookkeeping for every dataset image under main.py defaults
    from PIL import Image
    ds = {}
    for folder in sorted(os.listdir(os.path.join(REF, "datasets"))):
        p = os.path.join(REF, "datasets", folder)
        imgs = [f for f in os.listdir(p) if f.lower().endswith((".png", ".jpg", ".jpeg"))]
        if not imgs:
            continue
        cfg = dict(image=f"{folder}/{imgs[0]}", image_size=None, auto_scale=50000, sf=1.411)
        _, _, sizes, losses, sf, n = run_create_img_scales(cfg, os.path.join(workdir, "g11_" + folder))
        W, H = Image.open(os.path.join(p, imgs[0])).size
        ds[folder] = dict(orig_size=[W, H], sizes=[list(map(int, s)) for s in sizes], scale_factor=sf, n_scales=n,
                          rescale_losses=losses)
    meta["datasets_default"] = ds
    with open(os.path.join(HERE, "g11_img_scales.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote g11_img_scales.json")

    # the C1 pyramid images (uint8) the trainer / chain fixtures use, plus the source image
    from PIL import Image
    c1dir = os.path.join(workdir, "C1", "balloons") + "/"
    pyr = {}
    for i in range(meta["C1"]["n_scales"]):
        pyr[f"scale_{i}"] = np.asarray(Image.open(c1dir + f"scale_{i}/balloons.png").convert("RGB"))
        if i > 0:
            pyr[f"scale_{i}_recon"] = np.asarray(Image.open(c1dir + f"scale_{i}_recon/balloons.png").convert("RGB"))
    save("c1_pyramid.npz", **pyr)
    shutil.copy(os.path.join(REF, "datasets/balloons/balloons.png"), os.path.join(HERE, "balloons.png"))
    return meta


def g2():
    emb = rm.SinusoidalPosEmb(32)
    t = torch.arange(1000)
    s = torch.arange(6, dtype=torch.float32)
    save("g2_posemb.npz", t_emb=emb(t), s_emb=emb(s))


def g3():
    out = {}
    for dim, (H, W) in ((160, (37, 41)), (32, (67, 90)), (160, (24, 50))):
        net = ref_net(dim)
        x = closed_form_tensor((2, 3, H, W), phase=0.3, amp=1.2)
        t = torch.tensor([17, 3], dtype=torch.long)
        for s in (0, 2):
            with torch.no_grad():
                y = net(x, t, scale=s)
            out[f"d{dim}_{H}x{W}_s{s}"] = y
    # cond vector itself (dim-independent parts)
    net = ref_net(160)
    t = torch.tensor([0, 1, 17, 99, 999], dtype=torch.long)
    with torch.no_grad():
        ts = torch.cat((net.SinEmbTime(t), net.SinEmbScale(torch.ones(5) * 3)), dim=1)
        out["cond_vec_s3"] = net.time_mlp(ts)
    save("g3_net.npz", **out)


def g4():
    dim = 32
    net = ref_net(dim)
    out = {}
    cond = closed_form_tensor((2, 32), phase=1.0, amp=0.7)
    for name, blk in (("l1", net.l1), ("l2", net.l2), ("l3", net.l3), ("l4", net.l4)):
        cin = blk.ds_conv.weight.shape[0]
        x = closed_form_tensor((2, cin, 24, 20), phase=0.11 * cin, amp=0.9).requires_grad_(True)
        y = blk(x, cond)
        gy = closed_form_tensor(tuple(y.shape), phase=2.0, amp=1.0, freq=0.377)
        net.zero_grad()
        y.backward(gy)
        out[f"{name}_y"] = y
        out[f"{name}_gx"] = x.grad
        for pn, p in blk.named_parameters():
            out[f"{name}_g_{pn}"] = p.grad
    save("g4_block.npz", **out)


def _small_diffusion(dim, meta, T=None):
    c1 = meta["C1"]
    net = ref_net(dim)
    sizes = [tuple(s) for s in c1["sizes"]]
    d = make_diffusion(net, sizes, c1["rescale_losses"], c1["scale_factor"], c1["n_scales"], T or c1["T"])
    return net, d


def _pyr_tensor(arr):
    return torch.from_numpy(arr.transpose(2, 0, 1).copy()).to(torch.float32).div(255).mul(2).sub(1)


def g5(meta):
    net, d = _small_diffusion(32, meta)
    pyr = np.load(os.path.join(HERE, "c1_pyramid.npz"))
    out = {}
    B = 2
    for s in (0, 2):
        orig = _pyr_tensor(pyr[f"scale_{s}"])[None].repeat(B, 1, 1, 1)
        recon = _pyr_tensor(pyr[f"scale_{s}_recon"])[None].repeat(B, 1, 1, 1) if s > 0 else orig
        t = torch.tensor([37, 5], dtype=torch.long)
        noise = hash_randn(tuple(orig.shape), noise_key("train", s, 0))
        net.zero_grad()
        if s > 0:
            loss = d.p_losses(recon, t, s, noise=noise, x_orig=orig)
        else:
            loss = d.p_losses(orig, t, s, noise=noise)
        loss.backward()
        out[f"s{s}_loss"] = loss.detach()
        for pn, p in net.named_parameters():
            out[f"s{s}_g_{pn}"] = p.grad.clone()
    save("g5_losses.npz", **out)


def g17(meta=None):
    """p_losses with the two loss types main.py never selects ('l2', 'l1_pred_img'; models.py:595-607): value + two gradient
    tensors, at s = 0 and s = 2, t[0] > 0 and t[0] == 0 (the x_mix_prev = x_orig branch)."""
    if meta is None:
        import json
        meta = json.load(open(os.path.join(HERE, "g11_img_scales.json")))
    pyr = np.load(os.path.join(HERE, "c1_pyramid.npz"))
    out = {}
    B = 2
    for lt in ("l2", "l1_pred_img"):
        net, d = _small_diffusion(32, meta)
        d.loss_type = lt
        for s in (0, 2):
            orig = _pyr_tensor(pyr[f"scale_{s}"])[None].repeat(B, 1, 1, 1)
            recon = _pyr_tensor(pyr[f"scale_{s}_recon"])[None].repeat(B, 1, 1, 1) if s > 0 else orig
            for tag, tt in (("a", [37, 5]), ("b", [0, 9])):
                t = torch.tensor(tt, dtype=torch.long)
                noise = hash_randn(tuple(orig.shape), noise_key("train", s, 7))
                net.zero_grad()
                loss = d.p_losses(recon, t, s, noise=noise, x_orig=orig) if s > 0 else d.p_losses(orig, t, s, noise=noise)
                loss.backward()
                out[f"{lt}_s{s}{tag}_loss"] = loss.detach()
                for pn in ("final_conv.0.weight", "l2.net.0.weight", "l1.ds_conv.weight"):
                    out[f"{lt}_s{s}{tag}_g_{pn}"] = dict(net.named_parameters())[pn].grad.clone()
    save("g17_loss_types.npz", **out)


def g6_g7(meta):
    net, d = _small_diffusion(32, meta)
    out = {}
    for s, (H, W) in ((0, (48, 64)), (2, (94, 126))):
        for t in (17, 1, 0):
            x = closed_form_tensor((2, 3, H, W), phase=0.5 + t, amp=1.1)
            d.img_prev_upsample = closed_form_tensor((2, 3, H, W), phase=2.5, amp=0.8, freq=0.211)
            z = hash_randn((2, 3, H, W), noise_key("step", s, t))
            rm.noise_like = lambda shape, device, repeat=False, _z=z: _z
            tt = torch.full((2,), t, dtype=torch.long)
            y = d.p_sample(x, tt, s)
            out[f"psample_s{s}_t{t}"] = y
            with torch.no_grad():
                out[f"eps_s{s}_t{t}"] = net(x, tt, scale=s)
    save("g6_psample.npz", **out)
    # G7
    x0 = closed_form_tensor((3, 3, 20, 30), phase=0.2)
    nz = hash_randn((3, 3, 20, 30), 77)
    t = torch.tensor([0, 41, 99], dtype=torch.long)
    save("g7_qsample.npz", y=d.q_sample(x0, t, noise=nz))


def g8():
    import torch.nn.functional as F
    out = {}
    for (h, w), (H, W), C in (((48, 64), (67, 90), 3), ((133, 177), (186, 248), 1), ((46, 69), (92, 276), 2),
                              ((67, 90), (94, 126), 3)):
        x = closed_form_tensor((1, C, h, w), phase=0.9, amp=1.0, freq=0.271)
        out[f"{h}x{w}_to_{H}x{W}"] = F.interpolate(x, size=(H, W), mode="bilinear")
    # white-noise strip at the source coordinates config C5 reaches (--scale_mul 2 4: 776 -> 1092 columns): one fp32 ulp
    # of the coordinate is 6e-5 there, which separates a single-rounding (fma) source index from a two-rounding one
    x = (hash_randn((1, 1, 8, 776), 901) * 0.6).clamp(-1, 1)
    out["hash_8x776_to_11x1092"] = F.interpolate(x, size=(11, 1092), mode="bilinear")
    save("g8_bilinear.npz", **out)


class NoiseFeeder:
    """Feeds hash noise into the reference's three draw sites in call order."""

    def __init__(self, plan):
        self.plan = list(plan)
        self.i = 0

    def next(self, shape):
        kind, s, t = self.plan[self.i]
        self.i += 1
        return hash_randn(tuple(shape), noise_key(kind, s, t))


@contextlib.contextmanager
def patched_noise(feeder):
    o_randn, o_randn_like, o_noise_like = torch.randn, torch.randn_like, rm.noise_like
    torch.randn = lambda *shape, **kw: feeder.next(shape[0] if isinstance(shape[0], (tuple, list, torch.Size)) else shape)
    torch.randn_like = lambda x, **kw: feeder.next(x.shape)
    rm.noise_like = lambda shape, device, repeat=False: feeder.next(shape)
    try:
        yield
    finally:
        torch.randn, torch.randn_like, rm.noise_like = o_randn, o_randn_like, o_noise_like


def g9(meta):
    net, d = _small_diffusion(160, meta)
    c1 = meta["C1"]
    ideal = d.num_timesteps_ideal
    plan = [("init", 0, 0)] + [("step", 0, t) for t in reversed(range(c1["T"]))]
    for s in range(1, c1["n_scales"]):
        plan += [("renoise", s, 0)] + [("step", s, t) for t in reversed(range(ideal[s]))]
    feeder = NoiseFeeder(plan)
    outs = []
    with patched_noise(feeder), torch.no_grad():
        img = d.sample(batch_size=1, s=0)
        outs.append(img)
        for s in range(1, c1["n_scales"]):
            img = d.sample_via_scale(1, outs[-1], s=s, scale_mul=(1, 1), custom_sample=True,
                                     custom_img_size_idx=s, custom_t=ideal[1:][s - 1])
            outs.append(img)
    assert feeder.i == len(plan), (feeder.i, len(plan))
    save("g9_chain_c1.npz", **{f"out_s{i}": o for i, o in enumerate(outs)},
         plan_len=np.array(len(plan)), ideal=np.array(ideal))



def g14():
    """Full C2 chain: 5 scales, T=1000, B=1, dim=160 = 2 478 chained reference evaluations (the length the
    headline config runs), hash noise.  Also the chain restarted at every scale from the reference's own previous-scale
    output (same thing, stored once) and snapshots of the running sample every 250 steps of scale 0."""
    with open(os.path.join(HERE, "g11_img_scales.json")) as f:
        c2 = json.load(f)["C2"]
    net = ref_net(160)
    sizes = [tuple(s) for s in c2["sizes"]]
    d = make_diffusion(net, sizes, c2["rescale_losses"], c2["scale_factor"], c2["n_scales"], c2["T"])
    ideal = d.num_timesteps_ideal
    assert ideal == c2["num_timesteps_ideal"]
    plan = [("init", 0, 0)] + [("step", 0, t) for t in reversed(range(c2["T"]))]
    for s in range(1, c2["n_scales"]):
        plan += [("renoise", s, 0)] + [("step", s, t) for t in reversed(range(ideal[s]))]
    feeder = NoiseFeeder(plan)
    snaps = {}
    o_ps = d.p_sample

    def rec_p_sample(x, t, s, *a, **k):
        y = o_ps(x, t, s, *a, **k)
        ti = int(t[0])
        if ti % 250 == 0:
            snaps[f"snap_s{int(s)}_t{ti}"] = y.detach().clone()
        return y

    d.p_sample = rec_p_sample
    outs = []
    import time
    t0 = time.time()
    with patched_noise(feeder), torch.no_grad():
        img = d.sample(batch_size=1, s=0)
        outs.append(img)
        print("g14 scale 0 done", time.time() - t0, flush=True)
        for s in range(1, c2["n_scales"]):
            img = d.sample_via_scale(1, outs[-1], s=s, scale_mul=(1, 1), custom_sample=True,
                                     custom_img_size_idx=s, custom_t=ideal[1:][s - 1])
            outs.append(img)
            print("g14 scale", s, "done", time.time() - t0, flush=True)
    assert feeder.i == len(plan), (feeder.i, len(plan))
    save("g14_chain_c2.npz", **{f"out_s{i}": o for i, o in enumerate(outs)}, **snaps,
         plan_len=np.array(len(plan)), ideal=np.array(ideal))


def g_chain(cfg_name, out_name, scale_mul=(1, 1)):
    """Full chain of a headline config from the REFERENCE: T=1000, B=1, dim=160, hash noise -- G14's recipe for any
    config of g11_img_scales.json.  G18 = C3 (6 scales, finest 411x512: 2 551 chained evaluations, the workload bench.py
    is quoted on); G19 = C5 with scale_mul=(2, 4) (the odd 364x1092 geometry, reference sample_via_scale size selection
    models.py:549-568 with custom_sample=False).  Stores the per-scale outputs (float32, compressed)."""
    with open(os.path.join(HERE, "g11_img_scales.json")) as f:
        c = json.load(f)[cfg_name]
    net = ref_net(160)
    sizes = [tuple(s) for s in c["sizes"]]
    d = make_diffusion(net, sizes, c["rescale_losses"], c["scale_factor"], c["n_scales"], c["T"],
                       scale_mul=tuple(scale_mul))
    ideal = d.num_timesteps_ideal
    assert ideal == c["num_timesteps_ideal"]
    plan = [("init", 0, 0)] + [("step", 0, t) for t in reversed(range(c["T"]))]
    for s in range(1, c["n_scales"]):
        plan += [("renoise", s, 0)] + [("step", s, t) for t in reversed(range(ideal[s]))]
    feeder = NoiseFeeder(plan)
    outs = []
    import time
    t0 = time.time()
    with patched_noise(feeder), torch.no_grad():
        if tuple(scale_mul) == (1, 1):
            img = d.sample(batch_size=1, s=0)
        else:
            # trainer.py:247-252: scale 0 of a scale_mul run is sampled at the multiplied size
            img = d.sample(batch_size=1, scale_0_size=(int(d.image_sizes[0][0] * scale_mul[0]),
                                                       int(d.image_sizes[0][1] * scale_mul[1])), s=0)
        outs.append(img)
        print(out_name, "scale 0 done", tuple(img.shape), time.time() - t0, flush=True)
        for s in range(1, c["n_scales"]):
            if tuple(scale_mul) == (1, 1):
                img = d.sample_via_scale(1, outs[-1], s=s, scale_mul=(1, 1), custom_sample=True,
                                         custom_img_size_idx=s, custom_t=ideal[1:][s - 1])
            else:
                img = d.sample_via_scale(1, outs[-1], s=s, scale_mul=tuple(scale_mul), custom_t=ideal[1:][s - 1])
            outs.append(img)
            print(out_name, "scale", s, "done", tuple(img.shape), time.time() - t0, flush=True)
    assert feeder.i == len(plan), (feeder.i, len(plan))
    save(out_name, **{f"out_s{i}": o for i, o in enumerate(outs)},
         plan_len=np.array(len(plan)), ideal=np.array(ideal), scale_mul=np.array(scale_mul))


def g10(meta, workdir):
    """20 reference train() steps at dim=32, B=2, with injected (s, t, noise)."""
    c1 = meta["C1"]
    net, d = _small_diffusion(32, meta)
    folder = os.path.join(workdir, "C1", "balloons") + "/"
    sizes = [tuple(s) for s in c1["sizes"]]
    tr = rt.MultiscaleTrainer(d, folder=folder, n_scales=c1["n_scales"], scale_factor=c1["scale_factor"],
                              image_sizes=sizes, train_batch_size=2, train_lr=1e-3, train_num_steps=20,
                              gradient_accumulate_every=1, ema_decay=0.995, fp16=False, step_start_ema=6,
                              update_ema_every=2, save_and_sample_every=10 ** 9, avg_window=100,
                              sched_milestones=[5, 12], results_folder=tempfile.mkdtemp(), device=DEV)
    steps = 20
    s_seq = [(7 * i + 1) % c1["n_scales"] for i in range(steps)]
    t_seq = [[(13 * i + 5) % c1["T"], (29 * i + 2) % c1["T"]] for i in range(steps)]
    state = dict(i=0)
    losses, lrs = [], []
    o_mult, o_randint, o_randn_like = torch.multinomial, torch.randint, torch.randn_like

    def f_mult(input, num_samples, **kw):
        return torch.tensor([s_seq[state["i"]]], dtype=torch.long)

    def f_randint(lo, hi, size, **kw):
        return torch.tensor(t_seq[state["i"]], dtype=torch.long)

    def f_randn_like(x, **kw):
        return hash_randn(tuple(x.shape), noise_key("train", s_seq[state["i"]], state["i"]))

    # wrap the model call to record the loss and advance the injection counter
    orig_forward = d.forward

    def rec_forward(x, s, *a, **k):
        loss = orig_forward(x, s, *a, **k)
        losses.append(float(loss.detach()))
        lrs.append(tr.opt.param_groups[0]["lr"])
        return loss

    d.forward = rec_forward
    orig_sched_step = tr.scheduler.step

    def sched_step(*a, **k):
        r = orig_sched_step(*a, **k)
        state["i"] += 1
        return r

    tr.scheduler.step = sched_step
    torch.multinomial, torch.randint, torch.randn_like = f_mult, f_randint, f_randn_like
    try:
        tr.train()
    finally:
        torch.multinomial, torch.randint, torch.randn_like = o_mult, o_randint, o_randn_like
    out = dict(losses=np.array(losses), lrs=np.array(lrs), s_seq=np.array(s_seq), t_seq=np.array(t_seq))
    for pn, p in tr.model.denoise_fn.named_parameters():
        out[f"p_{pn}"] = p.detach().clone()
    for pn, p in tr.ema_model.denoise_fn.named_parameters():
        out[f"ema_{pn}"] = p.detach().clone()
    save("g10_train.npz", **out)


def g12(workdir):
    """Checkpoint interop: the reference trains 3 steps (its own RNG), save()s model-1.pt, load()s it into a fresh
    trainer and evaluates the EMA network / one reverse step on recorded inputs."""
    cfg = CONFIGS["C1"]
    dst, fname, sizes, losses, sf, n = run_create_img_scales(cfg, os.path.join(workdir, "G12"))
    dim = 16

    def build(results):
        net = rm.SinDDMNet(dim=dim, multiscale=True, device=DEV)
        d = make_diffusion(net, sizes, losses, sf, n, cfg["T"])
        tr = rt.MultiscaleTrainer(d, folder=dst, n_scales=n, scale_factor=sf, image_sizes=sizes, train_batch_size=2,
                                  train_lr=1e-3, train_num_steps=3, gradient_accumulate_every=1, ema_decay=0.9,
                                  fp16=False, step_start_ema=1, update_ema_every=1, save_and_sample_every=10 ** 9,
                                  avg_window=1, sched_milestones=[2], results_folder=results, device=DEV)
        return tr

    res = tempfile.mkdtemp()
    torch.manual_seed(20240612)
    tr = build(res)
    tr.train()
    tr.save(1)
    shutil.copy(os.path.join(res, "model-1.pt"), os.path.join(HERE, "g12_model-1.pt"))
    # a fresh reference trainer loads the file and evaluates
    tr2 = build(res)
    tr2.load(1)
    em = tr2.ema_model
    x = hash_randn((2, 3, 37, 41), 1201)
    t = torch.tensor([3, 57], dtype=torch.long)
    with torch.no_grad():
        y_ema = em.denoise_fn(x, t, scale=1)
        y_model = tr2.model.denoise_fn(x, t, scale=1)
    s = 1
    H, W = em.image_sizes[s]
    xt = hash_randn((2, 3, H, W), 1202)
    em.img_prev_upsample = hash_randn((2, 3, H, W), 1203).clamp(-1, 1)
    z = hash_randn((2, 3, H, W), 1204)
    o_noise_like = rm.noise_like
    rm.noise_like = lambda shape, device, repeat=False, _z=z: _z
    try:
        with torch.no_grad():
            x_prev = em.p_sample(xt, torch.full((2,), 17, dtype=torch.long), s)
    finally:
        rm.noise_like = o_noise_like
    # inputs are hash noise (keys 1201..1204 above): only the reference's outputs are stored
    save("g12_ckpt.npz", t=t, y_ema=y_ema, y_model=y_model, step=np.array(tr2.step),
         lr=np.array(tr2.opt.param_groups[0]["lr"]), sched_last_epoch=np.array(tr2.scheduler.last_epoch),
         sizes=np.array(sizes), losses=np.array(losses), sf=np.array(sf), x_prev=x_prev)


def g13(workdir):
    """ROI guided p_sample steps + image2image through the reference."""
    cfg = CONFIGS["C1"]
    dst, fname, sizes, losses, sf, n = run_create_img_scales(cfg, os.path.join(workdir, "G13"))
    net = ref_net(32)
    d = make_diffusion(net, sizes, losses, sf, n, cfg["T"])
    from PIL import Image
    out = {}
    # ---- ROI: two overlapping boxes (finest-scale coordinates [y, x, h, w]), target patch from the pyramid ----
    roi_bbs = [[20, 30, 40, 36], [35, 50, 30, 30]]
    target_roi = [10, 12, 30, 40]
    d.roi_guided_sampling = True
    d.roi_bbs = roi_bbs
    d.roi_target_patch = []
    for s in range(n):
        img = Image.open(os.path.join(dst, f"scale_{s}", fname.rsplit(".", 1)[0] + ".png")).convert("RGB")
        ten = torch.from_numpy(np.asarray(img).transpose(2, 0, 1).copy()).float().div(255).mul(2).sub(1)[None]
        bb = [int(b / np.power(sf, n - s - 1)) for b in target_roi]
        d.roi_target_patch.append(rf.extract_patch(ten, bb))
    o_noise_like = rm.noise_like
    try:
        for s, (H, W) in ((0, (48, 64)), (1, (67, 90))):
            for t in (17, 0):
                x = closed_form_tensor((2, 3, H, W), phase=0.5 + t, amp=1.1)
                d.img_prev_upsample = closed_form_tensor((2, 3, H, W), phase=2.5, amp=0.8, freq=0.211)
                z = hash_randn((2, 3, H, W), noise_key("step", s, t))
                rm.noise_like = lambda shape, device, repeat=False, _z=z: _z
                with torch.no_grad():
                    out[f"roi_psample_s{s}_t{t}"] = d.p_sample(x, torch.full((2,), t, dtype=torch.long), s)
    finally:
        rm.noise_like = o_noise_like
        d.roi_guided_sampling = False
    out["roi_bbs"] = np.array(roi_bbs)
    out["target_roi"] = np.array(target_roi)
    # ---- image2image, style-transfer configuration of main.py:296-322 without histogram matching ----
    d2 = make_diffusion(ref_net(32), sizes, losses, sf, n, cfg["T"])
    tr = rt.MultiscaleTrainer(d2, folder=dst, n_scales=n, scale_factor=sf, image_sizes=sizes, train_batch_size=2,
                              train_lr=1e-3, train_num_steps=1, gradient_accumulate_every=1, ema_decay=0.995,
                              fp16=False, step_start_ema=1, update_ema_every=1, save_and_sample_every=10 ** 9,
                              avg_window=1, sched_milestones=[100], results_folder=tempfile.mkdtemp(), device=DEV)
    i2i = os.path.join(workdir, "G13", "i2i")
    os.makedirs(i2i, exist_ok=True)
    # the input: the training image with swapped channels and a gradient (any RGB image of the finest size)
    base = np.asarray(Image.open(os.path.join(dst, f"scale_{n - 1}", fname.rsplit(".", 1)[0] + ".png")).convert("RGB"))
    inp = base[:, ::-1, ::-1].copy()
    inp[:, :, 0] = (inp[:, :, 0].astype(np.int32) * 3 // 4 + np.arange(inp.shape[1])[None, :] // 2).clip(0, 255)
    Image.fromarray(inp.astype(np.uint8)).save(os.path.join(i2i, "input.png"))
    start_s, start_t = n - 1, 7
    custom_t = [0] * (n - 1) + [start_t]
    feeder = NoiseFeeder([("renoise", start_s, 0)] + [("step", start_s, t) for t in reversed(range(start_t))])
    tr.ema_model.reblurring = True
    saved = []
    o_save = rt.utils.save_image
    rt.utils.save_image = lambda img, *a, **k: saved.append(img.detach().clone())
    try:
        with patched_noise(feeder):
            tr.image2image(input_folder=i2i, input_file="input.png", mask="", hist_ref_path="", batch_size=2,
                           image_name=fname, start_s=start_s, custom_t=custom_t, scale_mul=(1, 1), device=DEV,
                           use_hist=False, save_unbatched=False, auto_scale=50000, mode="style_transfer")
    finally:
        rt.utils.save_image = o_save
    out["i2i_input"] = inp.astype(np.uint8)
    out["i2i_final"] = saved[-1]
    out["i2i_custom_t"] = np.array(custom_t)
    out["i2i_gamma_row_after"] = tr.ema_model.gammas[start_s - 1].detach().clone()
    save("g13_roi_i2i.npz", **out)


class SyntheticScore:
    """Stands in for clip.ClipExtractor: `calculate_clip_loss(img in [0,1], embedding)` -> scalar, differentiable.
    The 'embedding' is an image; the loss is a weighted squared distance after a smooth nonlinearity."""

    def zero_grad(self):
        pass

    def calculate_clip_loss(self, x, emb):
        w = 0.5 + closed_form_tensor(tuple(x.shape), phase=1.3, amp=0.5, freq=0.173).abs()
        return (w * (torch.tanh(2.0 * x) - emb) ** 2).mean() * 3.0


def g15(workdir):
    """The guidance branch of p_mean_variance through the reference (models.py:367-431)."""
    cfg = CONFIGS["C1"]
    dst, fname, sizes, losses, sf, n = run_create_img_scales(cfg, os.path.join(workdir, "G15"))
    d = make_diffusion(ref_net(32), sizes, losses, sf, n, cfg["T"])
    res = str(d.results_folder)
    d.clip_guided_sampling = True
    d.clip_model = SyntheticScore()
    d.guidance_sub_iters = [1, 2, 0]
    d.stop_guidance = 3
    d.quantile = 0.8
    d.clip_strength = 0.3
    d.llambda = 0.2
    out = {}
    o_noise_like = rm.noise_like
    try:
        for s, (H, W), ts in ((0, (48, 64), (17, 16)), (1, (67, 90), (17, 16, 0))):
            d.clip_mask = None
            d.x_recon_prev = None
            d.clip_score = []
            d.text_embedds_hr = closed_form_tensor((2, 3, H, W), phase=0.9, amp=0.4, freq=0.131) + 0.5
            d.text_embedds_lr = closed_form_tensor((2, 3, H, W), phase=2.1, amp=0.3, freq=0.117) + 0.5
            x = closed_form_tensor((2, 3, H, W), phase=0.7 + s, amp=1.1)
            d.img_prev_upsample = closed_form_tensor((2, 3, H, W), phase=2.5, amp=0.8, freq=0.211)
            out[f"x_s{s}"] = x
            for t in ts:
                z = hash_randn((2, 3, H, W), noise_key("step", s, t))
                rm.noise_like = lambda shape, device, repeat=False, _z=z: _z
                x = d.p_sample(x, torch.full((2,), t, dtype=torch.long), s).detach()
                out[f"psample_s{s}_t{t}"] = x
            out[f"clip_mask_s{s}"] = d.clip_mask
            out[f"x_recon_prev_s{s}"] = d.x_recon_prev
            out[f"clip_score_s{s}"] = torch.stack([c.reshape(()) for c in d.clip_score])
    finally:
        rm.noise_like = o_noise_like
        shutil.rmtree(res, ignore_errors=True)
    save("g15_clip_guided.npz", **out)


class SyntheticRoiScore(SyntheticScore):
    """SyntheticScore with the two more members trainer.clip_roi_sampling touches: a config dict and a text 'embedding'
    (an image of the ROI's size, whatever the text)."""
    cfg = {"n_aug": 0}

    def __init__(self, shape):
        self.shape = shape

    def get_text_embedding(self, text, template=None):
        return closed_form_tensor(self.shape, phase=1.7, amp=0.35, freq=0.149) + 0.5


def g16(workdir):
    """trainer.clip_roi_sampling (trainer.py:412-468) through the reference: gradient ascent of an external score on a ROI
    of the training image, the patch pasted back, a few reverse steps of the finest scale on top."""
    cfg = CONFIGS["C1"]
    dst, fname, sizes, losses, sf, n = run_create_img_scales(cfg, os.path.join(workdir, "G16"))
    d = make_diffusion(ref_net(32), sizes, losses, sf, n, cfg["T"])
    tr = rt.MultiscaleTrainer(d, folder=dst, n_scales=n, scale_factor=sf, image_sizes=sizes, train_batch_size=2,
                              train_lr=1e-3, train_num_steps=1, gradient_accumulate_every=1, ema_decay=0.995,
                              fp16=False, step_start_ema=1, update_ema_every=1, save_and_sample_every=10 ** 9,
                              avg_window=1, sched_milestones=[100], results_folder=tempfile.mkdtemp(), device=DEV)
    tr.ema_model.reblurring = True
    B, bb, iters, steps, strength = 2, [30, 20, 40, 56], 6, 5, 0.25
    score = SyntheticRoiScore((B, 3, bb[2], bb[3]))
    feeder = NoiseFeeder([("renoise", n - 1, 0)] + [("step", n - 1, t) for t in reversed(range(steps))])
    saved = []
    o_save = rt.utils.save_image
    rt.utils.save_image = lambda img, *a, **k: saved.append(img.detach().clone())
    try:
        with patched_noise(feeder):
            tr.clip_roi_sampling(score, "a synthetic prompt", strength, B, num_clip_iters=iters,
                                 num_denoising_steps=steps, clip_roi_bb=bb, save_unbatched=False)
    finally:
        rt.utils.save_image = o_save
    save("g16_clip_roi.npz", final=saved[-1], bb=np.array(bb), iters=np.array(iters), steps=np.array(steps),
         strength=np.array(strength))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "g16":
        workdir = tempfile.mkdtemp(prefix="sinddm_golden_")
        try:
            g16(workdir)
        finally:
            shutil.rmtree(workdir, ignore_errors=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g15":
        workdir = tempfile.mkdtemp(prefix="sinddm_golden_")
        try:
            g15(workdir)
        finally:
            shutil.rmtree(workdir, ignore_errors=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g13":
        workdir = tempfile.mkdtemp(prefix="sinddm_golden_")
        try:
            g13(workdir)
        finally:
            shutil.rmtree(workdir, ignore_errors=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g8":
        g8()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g17":
        g17()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g14":
        g14()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g18":
        g_chain("C3", "g18_chain_c3.npz")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g21":
        g_chain("C4", "g21_chain_c4.npz")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g19":
        g_chain("C5", "g19_chain_c5_mul24.npz", scale_mul=(2, 4))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g12":
        workdir = tempfile.mkdtemp(prefix="sinddm_golden_")
        try:
            g12(workdir)
        finally:
            shutil.rmtree(workdir, ignore_errors=True)
        return
    workdir = tempfile.mkdtemp(prefix="sinddm_golden_")
    try:
        meta = g1_g11(workdir)
        g2()
        g3()
        g4()
        g5(meta)
        g6_g7(meta)
        g8()
        g9(meta)
        g10(meta, workdir)
        g12(workdir)
        g13(workdir)
        g15(workdir)
        g16(workdir)
        g14()
        g17(meta)
        g_chain("C3", "g18_chain_c3.npz")
        g_chain("C5", "g19_chain_c5_mul24.npz", scale_mul=(2, 4))
        g_chain("C4", "g21_chain_c4.npz")
    finally:
        shutil.rmtree(workdir, ignore_errors=True)


if __name__ == "__main__":
    main()
