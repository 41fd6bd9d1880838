"""GPU end-to-end tests through the reference-shaped public API: MultiscaleTrainer.sample_scales (incl. --scale_mul
retargeting and size extrapolation), checkpoint save/load, and train() -> sample() on a tiny configuration."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import rel_l2
from oracle import sinddm_oracle as O
from sinddm_amd.synth import closed_form_state_dict, hash_randn, noise_key

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _trainer(golden, tmp_path, dim=32, T=20, batch=2, **kw):
    from sinddm_amd.models import MultiScaleGaussianDiffusion, SinDDMNet
    from sinddm_amd.trainer import MultiscaleTrainer
    meta = golden("g11_img_scales.json")["C1"]
    pyr = golden("c1_pyramid.npz")
    folder = str(tmp_path / "balloons") + "/"
    for key in pyr.files:
        os.makedirs(folder + key, exist_ok=True)
        Image.fromarray(pyr[key]).save(folder + key + "/balloons.png")
    net = SinDDMNet(dim=dim, multiscale=True, device=DEV).to(DEV)
    net.load_state_dict(closed_form_state_dict(dim))
    sizes = [tuple(s) for s in meta["sizes"]]
    d = MultiScaleGaussianDiffusion(net, n_scales=meta["n_scales"], scale_factor=meta["scale_factor"], image_sizes=sizes,
                                    timesteps=T, train_full_t=True, scale_losses=meta["rescale_losses"], loss_factor=1,
                                    loss_type="l1", device=DEV, reblurring=True, omega=0,
                                    results_folder=str(tmp_path / "res")).to(DEV)
    tr = MultiscaleTrainer(d, folder=folder, n_scales=meta["n_scales"], scale_factor=meta["scale_factor"],
                           image_sizes=sizes, train_batch_size=batch, train_lr=1e-3, train_num_steps=6,
                           gradient_accumulate_every=1, step_start_ema=2, update_ema_every=2,
                           save_and_sample_every=10 ** 9, avg_window=2, sched_milestones=[3],
                           results_folder=str(tmp_path / "res"), device=DEV, **kw)
    return tr, meta


def test_sample_scales_matches_oracle_chain(golden, tmp_path):
    """sample_scales (the main.py --mode sample path) == the oracle's chain with the same hash noise, T=20."""
    tr, meta = _trainer(golden, tmp_path)
    d = tr.ema_model
    d.noise_fn = lambda kind, shape, s, t, dev: hash_randn(shape, noise_key(kind, s, t)).to(dev)
    outs = tr.sample_scales(scale_mul=(1, 1), custom_sample=True, batch_size=2, custom_t_list=d.num_timesteps_ideal[1:],
                            desc="t", save_unbatched=True)
    assert len(outs) == meta["n_scales"]
    sched = O.make_schedule(20, meta["n_scales"], meta["rescale_losses"], 1, train_full_t=True)
    sizes = [tuple(s) for s in meta["image_sizes_hw"]]

    class Noise(dict):
        def __missing__(self, k):
            kind, s = k[0], k[1]
            t = k[2] if len(k) > 2 else 0
            return hash_randn((2, 3) + sizes[s], noise_key(kind, s, t))

    with torch.no_grad():
        ref = O.sample_chain(sched, closed_form_state_dict(32), sizes, Noise(), 2)
    for i, (a, b) in enumerate(zip(outs, ref)):
        assert tuple(a.shape) == tuple(b.shape)
        assert rel_l2(a.cpu(), b) < 1e-4, i
    pngs = [f for _, _, fs in os.walk(tmp_path / "res") for f in fs if f.endswith(".png")]
    assert len(pngs) >= meta["n_scales"] + 2          # per-scale grids + unbatched finals


def test_scale_mul_and_extrapolated_sizes(golden, tmp_path):
    tr, meta = _trainer(golden, tmp_path, T=4)
    outs = tr.sample_scales(scale_mul=(2, 1.5), custom_sample=True, batch_size=1, desc="m", save_unbatched=False,
                            save_images=False)
    for i, o in enumerate(outs):
        h, w = meta["image_sizes_hw"][i]
        assert tuple(o.shape) == (1, 3, int(h * 2), int(w * 1.5))
        assert torch.isfinite(o).all()
    d = tr.ema_model
    big = d.sample_via_scale(1, outs[-1], s=2, scale_mul=(1, 1), custom_sample=True, custom_img_size_idx=3, custom_t=2)
    assert tuple(big.shape[2:]) == d.target_size(2, (1, 1), True, 3)      # extrapolated size (models.py:555-558)


def test_train_then_checkpoint_roundtrip(golden, tmp_path):
    tr, meta = _trainer(golden, tmp_path, T=20)
    torch.manual_seed(0)
    tr.train()
    assert tr.step == 6 and len(tr.running_loss) >= 1
    p_before = tr.model.denoise_fn.flat_params.clone()
    e_before = tr.ema_model.denoise_fn.flat_params.clone()
    assert not torch.equal(p_before, e_before)            # EMA lags the model after step_start_ema
    tr.save(1)
    ck = torch.load(str(tmp_path / "res" / "model-1.pt"), weights_only=False)
    assert set(ck.keys()) >= {"step", "model", "ema", "sched", "running_loss", "running_scale"}
    assert "denoise_fn.l3.net.2.weight" in ck["model"] and "gammas" in ck["model"]
    tr2, _ = _trainer(golden, tmp_path, T=20)
    tr2.load(1)
    assert tr2.step == 6
    assert torch.equal(tr2.model.denoise_fn.flat_params, p_before)
    assert torch.equal(tr2.ema_model.denoise_fn.flat_params, e_before)
    # the loaded EMA model samples (packed weights are rebuilt from the loaded parameters)
    x = tr2.ema_model.sample(batch_size=1)
    assert torch.isfinite(x).all()


def test_load_reference_checkpoint(golden, tmp_path):
    """G12 / SURVEY 8(f) row 1: `MultiscaleTrainer.load()` reads a model-N.pt written by the REFERENCE trainer
    (same key names) and the HIP path reproduces what the reference computes from it."""
    import shutil
    from sinddm_amd.models import MultiScaleGaussianDiffusion, SinDDMNet
    from sinddm_amd.trainer import MultiscaleTrainer
    g = golden("g12_ckpt.npz")
    pyr = golden("c1_pyramid.npz")
    folder = str(tmp_path / "balloons") + "/"
    for key in pyr.files:
        os.makedirs(folder + key, exist_ok=True)
        Image.fromarray(pyr[key]).save(folder + key + "/balloons.png")
    res = tmp_path / "res"
    res.mkdir()
    shutil.copy(os.path.join(os.path.dirname(__file__), "golden", "g12_model-1.pt"), str(res / "model-1.pt"))
    sizes = [tuple(int(v) for v in s) for s in g["sizes"]]
    net = SinDDMNet(dim=16, multiscale=True, device=DEV).to(DEV)
    d = MultiScaleGaussianDiffusion(net, n_scales=3, scale_factor=float(g["sf"]), image_sizes=sizes, timesteps=100,
                                    train_full_t=True, scale_losses=[float(v) for v in g["losses"]], loss_factor=1,
                                    loss_type="l1", device=DEV, reblurring=True, omega=0).to(DEV)
    tr = MultiscaleTrainer(d, folder=folder, n_scales=3, scale_factor=float(g["sf"]), image_sizes=sizes,
                           train_batch_size=2, train_lr=1e-3, train_num_steps=3, gradient_accumulate_every=1,
                           ema_decay=0.9, fp16=False, step_start_ema=1, update_ema_every=1,
                           save_and_sample_every=10 ** 9, avg_window=1, sched_milestones=[2],
                           results_folder=str(res), device=DEV)
    tr.load(1)
    assert tr.step == int(g["step"])
    assert tr.scheduler.last_epoch == int(g["sched_last_epoch"])
    assert abs(tr.opt.param_groups[0]["lr"] - float(g["lr"])) < 1e-12
    x = hash_randn((2, 3, 37, 41), 1201).to(DEV)
    t = torch.from_numpy(g["t"]).to(DEV)
    with torch.no_grad():
        assert rel_l2(tr.ema_model.denoise_fn(x, t, scale=1).cpu(), g["y_ema"]) < 1e-5
        assert rel_l2(tr.model.denoise_fn(x, t, scale=1).cpu(), g["y_model"]) < 1e-5
    H, W = 67, 90
    em = tr.ema_model
    em.img_prev_upsample = hash_randn((2, 3, H, W), 1203).clamp(-1, 1).to(DEV)
    z = hash_randn((2, 3, H, W), 1204).to(DEV)
    em.noise_fn = lambda kind, shape, s, t, device: z
    y = em.p_sample(hash_randn((2, 3, H, W), 1202).to(DEV), torch.full((2,), 17, dtype=torch.long, device=DEV), 1)
    assert rel_l2(y.cpu(), g["x_prev"]) < 1e-5


def _roi_patches(golden, meta, target_roi):
    pyr = golden("c1_pyramid.npz")
    n, sf = meta["n_scales"], meta["scale_factor"]
    out = []
    for s in range(n):
        ten = torch.from_numpy(pyr[f"scale_{s}"].transpose(2, 0, 1).copy()).float().div(255).mul(2).sub(1)[None]
        y, x, h, w = [int(b / np.power(sf, n - s - 1)) for b in target_roi]
        out.append(ten[:, :, y:y + h, x:x + w])
    return out


def test_roi_guided_psample_golden(golden, tmp_path):
    """G13: the fused reverse step with ROI guidance (sinddm_reverse_step_edit) vs the reference's p_sample with
    roi_guided_sampling, s=0 (DDPM posterior) and s=1 (re-blur branch), t>0 and t=0."""
    from sinddm_amd.synth import closed_form_tensor
    g = golden("g13_roi_i2i.npz")
    tr, meta = _trainer(golden, tmp_path, T=100)
    d = tr.ema_model
    d.roi_guided_sampling = True
    d.roi_bbs = [list(map(int, bb)) for bb in g["roi_bbs"]]
    d.roi_target_patch = [p.to(DEV) for p in _roi_patches(golden, meta, g["target_roi"])]
    for s, (H, W) in ((0, (48, 64)), (1, (67, 90))):
        for t in (17, 0):
            x = closed_form_tensor((2, 3, H, W), phase=0.5 + t, amp=1.1).to(DEV)
            d.img_prev_upsample = closed_form_tensor((2, 3, H, W), phase=2.5, amp=0.8, freq=0.211).to(DEV)
            d.noise_fn = lambda kind, shape, s_, t_, dev: hash_randn(shape, noise_key("step", s_, t_)).to(dev)
            y = d.p_sample(x, torch.full((2,), t, dtype=torch.long, device=DEV), s)
            assert rel_l2(y.cpu(), g[f"roi_psample_s{s}_t{t}"]) < 1e-5, (s, t)
    # the finest scale is never guided (models.py:430)
    s = 2
    H, W = meta["image_sizes_hw"][s]
    x = closed_form_tensor((1, 3, H, W), phase=0.3, amp=1.0).to(DEV)
    d.img_prev_upsample = closed_form_tensor((1, 3, H, W), phase=2.5, amp=0.8, freq=0.211).to(DEV)
    y_on = d.p_sample(x, torch.full((1,), 5, dtype=torch.long, device=DEV), s)
    d.roi_guided_sampling = False
    y_off = d.p_sample(x, torch.full((1,), 5, dtype=torch.long, device=DEV), s)
    assert torch.equal(y_on, y_off)


def test_image2image_golden(golden, tmp_path):
    """G13: MultiscaleTrainer.image2image (style-transfer configuration of main.py, no histogram matching) vs the
    reference run with the same hash noise."""
    g = golden("g13_roi_i2i.npz")
    tr, meta = _trainer(golden, tmp_path, T=100)
    i2i = tmp_path / "i2i"
    i2i.mkdir()
    Image.fromarray(g["i2i_input"]).save(str(i2i / "input.png"))
    d = tr.ema_model
    d.noise_fn = lambda kind, shape, s, t, dev: hash_randn(shape, noise_key(kind, s, t)).to(dev)
    n = meta["n_scales"]
    outs = tr.image2image(input_folder=str(i2i), input_file="input.png", mask="", hist_ref_path="", batch_size=2,
                          image_name="balloons.png", start_s=n - 1, custom_t=[int(v) for v in g["i2i_custom_t"]],
                          scale_mul=(1, 1), device=DEV, use_hist=False, save_unbatched=True, auto_scale=50000,
                          mode="style_transfer")
    assert len(outs) == 1
    assert float(d.gammas[n - 2].abs().max()) == 0.0                     # trainer.py:326-327
    assert rel_l2(tr.last_i2i_image.cpu(), g["i2i_final"]) < 1e-5
    pngs = [f for _, _, fs in os.walk(tmp_path / "res") for f in fs if "i2i" in f and f.endswith(".png")]
    assert len(pngs) >= 3


def test_roi_guided_sampling_chain_vs_oracle(golden, tmp_path):
    """MultiscaleTrainer.roi_guided_sampling (trainer.py:436-454) over all scales == the oracle chain with the ROI
    blend at every scale but the finest, T=20."""
    tr, meta = _trainer(golden, tmp_path)
    d = tr.ema_model
    d.noise_fn = lambda kind, shape, s, t, dev: hash_randn(shape, noise_key(kind, s, t)).to(dev)
    bbs, target = [[20, 30, 40, 36], [35, 50, 30, 30]], [10, 12, 30, 40]
    outs = tr.roi_guided_sampling(custom_t_list=d.num_timesteps_ideal[1:], target_roi=target, roi_bb_list=bbs,
                                  save_unbatched=False, batch_size=2, scale_mul=(1, 1), save_images=False)
    assert d.roi_guided_sampling is False
    sched = O.make_schedule(20, meta["n_scales"], meta["rescale_losses"], 1, train_full_t=True)
    sizes = [tuple(s) for s in meta["image_sizes_hw"]]

    class Noise(dict):
        def __missing__(self, k):
            kind, s = k[0], k[1]
            t = k[2] if len(k) > 2 else 0
            return hash_randn((2, 3) + sizes[s], noise_key(kind, s, t))

    roi = dict(bbs=bbs, target_patch=_roi_patches(golden, meta, target), scale_factor=meta["scale_factor"])
    with torch.no_grad():
        ref = O.sample_chain(sched, closed_form_state_dict(32), sizes, Noise(), 2, roi=roi)
        plain = O.sample_chain(sched, closed_form_state_dict(32), sizes, Noise(), 2)
    for i, (a, b) in enumerate(zip(outs, ref)):
        assert rel_l2(a.cpu(), b) < 1e-4, i
    assert rel_l2(ref[0], plain[0]) > 1e-2            # the guidance is not a no-op


def test_harmonization_mask_blend(golden, tmp_path):
    """image2image(mode='harmonization'): far outside the (dilated, blurred) mask the result is the input image,
    inside it is the model's sample (trainer.py:299-305,352-355)."""
    g = golden("g13_roi_i2i.npz")
    tr, meta = _trainer(golden, tmp_path, T=100)
    i2i = tmp_path / "i2i"
    i2i.mkdir()
    inp = g["i2i_input"]
    Image.fromarray(inp).save(str(i2i / "input.png"))
    H, W = inp.shape[:2]
    mask = np.zeros((H, W, 3), dtype=np.uint8)
    mask[30:50, 40:70] = 255
    Image.fromarray(mask).save(str(i2i / "mask.png"))
    n = meta["n_scales"]
    outs = tr.image2image(input_folder=str(i2i), input_file="input.png", mask="mask.png", hist_ref_path="", batch_size=2,
                          image_name="balloons.png", start_s=n - 1, custom_t=[0] * (n - 1) + [5], scale_mul=(1, 1),
                          device=DEV, use_hist=False, save_unbatched=False, auto_scale=50000, mode="harmonization",
                          save_images=False)
    final = tr.last_i2i_image.cpu()
    sample = ((outs[-1] + 1) * 0.5).cpu()
    src = torch.from_numpy(inp.transpose(2, 0, 1).copy()).float().div(255)[None].repeat(2, 1, 1, 1)
    assert torch.allclose(final[:, :, :5, :5], src[:, :, :5, :5], atol=1e-6)             # mask == 0
    assert torch.allclose(final[:, :, 38:42, 53:57], sample[:, :, 38:42, 53:57], atol=2e-3)   # mask ~ 1
    from sinddm_amd.functions import dilate_mask
    m = torch.from_numpy(dilate_mask(torch.from_numpy(mask.transpose(2, 0, 1).copy()).float().div(255), "harmonization")).float()
    assert torch.allclose(final, m * sample + (1 - m) * src, atol=2e-6)
    assert torch.isfinite(final).all()


def test_save_interm_dumps(golden, tmp_path):
    """save_interm=True writes the per-step PNG grids of the reference: the running sample (models.py:469-485,520-546) and the
    step's denoised estimate x_recon `denoised_t-TTT_s-S.png` (models.py:360-366) -- recomputed from eps for the dump, the
    step itself still runs the fused kernel; the dumped estimate is checked against the formula of models.py:306-318."""
    import numpy as np
    from PIL import Image
    tr, meta = _trainer(golden, tmp_path, T=4)
    d = tr.ema_model
    d.save_interm = True
    d.results_folder = tmp_path / "interm"
    seen = {}
    orig = d._dump_x_recon
    d._dump_x_recon = lambda xr, t, s: (seen.__setitem__((int(s), int(t)), xr.detach().cpu().clone()), orig(xr, t, s))[1]
    x0 = d.sample(batch_size=1)
    d.sample_via_scale(1, x0, s=1, custom_t=2)
    f0 = sorted(os.listdir(tmp_path / "interm" / "interm_samples_scale_0"))
    f1 = sorted(os.listdir(tmp_path / "interm" / "interm_samples_scale_1"))
    assert f0 == sorted(["input_noise_s-0.png"] + [f"output_t-{i:03}_s-0.png" for i in range(4)]
                        + [f"denoised_t-{i:03}_s-0.png" for i in range(4)])
    assert f1 == sorted(["noisy_input_s_1.png", "output_t-000_s-1.png", "output_t-001_s-1.png",
                         "denoised_t-000_s-1.png", "denoised_t-001_s-1.png"])
    assert set(seen) == {(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1)}
    # the PNG holds (clamp(x_recon) + 1) / 2 (a single image is saved as it is, like torchvision's make_grid does)
    xr = seen[(1, 0)]
    png = np.asarray(Image.open(tmp_path / "interm" / "interm_samples_scale_1" / "denoised_t-000_s-1.png"), dtype=np.float32) / 255
    want = ((xr.clamp(-1, 1) + 1) * 0.5)[0].permute(1, 2, 0).numpy()
    assert png.shape == want.shape and np.abs(png - want).max() <= 1.0 / 255 + 1e-6


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no torch.distributed environment re-launches itself as 2 ranks (here: both on
    the one device, gloo for the timing collectives -- the hooks bench.py documents) and reports n_gpus = 2."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(SINDDM_BENCH_BACKEND="gloo", SINDDM_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--config", "C1", "--batch", "1",
                        "--steps", "3", "--warmup", "1", "--no-cpu", "--no-c2", "--no-train"], capture_output=True,
                       text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["comm_world_size"] == 2 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 2 * line["config"]["batch_per_gpu"]
    assert line["full_sample"]["images"] == 2 * line["config"]["batch_per_gpu"] and line["full_sample"]["finite"]
    assert 0 < line["roofline"]["frac"] <= 1.0


def test_bench_strong_scaling_c4_two_ranks():
    """The N > 1 default of bench.py: C4 at a FIXED global batch split over the ranks (128 -> 16 per GPU on
    8 GPUs).  Two ranks on the one device (gloo for the collectives), global batch 9 -> shards [5, 4]: value counts
    the global batch, the full sample gathers 9 images through the padded all-gather, per-rank min / max times and the
    all-gather time are reported."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(SINDDM_BENCH_BACKEND="gloo", SINDDM_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--global-batch", "9", "--steps", "2",
                        "--warmup", "1", "--no-cpu", "--no-c2", "--no-train", "--no-strong"], capture_output=True,
                       text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["comm_world_size"] == 2 and line["scaling"] == "strong"
    assert line["config"]["workload"].startswith("C4") and line["config"]["shards"] == [5, 4]
    assert line["config"]["global_batch"] == 9 and line["config"]["finest_hw"] == [198, 252]
    assert abs(line["value"] - 9 * 2 / (line["ms_per_step"] * 2e-3)) / line["value"] < 1e-3
    lo, hi = line["ms_per_step_rank_min_max"]
    assert 0 < lo <= hi == line["ms_per_step"]
    fs = line["full_sample"]
    assert fs["images"] == 9 and fs["finite"] and fs["all_gather_seconds"] > 0


def test_bench_default_workload_is_the_same_at_every_n():
    """`bench.py --gpus N` must run ONE workload at every N so that value(N) / value(1) is a speed-up.
    The default line of one rank and of two ranks (both on the one device, gloo for the collectives) name the same
    config.workload (C3 at 64 chains per GPU, weak scaling) and differ in n_gpus / global_batch / timings only; the
    strong-scaling records sit under the same keys at every N."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(SINDDM_BENCH_BACKEND="gloo", SINDDM_BENCH_ONE_DEVICE="1")
    lines = {}
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1",
                            "--no-cpu", "--no-c2", "--no-train", "--no-full", "--no-ab"] + (["--no-strong"] if n == 2 else []),
                           capture_output=True, text=True, env=env, timeout=1500)
        assert r.returncode == 0, r.stderr[-2000:]
        lines[n] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    a, b = lines[1], lines[2]
    assert a["config"]["workload"] == b["config"]["workload"] and a["config"]["workload"].startswith("C3")
    assert a["scaling"] == b["scaling"] == "weak" and a["metric"] == b["metric"] and a["unit"] == b["unit"] and a["dtype"] == b["dtype"]
    assert a["config"]["batch_per_gpu"] == b["config"]["batch_per_gpu"] == 64
    assert (a["n_gpus"], a["comm_world_size"], a["config"]["global_batch"]) == (1, 1, 64)
    assert (b["n_gpus"], b["comm_world_size"], b["config"]["global_batch"]) == (2, 2, 128)
    assert a["config"]["finest_hw"] == b["config"]["finest_hw"] == [411, 512]
    # N = 1 carries the strong-scaling records under the keys an N > 1 line uses: the FULL global batch on the one GPU
    assert a["c4_strong"]["global_batch"] == 128 and a["c4_strong"]["n_gpus"] == 1 and a["c4_strong"]["scaling"] == "strong"
    assert a["c5_strong"]["global_batch"] == 32 and a["c5_strong"]["value"] > 0


def test_integration_option_b_snippet():
    """INTEGRATION.md option B, executed as written: the reference-side ctypes stub (extracted from the markdown) wraps
    a network with the reference's parameter order and must reproduce the oracle; plugged into
    MultiScaleGaussianDiffusion as a FOREIGN denoise_fn it goes through the generic plug point (models.py:356)."""
    import re
    from sinddm_amd.configs import CONFIGS
    from sinddm_amd.models import MultiScaleGaussianDiffusion, SinDDMNet
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(repo, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    code = [b for b in blocks if "class HipSinDDMNet" in b]
    assert len(code) == 1
    code = code[0].replace('C.CDLL("libsinddm_hip.so")', f'C.CDLL({os.path.join(repo, "sinddm_amd", "libsinddm_hip.so")!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md:option-B", "exec"), ns)
    dim = 32
    ref_like = SinDDMNet(dim=dim, multiscale=True, device="cpu")          # parameters in the reference's order
    ref_like.load_state_dict(closed_form_state_dict(dim))
    hip = ns["HipSinDDMNet"](ref_like, dim)
    sd = closed_form_state_dict(dim)
    x = hash_randn((2, 3, 37, 45), 5)
    t = torch.tensor([3, 77])
    y = hip(x.cuda(), t.cuda(), scale=1)
    assert rel_l2(y.cpu(), O.net_forward(sd, x, t, 1)) < 1e-5
    # as a foreign denoiser inside the diffusion class
    meta = CONFIGS["C1"]
    d = MultiScaleGaussianDiffusion(hip, n_scales=3, scale_factor=meta["scale_factor"], image_sizes=meta["sizes"],
                                    timesteps=meta["T"], train_full_t=True, scale_losses=meta["rescale_losses"],
                                    loss_factor=1, loss_type="l1", device="cuda:0", reblurring=True, omega=0).to("cuda:0")
    sched = O.make_schedule(meta["T"], 3, meta["rescale_losses"], 1, train_full_t=True)
    H, W = d.image_sizes[1]
    xs, xt, z = hash_randn((2, 3, H, W), 6), hash_randn((2, 3, H, W), 7) * 0.5, hash_randn((2, 3, H, W), 8)
    d.img_prev_upsample = xt.cuda()
    d.noise_fn = lambda kind, shape, s, tt, dev: z.to(dev)
    got = d.p_sample(xs.cuda(), torch.full((2,), 20, device="cuda:0", dtype=torch.long), 1)
    assert rel_l2(got.cpu(), O.p_sample(sched, sd, xs, 20, 1, z, xt)) < 1e-5


class _SyntheticScore:
    """The stand-in for clip.ClipExtractor that fixture G15 was generated with (tests/golden/make_golden.py)."""
    cfg = {"n_aug": 0}

    def zero_grad(self):
        pass

    def get_text_embedding(self, text, template=None):
        return self.emb[template]

    def calculate_clip_loss(self, x, emb):
        from sinddm_amd.synth import closed_form_tensor
        w = 0.5 + closed_form_tensor(tuple(x.shape), phase=1.3, amp=0.5, freq=0.173).abs().to(x.device)
        return (w * (torch.tanh(2.0 * x) - emb) ** 2).mean() * 3.0


def test_clip_guided_p_sample_golden(golden):
    """G15: the guidance branch of p_mean_variance (reference SinDDM/models.py:367-431) run by the REFERENCE with a
    synthetic differentiable score in place of CLIP: mask creation from the thresholded gradient at the first step,
    sub-iterations, lambda blending of the previous step's x_recon, the t = 0 step; scale 0 (low-res embedding) and
    scale 1 (reblurring branch).  Network on the HIP path, guidance on stock PyTorch autograd."""
    from sinddm_amd.configs import build_diffusion
    from sinddm_amd.synth import closed_form_tensor, hash_randn, noise_key
    g = golden("g15_clip_guided.npz")
    net, d = build_diffusion("C1", dim=32, device=torch.device(DEV))
    d.clip_guided_sampling = True
    d.clip_model = _SyntheticScore()
    d.guidance_sub_iters = [1, 2, 0]
    d.stop_guidance = 3
    d.quantile = 0.8
    d.clip_strength = 0.3
    d.llambda = 0.2
    for s, (H, W), ts in ((0, (48, 64), (17, 16)), (1, (67, 90), (17, 16, 0))):
        d.clip_mask, d.x_recon_prev, d.clip_score = None, None, []
        d.text_embedds_hr = (closed_form_tensor((2, 3, H, W), phase=0.9, amp=0.4, freq=0.131) + 0.5).to(DEV)
        d.text_embedds_lr = (closed_form_tensor((2, 3, H, W), phase=2.1, amp=0.3, freq=0.117) + 0.5).to(DEV)
        x = torch.from_numpy(g[f"x_s{s}"]).to(DEV)
        d.img_prev_upsample = closed_form_tensor((2, 3, H, W), phase=2.5, amp=0.8, freq=0.211).to(DEV)
        d.noise_fn = lambda kind, shape, ss, tt, dev: hash_randn(shape, noise_key("step", ss, tt)).to(dev)
        for t in ts:
            x = d.p_sample(x, torch.full((2,), t, dtype=torch.long, device=DEV), s)
            assert rel_l2(x.cpu(), g[f"psample_s{s}_t{t}"]) < 2e-5, (s, t)
        # the mask is a thresholded comparison: allow a handful of pixels on the threshold to flip
        assert float((d.clip_mask.cpu() != torch.from_numpy(g[f"clip_mask_s{s}"])).float().mean()) < 1e-3
        assert rel_l2(d.x_recon_prev.cpu(), g[f"x_recon_prev_s{s}"]) < 2e-5
        assert rel_l2(torch.stack([c.reshape(()) for c in d.clip_score]), g[f"clip_score_s{s}"]) < 1e-5


def test_clip_guided_save_interm_dumps(tmp_path):
    """save_interm=True in the CLIP-guided branch: the reference's `clip_mask_s-S.png` and
    `clip_out_s-S_t-T_subiter_I.png` (models.py:394-404) next to `denoised_t-TTT_s-S.png` (models.py:360-366); the dumps
    do not change the samples."""
    from sinddm_amd.configs import build_diffusion
    from sinddm_amd.synth import closed_form_tensor, hash_randn, noise_key
    outs = []
    for dump in (False, True):
        net, d = build_diffusion("C1", dim=32, device=torch.device(DEV))
        d.clip_guided_sampling, d.clip_model = True, _SyntheticScore()
        d.guidance_sub_iters, d.stop_guidance, d.quantile, d.clip_strength, d.llambda = [2, 0, 0], 0, 0.8, 0.3, 0.2
        d.save_interm, d.results_folder = dump, tmp_path / "clipdump"
        H, W = 48, 64
        d.clip_mask, d.x_recon_prev, d.clip_score = None, None, []
        d.text_embedds_lr = (closed_form_tensor((2, 3, H, W), phase=2.1, amp=0.3, freq=0.117) + 0.5).to(DEV)
        d.text_embedds_hr = d.text_embedds_lr
        d.noise_fn = lambda kind, shape, ss, tt, dev: hash_randn(shape, noise_key("step", ss, tt)).to(dev)
        x = hash_randn((2, 3, H, W), 91).to(DEV)
        for t in (5, 4):
            x = d.p_sample(x, torch.full((2,), t, dtype=torch.long, device=DEV), 0)
        outs.append(x.cpu())
    assert torch.equal(outs[0], outs[1])
    files = sorted(os.listdir(tmp_path / "clipdump" / "interm_samples_scale_0"))
    assert files == sorted(["clip_mask_s-0.png", "denoised_t-004_s-0.png", "denoised_t-005_s-0.png"]
                           + [f"clip_out_s-0_t-{t}_subiter_{i}.png" for t in (4, 5) for i in (0, 1)])


def test_clip_roi_sampling_golden(golden, tmp_path):
    """G16: MultiscaleTrainer.clip_roi_sampling (reference SinDDM/trainer.py:412-468) run by the REFERENCE with the
    synthetic score: six steps of normalised gradient ascent on a 40x56 region of the training image, the patch pasted
    back, then q_sample to t = 5 and five reverse steps of the finest scale (HIP chain) with the same hash noise."""
    from sinddm_amd.synth import closed_form_tensor
    g = golden("g16_clip_roi.npz")
    tr, meta = _trainer(golden, tmp_path, T=100)
    d = tr.ema_model
    d.noise_fn = lambda kind, shape, s, t, dev: hash_randn(shape, noise_key(kind, s, t)).to(dev)
    bb = [int(v) for v in g["bb"]]
    score = _SyntheticScore()
    score.emb = {"lr": (closed_form_tensor((2, 3, bb[2], bb[3]), phase=1.7, amp=0.35, freq=0.149) + 0.5).to(DEV)}
    before = tr.data_list[meta["n_scales"] - 1][0][0].clone()
    final = tr.clip_roi_sampling(score, "a synthetic prompt", float(g["strength"]), 2, num_clip_iters=int(g["iters"]),
                                 num_denoising_steps=int(g["steps"]), clip_roi_bb=bb, save_unbatched=True)
    assert rel_l2(((final + 1) * 0.5).cpu(), g["final"]) < 1e-5
    assert torch.equal(tr.data_list[meta["n_scales"] - 1][0][0], before)          # the training image itself is untouched
    pngs = [f for _, _, fs in os.walk(tmp_path / "res") for f in fs if f.startswith("clip_roi_") and f.endswith(".png")]
    assert len(pngs) == 3                                                          # the grid + two unbatched images
