"""CPU: host-side logic of the build (schedule/integer bookkeeping, pyramid construction, state-dict
compatibility, optimizer/LR plumbing, sharding arithmetic) against the golden fixtures."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, max_abs
from oracle import sinddm_oracle as O
from sinddm_amd import functions as F
from sinddm_amd.configs import CONFIGS
from sinddm_amd.models import MultiScaleGaussianDiffusion, SinDDMNet
from sinddm_amd.synth import closed_form_state_dict, net_param_shapes


def _diffusion(meta, dim=16, **kw):
    net = SinDDMNet(dim=dim, multiscale=True, device="cpu")
    return MultiScaleGaussianDiffusion(net, n_scales=meta["n_scales"], scale_factor=meta["scale_factor"],
                                       image_sizes=[tuple(s) for s in meta["sizes"]], timesteps=meta["T"],
                                       train_full_t=True, scale_losses=meta["rescale_losses"], loss_factor=1,
                                       loss_type="l1", device="cpu", **kw)


def test_schedule_buffers_and_timestep_bookkeeping_bit_exact(golden):
    meta = golden("g11_img_scales.json")
    g1 = golden("g1_schedule.npz")
    for name in ("C1", "C2", "C3", "C4", "C5"):
        d = _diffusion(meta[name])
        assert d.num_timesteps_ideal == meta[name]["num_timesteps_ideal"]
        assert d.num_timesteps_trained == meta[name]["num_timesteps_trained"]
        assert [list(s) for s in d.image_sizes] == meta[name]["image_sizes_hw"]
        assert np.array_equal(d.gammas.numpy(), g1[f"{name}_gammas"])
        if name in ("C1", "C2"):
            for b in O.SCHEDULE_BUFFERS:
                assert np.array_equal(getattr(d, b).numpy(), g1[f"T{meta[name]['T']}_{b}"]), b
    # buffer names == the reference's (checkpoint compatibility)
    d = _diffusion(meta["C1"])
    bufs = [k for k, _ in d.named_buffers()]
    assert bufs == list(O.SCHEDULE_BUFFERS) + ["gammas"]


def test_configs_table_matches_golden(golden):
    meta = golden("g11_img_scales.json")
    for name, c in CONFIGS.items():
        m = meta[name]
        assert [list(s) for s in c["sizes"]] == m["sizes"]
        assert c["rescale_losses"] == m["rescale_losses"]
        assert c["scale_factor"] == m["scale_factor"]
        assert c["num_timesteps_ideal"] == m["num_timesteps_ideal"]
        assert c["T"] == m["T"]


def test_step_coefs_match_oracle_branches(golden):
    """Host scalars fed to the fused reverse-step kernel reproduce the oracle when applied in numpy."""
    meta = golden("g11_img_scales.json")["C1"]
    d = _diffusion(meta)
    sched = O.make_schedule(meta["T"], meta["n_scales"], meta["rescale_losses"], 1, train_full_t=True)
    x = torch.randn(2, 3, 5, 6, generator=torch.Generator().manual_seed(0))
    eps = torch.randn(2, 3, 5, 6, generator=torch.Generator().manual_seed(1))
    xt = torch.randn(2, 3, 5, 6, generator=torch.Generator().manual_seed(2))
    z = torch.randn(2, 3, 5, 6, generator=torch.Generator().manual_seed(3))
    for s in (0, 1, 2):
        for t in (99, 50, 1, 0):
            k = d.step_coefs(t, s, True)
            assert k.mode == (0 if s == 0 else (1 if t > 0 else 2))
            x0 = k.sqrt_recip_ac_t * x - k.sqrt_recipm1_ac_t * eps
            if k.mode == 0:
                mean = k.coef1_t * x0.clamp(-1, 1) + k.coef2_t * x
            else:
                xp = (x0 - k.gamma_t * xt) / (1 - k.gamma_t)
                if k.mode == 1:
                    mix = (k.gamma_tm1 * xt + (1 - k.gamma_tm1) * xp).clamp(-1, 1)
                    mean = k.sqrt_ac_tm1 * mix + k.sqrt_1m_ac_tm1_mvar * (x - k.sqrt_ac_t * x0.clamp(-1, 1)) / k.sqrt_1m_ac_t
                else:
                    mean = xp.clamp(-1, 1)
            got = mean + k.sigma * z
            ref = O.reverse_step(sched, x, eps, t, s, z, xt)
            assert max_abs(got, ref) <= 1e-4 * max(1.0, float(ref.abs().max())), (s, t)


def test_pyramid_geometry_all_datasets(golden):
    ds = golden("g11_img_scales.json")["datasets_default"]
    assert len(ds) >= 14
    for name, m in ds.items():
        sizes, sf, n, _ = F.pyramid_geometry(tuple(m["orig_size"]), 1.411, 50000)
        assert [list(s) for s in sizes] == m["sizes"], name
        assert n == m["n_scales"] and sf == m["scale_factor"], name


def test_create_img_scales_balloons(golden, tmp_path):
    """G11 with image data: same sizes / losses / files as the reference for C1 and the main.py defaults."""
    import shutil
    meta = golden("g11_img_scales.json")
    pyr = golden("c1_pyramid.npz")
    folder = str(tmp_path / "balloons") + "/"
    os.makedirs(folder)
    shutil.copy(os.path.join(GOLDEN, "balloons.png"), folder + "balloons.png")
    sizes, losses, sf, n = F.create_img_scales(folder, "balloons.png", scale_factor=1.411, image_size=(126, 94),
                                               create=True, auto_scale=None)
    c1 = meta["C1"]
    assert [list(s) for s in sizes] == c1["sizes"] and n == c1["n_scales"] and sf == c1["scale_factor"]
    assert np.allclose(losses, c1["rescale_losses"], rtol=0, atol=1e-12)
    from PIL import Image
    for key in pyr.files:
        got = np.asarray(Image.open(folder + key + "/balloons.png").convert("RGB"))
        assert np.array_equal(got, pyr[key]), key
    sizes, losses, sf, n = F.create_img_scales(folder, "balloons.png", scale_factor=1.411, create=False, auto_scale=50000)
    c2 = meta["C2"]
    assert [list(s) for s in sizes] == c2["sizes"] and np.allclose(losses, c2["rescale_losses"], atol=1e-12)


def test_state_dict_keys_shapes_and_roundtrip(tmp_path):
    net = SinDDMNet(dim=160, multiscale=True, device="cpu")
    sd = net.state_dict()
    exp = net_param_shapes(160)
    assert list(sd.keys()) == list(exp.keys())
    assert all(tuple(sd[k].shape) == exp[k] for k in exp)
    ref = closed_form_state_dict(160)
    net.load_state_dict(ref)
    flat = torch.cat([v.reshape(-1) for v in ref.values()])
    assert torch.equal(net.flat_params, flat)             # parameters ARE views of the flat buffer
    torch.save({"model": net.state_dict()}, tmp_path / "m.pt")
    net2 = SinDDMNet(dim=160, multiscale=True, device="cpu")
    net2.load_state_dict(torch.load(tmp_path / "m.pt")["model"])
    assert torch.equal(net2.flat_params, flat)
    twin = copy.deepcopy(net)
    assert torch.equal(twin.flat_params, flat) and twin.flat_params.data_ptr() != net.flat_params.data_ptr()
    # diffusion checkpoint keys carry the reference's prefix
    meta = dict(n_scales=3, scale_factor=1.4, sizes=[(64, 48), (90, 67), (126, 94)], T=100, rescale_losses=[1.0, 0.7])
    d = _diffusion(meta, dim=16)
    keys = list(d.state_dict().keys())
    assert keys[0] == "betas" and "gammas" in keys and "denoise_fn.l1.ds_conv.weight" in keys
    assert "denoise_fn.final_conv.0.bias" in keys


def test_default_init_ranges():
    """torch default Conv2d/Linear init (kaiming_uniform a=sqrt5 -> U(+-1/sqrt(fan_in)))."""
    torch.manual_seed(0)
    net = SinDDMNet(dim=32, multiscale=True, device="cpu")
    for name, p in net.named_parameters():
        shape = net_param_shapes(32)[name]
        fan_in = int(np.prod(shape[1:])) if name.endswith("weight") else None
        if fan_in:
            assert float(p.abs().max()) <= 1 / np.sqrt(fan_in) + 1e-6, name
        assert float(p.abs().max()) > 0


def test_unsupported_configurations_raise():
    with pytest.raises(NotImplementedError):
        SinDDMNet(dim=32, multiscale=False)
    with pytest.raises(NotImplementedError):
        SinDDMNet(dim=32, multiscale=True, channels=1)
    meta = dict(n_scales=3, scale_factor=1.4, sizes=[(64, 48), (90, 67), (126, 94)], T=100, rescale_losses=[1.0, 0.7])
    d = _diffusion(meta)
    # the CLIP guidance branch is built against an external scorer (models.py:367-421): switched on without one it
    # names what is missing instead of failing inside the step
    d.clip_guided_sampling = True
    with pytest.raises(RuntimeError, match="clip_model"):
        d.p_sample(torch.zeros(1, 3, 4, 4), torch.zeros(1, dtype=torch.long), 0)


def test_target_size_truncation_and_extrapolation(golden):
    meta = golden("g11_img_scales.json")["C5"]
    d = _diffusion(meta)
    got = [d.target_size(s, (2, 4), True, s) for s in range(meta["n_scales"])]
    assert got == [(92, 276), (130, 388), (182, 548), (258, 776), (364, 1092)]
    sizes_hw = [tuple(s) for s in meta["image_sizes_hw"]]
    for idx in (5, 6):
        assert d.target_size(4, (1, 1), True, idx) == O.scale_size(sizes_hw, 5, meta["scale_factor"], 4, (1, 1), True, idx)


def test_shard_sizes():
    from sinddm_amd.dist import shard_sizes
    assert shard_sizes(128, 8) == [16] * 8
    assert shard_sizes(32, 8) == [4] * 8
    assert shard_sizes(10, 4) == [3, 3, 2, 2]
    assert sum(shard_sizes(17, 8)) == 17


def test_helpers():
    assert F.num_to_groups(16, 32) == [16] and F.num_to_groups(70, 32) == [32, 32, 6]
    assert F.default(None, 3) == 3 and F.default(None, lambda: 4) == 4 and F.default(5, 3) == 5
    a = torch.arange(10.0)
    assert F.extract(a, torch.tensor([2, 7]), (2, 3, 4, 4)).shape == (2, 1, 1, 1)
    assert np.array_equal(F.cosine_beta_schedule(100), O.cosine_beta_schedule(100))


def test_roi_edit_maps_compose_sequential_blends():
    """roi_edit_maps folds the reference's sequential box blends (models.py:291-298) into one affine map."""
    from oracle import sinddm_oracle as O
    meta = dict(n_scales=3, scale_factor=1.4, sizes=[(64, 48), (90, 67), (126, 94)], T=100, rescale_losses=[1.0, 0.7])
    d = _diffusion(meta)
    g = torch.Generator().manual_seed(3)
    bbs = [[20, 30, 40, 36], [35, 50, 30, 30], [0, 0, 10, 10]]            # two overlap
    d.roi_bbs = bbs
    d.roi_target_patch = [torch.randn(1, 3, 9 + s, 11 + s, generator=g) for s in range(3)]
    for s, (H, W) in ((0, (48, 64)), (1, (67, 90))):
        x = torch.randn(2, 3, H, W, generator=g)
        w, c = d.roi_edit_maps(s, H, W, "cpu")
        want = O.roi_patch_modification(x, bbs, d.roi_target_patch[s], 1.4, 3, s)
        got = w[None, None] * x + c[None]
        assert torch.allclose(got, want, atol=1e-6)
        assert torch.equal(d.roi_patch_modification(x.clone(), scale=s), got)
        assert d.roi_edit_maps(s, H, W, "cpu")[0] is w                     # cached


def test_match_histograms_properties():
    from sinddm_amd.functions import match_histograms
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, size=(40, 50, 3), dtype=np.uint8)
    ref = (rng.integers(0, 128, size=(30, 20, 3)) + 64).astype(np.uint8)
    out = match_histograms(image=src, reference=ref, channel_axis=2)
    assert out.dtype == np.uint8 and out.shape == src.shape
    assert np.array_equal(match_histograms(src, src), src)                 # matching to itself is the identity
    for ch in range(3):
        assert out[..., ch].min() >= ref[..., ch].min() and out[..., ch].max() <= ref[..., ch].max()
        # monotone in the source level
        lv = [out[..., ch][src[..., ch] == v].max() for v in np.unique(src[..., ch])]
        assert all(a <= b for a, b in zip(lv, lv[1:]))
        assert abs(float(np.median(out[..., ch])) - float(np.median(ref[..., ch]))) <= 2
    with pytest.raises(ValueError):
        match_histograms(src, ref[..., :2])


def test_g20_scikit_image_helpers_pinned(golden):
    """G20: `dilate_mask` / `match_histograms` of the reference's image2image path (SinDDM/functions.py:21-33,
    trainer.py:312-314) computed by scikit-image ITSELF (0.18.3 under /opt/conda's Python 3.9 in the build container; the
    reference pins 0.19.3, same algorithms: tests/golden/make_golden_skimage.py runs the reference's lines on numpy arrays).
    sinddm_amd.functions restates them on scipy.ndimage / numpy: histogram matching must agree to the last bit, the blurred
    masks to rounding (scipy 1.7 there, 1.15 here)."""
    from sinddm_amd.functions import dilate_mask, match_histograms
    g = golden("g20_skimage.npz")
    assert str(g["skimage_version"]).startswith("0.18")
    n = 0
    for key in g.files:
        if not key.startswith("mask_"):
            continue
        name = key[len("mask_"):]
        m = torch.from_numpy(g[key])
        for mode in ("harmonization", "editing"):
            want = g[f"dilate_{name}_{mode}"]
            with np.errstate(invalid="ignore", divide="ignore"):
                got = dilate_mask(m, mode)
            assert got.shape == want.shape == (1, 1) + tuple(m.shape[1:]) and got.dtype == np.float64
            if np.isnan(want).any():                       # the dilated mask covers the image: 0 / 0, in the reference as here
                assert np.isnan(want).all() and np.isnan(got).all(), (name, mode)
            else:
                assert np.abs(got - want).max() < 1e-12, (name, mode, np.abs(got - want).max())
            n += 1
    assert n == 8
    for a, b, o in (("mh_src", "mh_ref", "mh_out"), ("mh_src2", "mh_ref2", "mh_out2"), ("mh_src", "mh_src", "mh_self")):
        got = match_histograms(image=g[a], reference=g[b], channel_axis=2)
        assert got.dtype == np.uint8 and np.array_equal(got, g[o]), (o, int(np.abs(got.astype(int) - g[o].astype(int)).max()))


def test_dilate_mask_properties():
    from sinddm_amd.functions import dilate_mask
    m = torch.zeros(3, 80, 100)
    m[:, 30:40, 45:55] = 1.0
    out = dilate_mask(m, mode="harmonization")
    assert out.shape == (1, 1, 80, 100) and out.dtype == np.float64
    assert out.min() == 0.0 and out.max() == 1.0
    assert out[0, 0, 35, 50] == 1.0                                         # centre of the blob stays saturated
    assert out[0, 0, 35, 50 + 10] > 0.5                                     # inside the 7-pixel dilation
    assert out[0, 0, 35, 50 + 5 + 7 + 20] < 1e-3                            # beyond dilation + 4 sigma
    assert out[0, 0, 0, 0] == 0.0
    big = dilate_mask(m, mode="editing")
    assert (big > 0.5).sum() > (out > 0.5).sum()
    with pytest.raises(ValueError):
        dilate_mask(m, mode="other")


def test_main_parser_modes():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("sinddm_main", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "main.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a = mod.build_parser().parse_args(["--mode", "roi", "--roi_target", "1", "2", "3", "4", "--roi_bbs", "5", "6", "7", "8", "9", "10", "11", "12"])
    assert a.roi_target == [1, 2, 3, 4] and len(a.roi_bbs) == 8
    a = mod.build_parser().parse_args(["--mode", "harmonization"])
    assert a.start_t_harm == 5 and a.start_t_style == 15 and a.harm_mask == "seascape_mask_dragon.png"


def test_bench_refuses_missing_gpus():
    """`python bench.py --gpus N` must become N ranks itself or fail loudly -- never silently run one rank
    (VERDICT r1 item 1).  Without N devices it exits non-zero and says why."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("SINDDM_BENCH_ONE_DEVICE", None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "9"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "exposes only" in (r.stderr + r.stdout)


def test_save_image_single_image_is_not_padded(tmp_path):
    """torchvision.utils.save_image writes a single (3,H,W) / (1,3,H,W) image as it is -- no grid frame
    (reference trainer.py:283-285 writes the final unbatched samples this way); batches get the 2-px grid."""
    from PIL import Image
    from sinddm_amd.trainer import save_image
    img = torch.rand(3, 20, 30)
    save_image(img, str(tmp_path / "a.png"))
    save_image(img[None], str(tmp_path / "b.png"))
    assert Image.open(tmp_path / "a.png").size == (30, 20)
    assert Image.open(tmp_path / "b.png").size == (30, 20)
    a = np.asarray(Image.open(tmp_path / "a.png"))
    assert np.array_equal(a, (img.mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8)).numpy())
    save_image(torch.rand(5, 3, 20, 30), str(tmp_path / "c.png"), nrow=4)
    assert Image.open(tmp_path / "c.png").size == (4 * 32 + 2, 2 * 22 + 2)
