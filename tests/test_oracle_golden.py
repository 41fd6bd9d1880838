"""Pins the CPU oracle (oracle/sinddm_oracle.py) to outputs of the reference itself
(fixtures produced by tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_l2, max_abs
from oracle import sinddm_oracle as O
from sinddm_amd.synth import closed_form_state_dict, closed_form_tensor, hash_randn, noise_key


def _sched(meta, name, T=None):
    c = meta[name]
    return O.make_schedule(T or c["T"], c["n_scales"], c["rescale_losses"], 1, train_full_t=True)


def test_g1_schedule_bit_exact(golden):
    meta = golden("g11_img_scales.json")
    g1 = golden("g1_schedule.npz")
    for name in ("C1", "C2", "C3", "C4", "C5"):
        s = _sched(meta, name)
        assert s["num_timesteps_ideal"] == meta[name]["num_timesteps_ideal"]
        assert s["num_timesteps_trained"] == meta[name]["num_timesteps_trained"]
        assert np.array_equal(s["gammas"].numpy(), g1[f"{name}_gammas"])
    for name, T in (("C1", 100), ("C2", 1000)):
        s = _sched(meta, name)
        for b in O.SCHEDULE_BUFFERS:
            assert np.array_equal(s[b].numpy(), g1[f"T{T}_{b}"]), b


def test_g2_posemb(golden):
    g = golden("g2_posemb.npz")
    assert max_abs(O.sinusoidal_emb(torch.arange(1000)), g["t_emb"]) == 0.0
    assert max_abs(O.sinusoidal_emb(torch.arange(6, dtype=torch.float32)), g["s_emb"]) == 0.0


@pytest.mark.parametrize("dim,H,W", [(160, 37, 41), (32, 67, 90), (160, 24, 50)])
def test_g3_net_forward(golden, dim, H, W):
    g = golden("g3_net.npz")
    sd = closed_form_state_dict(dim)
    x = closed_form_tensor((2, 3, H, W), phase=0.3, amp=1.2)
    t = torch.tensor([17, 3])
    for s in (0, 2):
        y = O.net_forward(sd, x, t, s)
        assert rel_l2(y, g[f"d{dim}_{H}x{W}_s{s}"]) < 2e-6


def test_g3_cond_vec(golden):
    g = golden("g3_net.npz")
    sd = closed_form_state_dict(160)
    c = O.cond_vector(sd, torch.tensor([0, 1, 17, 99, 999]), 3)
    assert max_abs(c, g["cond_vec_s3"]) < 1e-6


def test_g4_block_forward(golden):
    g = golden("g4_block.npz")
    sd = closed_form_state_dict(32)
    cond = closed_form_tensor((2, 32), phase=1.0, amp=0.7)
    for name, (cin, cout) in zip(("l1", "l2", "l3", "l4"), O.block_channels(32)):
        x = closed_form_tensor((2, cin, 24, 20), phase=0.11 * cin, amp=0.9)
        y = O.conv_block(sd, name, x, cond)
        assert rel_l2(y, g[f"{name}_y"]) < 2e-6


def test_g4_g5_grads_via_autograd(golden):
    """The oracle is differentiable torch code; its autograd grads must match the reference's."""
    meta = golden("g11_img_scales.json")
    g5 = golden("g5_losses.npz")
    pyr = golden("c1_pyramid.npz")
    sched = _sched(meta, "C1")
    to_t = lambda a: torch.from_numpy(a.transpose(2, 0, 1).copy()).float().div(255).mul(2).sub(1)
    for s in (0, 2):
        sd = {k: v.clone().requires_grad_(True) for k, v in closed_form_state_dict(32).items()}
        orig = to_t(pyr[f"scale_{s}"])[None].repeat(2, 1, 1, 1)
        recon = to_t(pyr[f"scale_{s}_recon"])[None].repeat(2, 1, 1, 1) if s > 0 else orig
        t = torch.tensor([37, 5])
        noise = hash_randn(tuple(orig.shape), noise_key("train", s, 0))
        loss = O.p_losses(sched, sd, recon if s > 0 else orig, t, s, noise, x_orig=orig if s > 0 else None)
        loss.backward()
        assert abs(float(loss) - float(g5[f"s{s}_loss"])) < 1e-6
        for k, v in sd.items():
            assert rel_l2(v.grad, g5[f"s{s}_g_{k}"]) < 5e-5, k


def test_g17_loss_types(golden):
    """The two loss types main.py never selects ('l2', 'l1_pred_img', models.py:595-607), both t[0] branches."""
    meta = golden("g11_img_scales.json")
    g = golden("g17_loss_types.npz")
    pyr = golden("c1_pyramid.npz")
    sched = _sched(meta, "C1")
    to_t = lambda a: torch.from_numpy(a.transpose(2, 0, 1).copy()).float().div(255).mul(2).sub(1)
    for lt in ("l2", "l1_pred_img"):
        for s in (0, 2):
            orig = to_t(pyr[f"scale_{s}"])[None].repeat(2, 1, 1, 1)
            recon = to_t(pyr[f"scale_{s}_recon"])[None].repeat(2, 1, 1, 1) if s > 0 else orig
            for tag, tt in (("a", [37, 5]), ("b", [0, 9])):
                sd = {k: v.clone().requires_grad_(True) for k, v in closed_form_state_dict(32).items()}
                noise = hash_randn(tuple(orig.shape), noise_key("train", s, 7))
                loss = O.p_losses(sched, sd, recon if s > 0 else orig, torch.tensor(tt), s, noise,
                                  x_orig=orig if s > 0 else None, loss_type=lt)
                loss.backward()
                key = f"{lt}_s{s}{tag}"
                assert abs(float(loss) - float(g[key + "_loss"])) < 2e-6 * max(1.0, abs(float(g[key + "_loss"]))), key
                for pn in ("final_conv.0.weight", "l2.net.0.weight", "l1.ds_conv.weight"):
                    assert rel_l2(sd[pn].grad, g[f"{key}_g_{pn}"]) < 5e-5, (key, pn)


def test_g6_p_sample(golden):
    meta = golden("g11_img_scales.json")
    g = golden("g6_psample.npz")
    sched = _sched(meta, "C1")
    sd = closed_form_state_dict(32)
    for s, (H, W) in ((0, (48, 64)), (2, (94, 126))):
        for t in (17, 1, 0):
            x = closed_form_tensor((2, 3, H, W), phase=0.5 + t, amp=1.1)
            xt = closed_form_tensor((2, 3, H, W), phase=2.5, amp=0.8, freq=0.211)
            z = hash_randn((2, 3, H, W), noise_key("step", s, t))
            # reverse step alone, fed the reference's own eps: must be (near) exact
            y = O.reverse_step(sched, x, torch.from_numpy(g[f"eps_s{s}_t{t}"]), t, s, z, xt)
            assert max_abs(y, g[f"psample_s{s}_t{t}"]) <= 1e-6, (s, t)
            y2 = O.p_sample(sched, sd, x, t, s, z, xt)
            assert rel_l2(y2, g[f"psample_s{s}_t{t}"]) < 1e-5, (s, t)


def test_g7_q_sample(golden):
    meta = golden("g11_img_scales.json")
    sched = _sched(meta, "C1")
    x0 = closed_form_tensor((3, 3, 20, 30), phase=0.2)
    nz = hash_randn((3, 3, 20, 30), 77)
    y = O.q_sample(sched, x0, torch.tensor([0, 41, 99]), nz)
    assert max_abs(y, golden("g7_qsample.npz")["y"]) == 0.0


def test_g8_bilinear(golden):
    g = golden("g8_bilinear.npz")
    for (h, w), (H, W), C in (((48, 64), (67, 90), 3), ((133, 177), (186, 248), 1), ((46, 69), (92, 276), 2),
                              ((67, 90), (94, 126), 3)):
        x = closed_form_tensor((1, C, h, w), phase=0.9, amp=1.0, freq=0.271)
        y = O.bilinear_upsample(x, (H, W))
        assert max_abs(y, g[f"{h}x{w}_to_{H}x{W}"]) < 2e-6
    # white noise at the source coordinates config C5 reaches: pins the single-rounding source index
    x = (hash_randn((1, 1, 8, 776), 901) * 0.6).clamp(-1, 1)
    assert max_abs(O.bilinear_upsample(x, (11, 1092)), g["hash_8x776_to_11x1092"]) < 1e-6


def test_g9_chain_c1(golden):
    """Full 3-scale C1 chain (193 net evals at dim=160) with hash noise: index bookkeeping
    bit-exact, images within 1e-4 rel-L2 (the north_star tolerance)."""
    meta = golden("g11_img_scales.json")
    g = golden("g9_chain_c1.npz")
    c1 = meta["C1"]
    sched = _sched(meta, "C1")
    assert sched["num_timesteps_ideal"] == list(g["ideal"])
    sd = closed_form_state_dict(160)
    sizes = [tuple(s) for s in c1["image_sizes_hw"]]

    class Noise(dict):
        def __missing__(self, k):
            kind, s = k[0], k[1]
            t = k[2] if len(k) > 2 else 0
            H, W = sizes[s]
            return hash_randn((1, 3, H, W), noise_key(kind, s, t))

    trace = []
    with torch.no_grad():
        outs = O.sample_chain(sched, sd, sizes, Noise(), 1, trace=trace)
    assert sum(len(t[2]) for t in trace) + len(trace) == int(g["plan_len"])
    for i, o in enumerate(outs):
        assert rel_l2(o, g[f"out_s{i}"]) < 1e-4, i


def test_g14_chain_c2_head(golden):
    """G14 (the reference's full C2 chain: T=1000, 5 scales, 2 478 evaluations): the oracle follows the reference over
    the first 251 steps of scale 0 (t = 999..749, snapshot at t = 750) and over the last 60 steps of scale 1
    restarted from the reference's own snapshot -- bounded so the CPU suite stays short; the whole chain is the GPU
    test's job (tests/test_gpu_parity_band.py)."""
    meta = golden("g11_img_scales.json")
    g = golden("g14_chain_c2.npz")
    c2 = meta["C2"]
    sched = _sched(meta, "C2")
    assert sched["num_timesteps_ideal"] == list(g["ideal"])
    sd = closed_form_state_dict(160)
    sizes = [tuple(s) for s in c2["image_sizes_hw"]]
    with torch.no_grad():
        x = hash_randn((1, 3) + sizes[0], noise_key("init", 0, 0))
        for t in range(999, 749, -1):
            x = O.p_sample(sched, sd, x, t, 0, hash_randn((1, 3) + sizes[0], noise_key("step", 0, t)), None)
        assert rel_l2(x, g["snap_s0_t750"]) < 1e-5
        # scale 1: t = 249 .. 0 would be 250 steps; take the tail from the t=250 snapshot down to t = 190, then
        # compare the t = 0 end point of a run that starts from the snapshot (all 250 steps at 67x90 are cheap)
        up = O.bilinear_upsample(torch.from_numpy(g["out_s0"]), sizes[1])
        x = torch.from_numpy(g["snap_s1_t250"])
        for t in range(249, -1, -1):
            x = O.p_sample(sched, sd, x, t, 1, hash_randn((1, 3) + sizes[1], noise_key("step", 1, t)), up)
        assert rel_l2(x, g["out_s1"]) < 1e-5


def test_g18_chain_c3_scale1_restart(golden):
    """G18 (the reference's full C3 chain -- the workload bench.py is quoted on: 6 scales, T=1000, 2 551 evaluations,
    finest 411x512): the oracle re-runs scale 1 (76x95, 543 steps: bilinear upsample of the reference's scale-0 image,
    q_sample at total_t = 543 without the -1, models.py:504-518, then the reverse steps) and must land on the reference's
    scale-1 image; the index bookkeeping of every scale is checked exactly.  Bounded so the CPU suite stays short -- all
    six scales are the GPU test's job (tests/test_gpu_chain_pin.py)."""
    meta = golden("g11_img_scales.json")
    g = golden("g18_chain_c3.npz")
    c3 = meta["C3"]
    sched = _sched(meta, "C3")
    assert sched["num_timesteps_ideal"] == list(g["ideal"]) == [1000, 543, 408, 289, 192, 119]
    assert int(g["plan_len"]) == 1 + sum(sched["num_timesteps_ideal"]) + 5
    sizes = [tuple(s) for s in c3["image_sizes_hw"]]
    for i, hw in enumerate(sizes):
        assert g[f"out_s{i}"].shape == (1, 3) + hw
    sd = closed_form_state_dict(160)
    s = 1
    with torch.no_grad():
        up = O.bilinear_upsample(torch.from_numpy(g["out_s0"]), sizes[s])
        total_t = sched["num_timesteps_ideal"][s]
        x = O.q_sample(sched, up, torch.full((1,), total_t, dtype=torch.long),
                       hash_randn((1, 3) + sizes[s], noise_key("renoise", s, 0)))
        for t in range(total_t - 1, -1, -1):
            x = O.p_sample(sched, sd, x, t, s, hash_randn((1, 3) + sizes[s], noise_key("step", s, t)), up)
    assert rel_l2(x, g["out_s1"]) < 1e-5


def test_g19_chain_c5_scale_mul_bookkeeping(golden):
    """G19 (the reference's C5 chain sampled with scale_mul = (2, 4): trainer.py:247-252, models.py:549-568): the sizes the
    reference actually sampled at -- int() truncation of the stretched pyramid sizes -- are what the oracle's size selection
    gives, and the draw count is 1 + sum(ideal) + (n - 1).  The images themselves are the GPU test's job."""
    meta = golden("g11_img_scales.json")
    g = golden("g19_chain_c5_mul24.npz")
    c5 = meta["C5"]
    sched = _sched(meta, "C5")
    assert sched["num_timesteps_ideal"] == list(g["ideal"]) == [1000, 544, 426, 322, 229]
    assert int(g["plan_len"]) == 1 + sum(sched["num_timesteps_ideal"]) + 4
    sizes = [tuple(s) for s in c5["image_sizes_hw"]]
    want = [(92, 276), (130, 388), (182, 548), (258, 776), (364, 1092)]
    for s, hw in enumerate(sizes):
        got = O.scale_size(sizes, len(sizes), c5["scale_factor"], s, scale_mul=(2, 4))
        assert tuple(got) == want[s] == tuple(g[f"out_s{s}"].shape[2:]), (s, got)
        assert np.isfinite(g[f"out_s{s}"]).all()


def test_g21_chain_c4_scale1_restart(golden):
    """G21 (the reference's full C4 chain): bookkeeping of all six scales exactly, and the oracle re-runs scale 1 (65x82, 499 steps)
    from the reference's scale-0 image onto the reference's scale-1 image."""
    meta = golden("g11_img_scales.json")
    g = golden("g21_chain_c4.npz")
    c4 = meta["C4"]
    sched = _sched(meta, "C4")
    assert sched["num_timesteps_ideal"] == list(g["ideal"]) == [1000, 499, 410, 330, 257, 197]
    assert int(g["plan_len"]) == 1 + sum(sched["num_timesteps_ideal"]) + 5
    sizes = [tuple(s) for s in c4["image_sizes_hw"]]
    for i, hw in enumerate(sizes):
        assert g[f"out_s{i}"].shape == (1, 3) + hw
    sd = closed_form_state_dict(160)
    s = 1
    with torch.no_grad():
        up = O.bilinear_upsample(torch.from_numpy(g["out_s0"]), sizes[s])
        total_t = sched["num_timesteps_ideal"][s]
        x = O.q_sample(sched, up, torch.full((1,), total_t, dtype=torch.long),
                       hash_randn((1, 3) + sizes[s], noise_key("renoise", s, 0)))
        for t in range(total_t - 1, -1, -1):
            x = O.p_sample(sched, sd, x, t, s, hash_randn((1, 3) + sizes[s], noise_key("step", s, t)), up)
    assert rel_l2(x, g["out_s1"]) < 1e-5


def test_adam_and_lr_restatement():
    torch.manual_seed(0)
    p = torch.randn(50)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2, 4], gamma=0.5)
    m, v = torch.zeros(50), torch.zeros(50)
    for n in range(1, 7):
        g = torch.randn(50)
        ref.grad = g.clone()
        lr = O.multistep_lr(1e-3, [2, 4], n)
        assert abs(lr - opt.param_groups[0]["lr"]) < 1e-12
        opt.step()
        sch.step()
        O.adam_step(p, g, m, v, n, lr)
        assert max_abs(p, ref.detach()) < 1e-7


def test_g12_reference_checkpoint(golden):
    """G12: a model-1.pt written by the reference trainer's save(): key set, and the oracle evaluated with its
    (trained, non-closed-form) weights reproduces what the reference computes after load()."""
    import os
    from sinddm_amd.synth import hash_randn
    g = golden("g12_ckpt.npz")
    ck = torch.load(os.path.join(os.path.dirname(__file__), "golden", "g12_model-1.pt"), map_location="cpu",
                    weights_only=False)
    assert set(ck) == {"step", "model", "ema", "sched", "running_loss", "running_scale"}     # trainer.py:162-170
    assert int(ck["step"]) == int(g["step"]) == 3
    x = hash_randn((2, 3, 37, 41), 1201)
    t = torch.from_numpy(g["t"])
    for which, key in (("ema", "y_ema"), ("model", "y_model")):
        sd = {k[len("denoise_fn."):]: v for k, v in ck[which].items() if k.startswith("denoise_fn.")}
        assert len(sd) == 52
        y = O.net_forward(sd, x, t, 1)
        assert rel_l2(y, g[key]) < 2e-6, which
    # the diffusion buffers in the file are the schedule the oracle derives from (T, losses)
    sched = O.make_schedule(100, 3, [float(v) for v in g["losses"]], loss_factor=1, train_full_t=True)
    for name in ("betas", "sqrt_alphas_cumprod", "posterior_mean_coef1", "gammas"):
        assert np.array_equal(ck["ema"][name].numpy(), np.asarray(sched[name], dtype=np.float32)), name
    sd = {k[len("denoise_fn."):]: v for k, v in ck["ema"].items() if k.startswith("denoise_fn.")}
    H, W = 67, 90
    xt, xtil, z = hash_randn((2, 3, H, W), 1202), hash_randn((2, 3, H, W), 1203).clamp(-1, 1), hash_randn((2, 3, H, W), 1204)
    y = O.p_sample(sched, sd, xt, 17, 1, z, xtil)
    assert rel_l2(y, g["x_prev"]) < 5e-6


def _roi_setup(golden):
    g = golden("g13_roi_i2i.npz")
    meta = golden("g11_img_scales.json")["C1"]
    pyr = golden("c1_pyramid.npz")
    n, sf = meta["n_scales"], meta["scale_factor"]
    patches = []
    for s in range(n):
        ten = torch.from_numpy(pyr[f"scale_{s}"].transpose(2, 0, 1).copy()).float().div(255).mul(2).sub(1)[None]
        y, x, h, w = [int(b / np.power(sf, n - s - 1)) for b in g["target_roi"]]
        patches.append(ten[:, :, y:y + h, x:x + w])
    return g, meta, patches


def test_g13_roi_guided_psample(golden):
    """G13: p_sample with roi_guided_sampling (roi_patch_modification at models.py:430-431), s=0 and s=1."""
    from sinddm_amd.synth import closed_form_state_dict, closed_form_tensor, hash_randn, noise_key
    g, meta, patches = _roi_setup(golden)
    n, sf = meta["n_scales"], meta["scale_factor"]
    sched = O.make_schedule(meta["T"], n, meta["rescale_losses"], loss_factor=1, train_full_t=True)
    sd = closed_form_state_dict(32)
    bbs = [list(map(int, bb)) for bb in g["roi_bbs"]]
    for s, (H, W) in ((0, (48, 64)), (1, (67, 90))):
        for t in (17, 0):
            x = closed_form_tensor((2, 3, H, W), phase=0.5 + t, amp=1.1)
            xtil = closed_form_tensor((2, 3, H, W), phase=2.5, amp=0.8, freq=0.211)
            z = hash_randn((2, 3, H, W), noise_key("step", s, t))
            edit = lambda xr, s=s: O.roi_patch_modification(xr, bbs, patches[s], sf, n, s)
            y = O.p_sample(sched, sd, x, t, s, z, xtil, x_recon_edit=edit)
            assert rel_l2(y, g[f"roi_psample_s{s}_t{t}"]) < 5e-6, (s, t)
            # and the guidance really changes the step
            assert rel_l2(O.p_sample(sched, sd, x, t, s, z, xtil), g[f"roi_psample_s{s}_t{t}"]) > 1e-3


def test_g13_image2image_chain(golden):
    """G13: image2image (style-transfer configuration, trainer.py:287-362): gamma row of the start scale zeroed,
    input re-noised to custom_t[start_s] and denoised at the finest scale."""
    from sinddm_amd.synth import closed_form_state_dict, hash_randn, noise_key
    g, meta, _ = _roi_setup(golden)
    n = meta["n_scales"]
    sched = O.make_schedule(meta["T"], n, meta["rescale_losses"], loss_factor=1, train_full_t=True)
    s0 = n - 1
    sched = dict(sched)
    gam = torch.as_tensor(np.asarray(sched["gammas"])).clone()
    gam[s0 - 1] = gam[s0 - 1].clamp(0, 0)                       # trainer.py:326-327
    sched["gammas"] = gam
    assert np.array_equal(g["i2i_gamma_row_after"], np.zeros_like(g["i2i_gamma_row_after"]))
    sd = closed_form_state_dict(32)
    inp = torch.from_numpy(g["i2i_input"].transpose(2, 0, 1).copy()).float().div(255).mul(2).sub(1)
    batch = inp[None].repeat(2, 1, 1, 1)
    total_t = int(g["i2i_custom_t"][s0])
    H, W = inp.shape[1:]
    up = O.bilinear_upsample(batch, (H, W))
    assert torch.equal(up, batch)                                # same-size bilinear resize is the identity
    img = O.q_sample(sched, up, torch.full((2,), total_t, dtype=torch.long), hash_randn((2, 3, H, W), noise_key("renoise", s0, 0)))
    for t in reversed(range(total_t)):
        img = O.p_sample(sched, sd, img, t, s0, hash_randn((2, 3, H, W), noise_key("step", s0, t)), up)
    assert rel_l2((img + 1) * 0.5, g["i2i_final"]) < 1e-5
