"""GPU parity tests of the forward / sampling path: HIP library (through the C ABI) vs the CPU oracle and
vs the golden fixtures produced by the reference.  Tolerances are fp32 reduction-order tolerances."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import max_abs, rel_l2
from oracle import sinddm_oracle as O
from sinddm_amd.synth import closed_form_state_dict, closed_form_tensor, hash_randn, noise_key

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net(dim):
    from sinddm_amd.models import SinDDMNet
    net = SinDDMNet(dim=dim, multiscale=True, device=DEV).to(DEV)
    net.load_state_dict(closed_form_state_dict(dim))
    return net


def _diffusion(golden, dim, cfg="C1", **kw):
    from sinddm_amd.models import MultiScaleGaussianDiffusion
    meta = golden("g11_img_scales.json")[cfg]
    net = _net(dim)
    d = MultiScaleGaussianDiffusion(net, n_scales=meta["n_scales"], scale_factor=meta["scale_factor"],
                                    image_sizes=[tuple(s) for s in meta["sizes"]], timesteps=meta["T"],
                                    train_full_t=True, scale_losses=meta["rescale_losses"], loss_factor=1,
                                    loss_type="l1", device=DEV, reblurring=True, omega=0, **kw).to(DEV)
    sched = O.make_schedule(meta["T"], meta["n_scales"], meta["rescale_losses"], 1, train_full_t=True)
    return net, d, sched, meta


def test_library_loaded_and_layout():
    from sinddm_amd import _lib
    lib = _lib.load()
    assert lib.sinddm_abi_version() == _lib.ABI_VERSION == 3
    assert lib.sinddm_param_count(160) == 1106772


@pytest.mark.parametrize("dim,H,W", [(160, 37, 41), (32, 67, 90), (160, 24, 50)])
def test_net_forward_golden(golden, dim, H, W):
    """G3: SinDDMNet.forward vs the reference's own outputs (distinct t per sample, s=0 and s=2)."""
    g = golden("g3_net.npz")
    net = _net(dim)
    x = closed_form_tensor((2, 3, H, W), phase=0.3, amp=1.2).to(DEV)
    t = torch.tensor([17, 3], device=DEV)
    for s in (0, 2):
        with torch.no_grad():
            y = net(x, t, scale=s)
        assert rel_l2(y.cpu(), g[f"d{dim}_{H}x{W}_s{s}"]) < 1e-5


@pytest.mark.parametrize("dim,B,H,W", [(160, 1, 5, 7), (160, 3, 8, 32), (160, 2, 9, 33), (160, 1, 48, 64),
                                       (32, 2, 17, 100), (16, 1, 33, 31), (160, 2, 94, 126),
                                       (48, 2, 19, 35), (64, 1, 13, 66), (80, 2, 14, 40), (96, 1, 20, 33),
                                       # several work items per workgroup whose tiles sit in DIFFERENT tile columns, one of
                                       # them cut by the right image edge (W = 70 -> 3 tile columns): strided and
                                       # tile-major item orders of the persistent Winograd kernel
                                       (160, 16, 36, 70), (160, 16, 48, 70)])
def test_net_forward_vs_oracle_edges(dim, B, H, W):
    """Ragged / tiny / tile-boundary sizes against the oracle; host-t and device-t paths agree."""
    net = _net(dim)
    sd = closed_form_state_dict(dim)
    x = hash_randn((B, 3, H, W), 11 + H)
    t = torch.arange(B) * 7 + 2
    ref = O.net_forward(sd, x, t, 1)
    with torch.no_grad():
        y = net(x.to(DEV), t.to(DEV), scale=1)
    assert rel_l2(y.cpu(), ref) < 1e-5
    t0 = int(t[0])
    y2 = net.infer(x[:1].to(DEV).contiguous(), None, t0, 1.0)
    assert rel_l2(y2.cpu(), ref[:1]) < 1e-5


def test_q_sample_golden(golden):
    net, d, sched, _ = _diffusion(golden, 32)
    x0 = closed_form_tensor((3, 3, 20, 30), phase=0.2).to(DEV)
    nz = hash_randn((3, 3, 20, 30), 77).to(DEV)
    y = d.q_sample(x0, torch.tensor([0, 41, 99], device=DEV), noise=nz)
    assert max_abs(y.cpu(), golden("g7_qsample.npz")["y"]) <= 1e-6


def test_q_sample_mix_vs_oracle(golden):
    net, d, sched, _ = _diffusion(golden, 32)
    xs = hash_randn((4, 3, 13, 21), 1)
    xo = hash_randn((4, 3, 13, 21), 2)
    nz = hash_randn((4, 3, 13, 21), 3)
    t = torch.tensor([0, 5, 50, 99])
    ref = O.p_losses_inputs(sched, xs, t, 2, nz, xo)
    gamma_row = d.gammas[1].reshape(-1).contiguous()
    y = d._q_sample_impl(xs.to(DEV), t.to(DEV), 0, nz.to(DEV), x_orig=xo.to(DEV), gamma_row=gamma_row)
    assert max_abs(y.cpu(), ref) <= 2e-6


def test_upsample_golden(golden):
    net, d, _, _ = _diffusion(golden, 32)
    g = golden("g8_bilinear.npz")
    for (h, w), (H, W), Cc in (((48, 64), (67, 90), 3), ((133, 177), (186, 248), 1), ((46, 69), (92, 276), 2),
                               ((67, 90), (94, 126), 3)):
        x = closed_form_tensor((1, Cc, h, w), phase=0.9, amp=1.0, freq=0.271).to(DEV)
        y = d.upsample(x, (H, W))
        assert max_abs(y.cpu(), g[f"{h}x{w}_to_{H}x{W}"]) < 2e-5
    x = (hash_randn((1, 1, 8, 776), 901) * 0.6).clamp(-1, 1).to(DEV)      # C5-sized source coordinates, white noise
    assert max_abs(d.upsample(x, (11, 1092)).cpu(), g["hash_8x776_to_11x1092"]) < 1e-6


def test_reverse_step_all_modes_vs_oracle(golden):
    """The fused reverse-step kernel against the oracle for every branch (s=0 / s>0, t>0 / t=0,
    clip on/off, omega 0 / >0, reblurring off)."""
    from sinddm_amd import _lib
    lib = _lib.load()
    net, d, sched, _ = _diffusion(golden, 32)
    shape = (2, 3, 19, 23)
    x = (hash_randn(shape, 5) * 1.3)
    eps = hash_randn(shape, 6)
    xt = hash_randn(shape, 7) * 0.7
    z = hash_randn(shape, 8)
    xd, ed, td_, zd = x.to(DEV), eps.to(DEV), xt.to(DEV), z.to(DEV)   # keep the device copies alive
    for reblur in (True, False):
        for omega in (0.0, 0.3):
            d.reblurring, d.omega = reblur, omega
            for s in (0, 1, 2):
                for t in (99, 17, 1, 0):
                    for clip in (True, False):
                        ref = O.reverse_step(sched, x, eps, t, s, z, xt, reblurring=reblur, omega=omega,
                                             clip_denoised=clip)
                        k = d.step_coefs(t, s, clip)
                        out = torch.empty(shape, device=DEV)
                        rc = lib.sinddm_reverse_step(xd.data_ptr(), ed.data_ptr(), td_.data_ptr(), zd.data_ptr(),
                                                     out.data_ptr(), C.byref(k), out.numel(),
                                                     torch.cuda.current_stream().cuda_stream)
                        assert rc == 0
                        torch.cuda.synchronize()
                        err = max_abs(out.cpu(), ref)
                        assert err <= 2e-5 * max(1.0, float(ref.abs().max())), (reblur, omega, s, t, clip, err)
    d.reblurring, d.omega = True, 0


def test_p_sample_golden(golden):
    """G6: whole p_sample (net + fused step) vs the reference with recorded noise."""
    g = golden("g6_psample.npz")
    net, d, sched, _ = _diffusion(golden, 32)
    for s, (H, W) in ((0, (48, 64)), (2, (94, 126))):
        for t in (17, 1, 0):
            x = closed_form_tensor((2, 3, H, W), phase=0.5 + t, amp=1.1).to(DEV)
            d.img_prev_upsample = closed_form_tensor((2, 3, H, W), phase=2.5, amp=0.8, freq=0.211).to(DEV)
            d.noise_fn = lambda kind, shape, ss, tt, dev: hash_randn(shape, noise_key(kind, ss, tt)).to(dev)
            y = d.p_sample(x, torch.full((2,), t, device=DEV, dtype=torch.long), s)
            assert rel_l2(y.cpu(), g[f"psample_s{s}_t{t}"]) < 1e-5, (s, t)
            with torch.no_grad():
                e = net(x, torch.full((2,), t, device=DEV, dtype=torch.long), scale=s)
            assert rel_l2(e.cpu(), g[f"eps_s{s}_t{t}"]) < 1e-5, (s, t)


def test_full_chain_c1_golden(golden):
    """G9: full 3-scale C1 chain (T=100, B=1, dim=160, 193 net evaluations) through the public
    sample()/sample_via_scale() API with hash noise: within the north_star's 1e-4 rel-L2 of the
    reference's images."""
    g = golden("g9_chain_c1.npz")
    net, d, sched, meta = _diffusion(golden, 160)
    assert d.num_timesteps_ideal == list(g["ideal"])
    d.noise_fn = lambda kind, shape, s, t, dev: hash_randn(shape, noise_key(kind, s, t)).to(dev)
    outs = [d.sample(batch_size=1, s=0)]
    for s in range(1, meta["n_scales"]):
        outs.append(d.sample_via_scale(1, outs[-1], s=s, scale_mul=(1, 1), custom_sample=True,
                                       custom_img_size_idx=s, custom_t=d.num_timesteps_ideal[1:][s - 1]))
    for i, o in enumerate(outs):
        assert tuple(o.shape[2:]) == tuple(meta["image_sizes_hw"][i])
        assert rel_l2(o.cpu(), g[f"out_s{i}"]) < 1e-4, i


def test_sampler_properties_full_size(golden):
    """Size-independent properties at the C2 finest size (186x248, B=4): batch independence (each
    sample's result does not depend on its neighbours, up to the rounding of the kernel variant) and determinism."""
    net, d, sched, meta = _diffusion(golden, 160, cfg="C2")
    H, W = meta["image_sizes_hw"][-1]
    s = meta["n_scales"] - 1
    x = hash_randn((4, 3, H, W), 21).to(DEV)
    d.img_prev_upsample = hash_randn((4, 3, H, W), 22).to(DEV) * 0.5
    d.noise_fn = lambda kind, shape, ss, tt, dev: hash_randn((4, 3, H, W), 23)[: shape[0]].to(dev)
    y4 = d._p_sample_host_t(x, 100, s)
    y4b = d._p_sample_host_t(x, 100, s)
    assert torch.equal(y4, y4b)
    d.img_prev_upsample = d.img_prev_upsample[:1].contiguous()
    y1 = d._p_sample_host_t(x[:1].contiguous(), 100, s)
    # (not bit-level: the library picks the Winograd variant -- F(2x2) or F(2x4) -- by launch size, like any conv
    # library picks its algorithm; what must hold is that a sample does not see its neighbours)
    assert rel_l2(y1.cpu(), y4[:1].cpu()) < 5e-6
    assert torch.isfinite(y4).all()


def test_large_batch_index_range():
    """C3 finest scale (B=64, 411x512): one 160-channel tensor has 2.15e9 elements > 2^31 -- the tail samples of the
    big batch must equal the same samples run alone (32-bit index overflow would corrupt exactly those).  The big batch
    takes conv_wh (binary16 Winograd), two samples alone the fp32 Winograd kernels: equal to rounding under the default
    path (an index error is O(1)), bit for bit with the binary16 kernels switched off (same kernel family both ways) and
    bit for bit between the binary16 batch and a 16-sample sub-batch that also takes conv_wh (per-sample scales: a
    sample does not see its neighbours)."""
    from sinddm_amd import _lib
    lib = _lib.load()
    net = _net(160)
    B, H, W = 64, 411, 512
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(B, 3, H, W, device=DEV, generator=g)
    t = torch.randint(0, 1000, (B,), device=DEV, generator=g)
    assert lib.sinddm_debug_infer_path(160, B, H, W) == 8 and lib.sinddm_debug_infer_path(160, 16, H, W) == 8
    with torch.no_grad():
        y = net(x, t, scale=5)
        y_tail = net(x[-2:].contiguous(), t[-2:].contiguous(), scale=5)
        y_head = net(x[:2].contiguous(), t[:2].contiguous(), scale=5)
        y_sub = net(x[-16:].contiguous(), t[-16:].contiguous(), scale=5)
    assert torch.isfinite(y).all()
    assert rel_l2(y[-2:].cpu(), y_tail.cpu()) < 2e-6 and rel_l2(y[:2].cpu(), y_head.cpu()) < 2e-6
    assert torch.equal(y[-16:], y_sub)
    del y_sub
    net.fp32_convs = True                      # the same launches on the fp32 matrix pipe (SINDDM_DIM_FP32_CONVS)
    try:
        with torch.no_grad():
            y0 = net(x, t, scale=5)
            y0_tail = net(x[-2:].contiguous(), t[-2:].contiguous(), scale=5)
            y0_head = net(x[:2].contiguous(), t[:2].contiguous(), scale=5)
        assert torch.equal(y0[-2:], y0_tail) and torch.equal(y0[:2], y0_head)
        assert rel_l2(y.cpu()[::21], y0.cpu()[::21]) < 2e-6
    finally:
        net.fp32_convs = False


@pytest.mark.parametrize("dim", [20, 28, 10])
def test_dims_with_channels_not_multiple_of_4(dim):
    """dim / 2 or dim not a multiple of 4 (main.py exposes --dim): the 3x3 convs whose C_in % 4 != 0 must take
    the direct kernel (the packed Winograd image has the second-generation layout, which the first-generation kernel --
    the only one that accepts such C_in -- cannot read).  Forward and all gradients against the oracle."""
    from sinddm_amd.models import SinDDMNet
    net = SinDDMNet(dim=dim, multiscale=True, device=DEV).to(DEV)
    net.load_state_dict(closed_form_state_dict(dim))
    net.bind_grads()
    net.flat_grads.zero_()
    B, H, W = 2, 21, 38
    x = hash_randn((B, 3, H, W), 40 + dim)
    gy = hash_randn((B, 3, H, W), 41 + dim) / (B * 3 * H * W)
    t = torch.tensor([3, 77])
    xd = x.to(DEV).requires_grad_(True)
    y = net(xd, t.to(DEV), scale=1)
    y.backward(gy.to(DEV))
    sd = {k: v.clone().requires_grad_(True) for k, v in closed_form_state_dict(dim).items()}
    xc = x.clone().requires_grad_(True)
    yc = O.net_forward(sd, xc, t, 1)
    yc.backward(gy)
    assert rel_l2(y.detach().cpu(), yc.detach()) < 1e-5
    assert rel_l2(xd.grad.cpu(), xc.grad) < 5e-5
    for name, p in net.named_parameters():
        assert rel_l2(p.grad.cpu(), sd[name].grad) < 3e-4, name


@pytest.mark.parametrize("dim", [20, 10, 8, 28, 12])
@pytest.mark.parametrize("B,H,W", [(2, 33, 45), (1, 40, 191), (3, 26, 202)])
def test_inference_odd_widths_at_dims_off_the_winograd_path(dim, B, H, W):
    """inference pads its rows to 4 floats only when every 3x3 conv of the plan runs on a kernel that
    writes the pad columns as zeros.  Dims whose channel counts send a conv to the direct implicit-GEMM kernel (dim / 2 or
    dim not a multiple of 4, 3 < C_in < 8) must keep plain rows: net.infer and a fused sampler run at W % 4 != 0 --
    including widths >= 190, where the register-window depthwise kernel would read the pads -- against the oracle."""
    from sinddm_amd.models import SinDDMNet
    net = SinDDMNet(dim=dim, multiscale=True, device=DEV).to(DEV)
    sd = closed_form_state_dict(dim)
    net.load_state_dict(sd)
    x = hash_randn((B, 3, H, W), 700 + dim + W)
    t = torch.tensor([(61 * (i + 2)) % 1000 for i in range(B)], dtype=torch.long)
    got = net.infer(x.to(DEV), t.to(DEV), 0, 2.0).cpu()
    ref = O.net_forward(sd, x, t, 2)
    assert rel_l2(got, ref) < 1e-5, (dim, W, rel_l2(got, ref))
    # right edge on its own (a wrong pad column shows up in the last few columns, diluted in the whole-image norm)
    assert rel_l2(got[..., -4:], ref[..., -4:]) < 2e-5
    # host-t path of the sampler (one row of conditioning for the batch)
    g1 = net.infer(x.to(DEV), None, 123, 1.0).cpu()
    r1 = O.net_forward(sd, x, torch.full((B,), 123, dtype=torch.long), 1)
    assert rel_l2(g1, r1) < 1e-5 and rel_l2(g1[..., -4:], r1[..., -4:]) < 2e-5


@pytest.mark.parametrize("dim,B,H,W", [(160, 1, 12, 20), (160, 2, 47, 61), (160, 4, 94, 126), (160, 16, 48, 64),
                                       (160, 16, 186, 248), (32, 3, 67, 90), (20, 2, 33, 41), (16, 5, 24, 50)])
def test_forward_and_input_gradient_bit_reproducible(dim, B, H, W):
    """Neither the forward nor the data gradients accumulate with atomics: the same call must return the same BITS.
    Guards the LDS-DMA publication rule (common.h dma_barrier): a wave reading another wave's DMA
    destination before it has landed shows up as run-to-run differences.  The shapes walk every 3x3 kernel family:
    direct conv (dims with C % 4 != 0), first-generation Winograd, conv_wino2, conv_wino3, conv_wino4."""
    net = _net(dim)
    net.bind_grads()
    x = hash_randn((B, 3, H, W), 41).to(DEV)
    gy = hash_randn((B, 3, H, W), 42).to(DEV)
    t = torch.tensor([(37 * i + 5) % 1000 for i in range(B)], device=DEV)
    outs = []
    for _ in range(3):
        net.flat_grads.zero_()
        xd = x.clone().requires_grad_(True)
        y = net(xd, t, scale=1)
        y.backward(gy)
        outs.append((y.detach().clone(), xd.grad.clone()))
    for y, gx in outs[1:]:
        assert torch.equal(y, outs[0][0])
        assert torch.equal(gx, outs[0][1])
