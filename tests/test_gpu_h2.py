"""GPU parity of the binary16 hi/lo 3x3 kernel of big launches:
  * conv_wh.h -- Winograd F(2x4,3x3) whose 24 frequency GEMMs run on the binary16 matrix pipe with the transformed input and the
    transformed weights each split into two binary16 pieces (all four MFMA terms, fp32 accumulate, exact power-of-two operand
    scales from the weights' per-channel max and the activation tensor's per-sample running max) (sinddm_debug_infer_path = 8)
(the direct implicit-GEMM sibling of round 5, conv_h2.h, is archived in tools/variants/.)

The gate: the kernel must not be narrower than fp32.  Every evaluation is compared with the FLOAT64
oracle and its error is held against the error of fp32 arithmetic on the same inputs -- the fp32 oracle (torch CPU fp32 = the
reference's own arithmetic) and the library's own fp32-MFMA Winograd path (SinDDMNet.fp32_convs = the per-call option
SINDDM_DIM_FP32_CONVS): <= 1.5 x.  Also: ragged widths (padded rows), tile rows cut by the image edge, dynamic-range stress
(activations scaled by 2^-12 and 2^+10, channel gains 2^+-8, one loud sample, weights after optimiser steps), dim = 80 / 240
(a binary16 conv behind an fp32 one), a fused sampler chain.
reference SinDDM/models.py:63,65 (the 3x3 convolutions), :69-80 (the block)
"""
import pytest
import torch

from conftest import rel_l2
from oracle import sinddm_oracle as O
from sinddm_amd.synth import closed_form_state_dict, hash_randn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lib():
    from sinddm_amd import _lib
    return _lib.load()


def _net(dim=160, sd=None):
    from sinddm_amd.models import SinDDMNet
    net = SinDDMNet(dim=dim, multiscale=True, device=DEV).to(DEV)
    net.load_state_dict(sd if sd is not None else closed_form_state_dict(dim))
    return net


def _net_forward_f64(sd, x, t, scale):
    """The oracle's network in float64 (the conditioning vector comes from the fp32 oracle: it is not what is tested)."""
    cond = O.cond_vector(sd, t, scale).double()
    sd64 = {k: v.double() for k, v in sd.items()}
    h = x.double()
    for name in ("l1", "l2", "l3", "l4"):
        h = O.conv_block(sd64, name, h, cond)
    return torch.nn.functional.conv2d(h, sd64["final_conv.0.weight"], sd64["final_conv.0.bias"])


@pytest.fixture
def h2_switch():
    lib = _lib()
    lib.path = 8
    return lib


# (batches: conv_wh takes a launch from 12 items of 8x32 pixels x 80 channels per CU)
@pytest.mark.parametrize("B,H,W", [(16, 186, 248),     # C2 finest at its benchmarked batch: W % 4 == 0, H % 8 = 2
                                    (28, 133, 177),     # odd width: rows padded to 180, last item 20 columns wide
                                    (4, 411, 512),      # C3 finest, exact items, H % 8 = 3
                                    (56, 94, 126)])     # W % 4 = 2
def test_error_vs_float64_not_wider_than_fp32(h2_switch, B, H, W):
    lib = h2_switch
    assert lib.sinddm_debug_infer_path(160, B, H, W) == lib.path
    sd = closed_form_state_dict(160)
    net = _net(160, sd)
    x = hash_randn((B, 3, H, W), 1234 + W) * 0.9
    t = torch.tensor([(53 * (i + 3)) % 1000 for i in range(B)], dtype=torch.long)
    idx = [0, B - 1]
    got = net.infer(x.to(DEV), t.to(DEV), 0, 2.0).cpu()
    ref64 = _net_forward_f64(sd, x[idx], t[idx], 2)
    ref32 = O.net_forward(sd, x[idx], t[idx], 2)
    e_h2 = rel_l2(got[idx], ref64)
    e_32 = rel_l2(ref32, ref64)
    from sinddm_amd._lib import DIM_FP32_CONVS
    assert lib.sinddm_debug_infer_path(160 | DIM_FP32_CONVS, B, H, W) != 8
    net.fp32_convs = True
    got_w = net.infer(x.to(DEV), t.to(DEV), 0, 2.0).cpu()
    net.fp32_convs = False
    e_w = rel_l2(got_w[idx], ref64)
    print(f"[conv_wh] {B}x{H}x{W}: vs float64  conv_wh {e_h2:.3e}  fp32-MFMA Winograd {e_w:.3e}  fp32 oracle {e_32:.3e};"
          f"  conv_wh vs fp32 Winograd {rel_l2(got, got_w):.3e}")
    assert rel_l2(got[idx], ref32) < 1e-5                  # the tolerance every net-forward parity test uses
    assert e_h2 <= 1.5 * e_32, (e_h2, e_32)
    assert rel_l2(got, got_w) < 5e-6
    assert torch.isfinite(got).all()


@pytest.mark.parametrize("gain", [2.0 ** -12, 2.0 ** 10])
def test_dynamic_range(h2_switch, gain):
    """Activations far from 1: the first block's output is scaled by `gain` (weights of l1.net.2 and its residual projection),
    so every later 3x3 conv sees inputs around `gain`.  The running-max scale must keep both binary16 pieces in range:
    same relative error as at gain 1."""
    lib = h2_switch
    B, H, W = 16, 186, 248
    sd = {k: v.clone() for k, v in closed_form_state_dict(160).items()}
    for k in ("l1.net.2.weight", "l1.net.2.bias", "l1.res_conv.weight", "l1.res_conv.bias"):
        sd[k] = sd[k] * gain
    for k in ("l2.ds_conv.bias", "l2.time_reshape.weight", "l2.time_reshape.bias"):
        sd[k] = sd[k] * gain
    net = _net(160, sd)
    x = hash_randn((B, 3, H, W), 99) * 0.9
    t = torch.tensor([(53 * (i + 3)) % 1000 for i in range(B)], dtype=torch.long)
    idx = [0, B - 1]
    got = net.infer(x.to(DEV), t.to(DEV), 0, 1.0).cpu()
    ref64 = _net_forward_f64(sd, x[idx], t[idx], 1)
    ref32 = O.net_forward(sd, x[idx], t[idx], 1)
    e_h2, e_32 = rel_l2(got[idx], ref64), rel_l2(ref32, ref64)
    print(f"[conv_wh] gain {gain:g}: vs float64  conv_wh {e_h2:.3e}  fp32 oracle {e_32:.3e}")
    assert torch.isfinite(got).all()
    assert e_h2 <= 1.5 * e_32, (e_h2, e_32)


def test_fused_chain_h2_vs_winograd(h2_switch):
    """The production sampler call (sinddm_sample_chain: in-kernel Philox, fused tail) for 12 steps at the C2 finest scale,
    batch 16, with the kernel on and off: same seed -> same noise, so the two runs differ by the convs' rounding only."""
    lib = h2_switch
    from sinddm_amd.configs import build_diffusion
    net, d = build_diffusion("C2", dim=160, device=torch.device(DEV))
    s = 4
    H, W = d.target_size(s, (1, 1), True, s)
    B = 16
    assert lib.sinddm_debug_infer_path(160, B, H, W) == lib.path
    x0 = hash_randn((B, 3, H, W), 5150).to(DEV)
    d.img_prev_upsample = (hash_randn((B, 3, H, W), 5151) * 0.5).clamp(-1, 1).to(DEV)
    outs = []
    for fp32 in (False, True):
        net.fp32_convs = fp32
        torch.manual_seed(7)
        outs.append(d._run_steps(x0.clone(), s, list(range(40, 28, -1))).cpu())
    net.fp32_convs = False
    err = rel_l2(outs[0], outs[1])
    print(f"12 fused steps, conv_wh vs fp32-MFMA Winograd: {err:.3e}")
    assert err < 2e-5


# ---- round 6: the gate on non-benign weights and batches (VERDICT r5 "what's weak" 1c) --------------------------------
def _gate(net, sd, x, t, scale, idx, label, factor=1.5, tol32=1e-5):
    """conv_wh's error against float64 <= factor x the error of fp32 arithmetic on the same inputs: the larger of the fp32
    oracle's (direct convolution, torch CPU) and the library's fp32-MFMA Winograd path's (F(2x4) in fp32 -- what rounds 3-4
    shipped; a Winograd transform in fp32 has its own, weight-dependent amplification that the direct form does not)."""
    got = net.infer(x.to(DEV), t.to(DEV), 0, float(scale)).cpu()
    net.fp32_convs = True
    got_w = net.infer(x.to(DEV), t.to(DEV), 0, float(scale)).cpu()
    net.fp32_convs = False
    assert torch.isfinite(got).all(), label
    worst = 0.0
    for i in idx:
        ref64 = _net_forward_f64(sd, x[i:i + 1], t[i:i + 1], scale)
        ref32 = O.net_forward(sd, x[i:i + 1], t[i:i + 1], scale)
        e_k, e_w, e_32 = rel_l2(got[i:i + 1], ref64), rel_l2(got_w[i:i + 1], ref64), rel_l2(ref32, ref64)
        print(f"[{label}] sample {i}: vs float64  conv_wh {e_k:.3e}  fp32-MFMA Winograd {e_w:.3e}  fp32 oracle {e_32:.3e}")
        assert e_k <= factor * max(e_32, e_w), (label, i, e_k, e_w, e_32)
        assert rel_l2(got[i:i + 1], ref32) < tol32, label          # (1e-5: the tolerance every net-forward parity test uses)
        worst = max(worst, e_k / max(e_32, e_w))
    return worst


def _heavy_tailed_state_dict(log2_span, seed=5):
    """Closed-form weights with per-channel gains 2^U(-span, +span) on the hidden tensors of blocks l2 and l3 -- g (conv1's
    output / conv2's input) and the block output -- compensated on the consumer side so the network stays finite: the per-
    sample activation scale (one running max for all channels of a tensor) and the per-output-channel weight scale of the
    binary16 kernels then see channels 2^(2 span) apart."""
    sd = {k: v.clone() for k, v in closed_form_state_dict(160).items()}
    gen = torch.Generator().manual_seed(seed)

    def gains(n):
        return torch.exp2((torch.rand(n, generator=gen) * 2 - 1) * log2_span)

    for blk in ("l2", "l3"):
        a = gains(sd[f"{blk}.net.0.weight"].shape[0])                 # conv1 output channels: g = GELU(a u)
        sd[f"{blk}.net.0.weight"] *= a.view(-1, 1, 1, 1)
        sd[f"{blk}.net.0.bias"] *= a
        sd[f"{blk}.net.2.weight"] /= a.view(1, -1, 1, 1)              # conv2 input channels
    # block l2's output channels (conv2 + residual projection) x b, undone by l3's depthwise conv and its residual identity:
    # l3 reads x through ds_conv (per channel: / b) -- its identity residual keeps the gain, so l3's conv2 output is scaled too
    b = gains(160)
    for k in ("l2.net.2.weight", "l2.res_conv.weight"):
        sd[k] *= b.view(-1, 1, 1, 1)
    for k in ("l2.net.2.bias", "l2.res_conv.bias"):
        sd[k] *= b
    sd["l3.ds_conv.weight"] /= b.view(-1, 1, 1, 1)
    sd["l3.net.2.weight"] *= b.view(-1, 1, 1, 1)                      # l3 out = conv2 + x: both carry b
    sd["l3.net.2.bias"] *= b
    sd["l4.ds_conv.weight"] /= b.view(-1, 1, 1, 1)
    sd["l4.res_conv.weight"] /= b.view(1, -1, 1, 1)
    return sd


@pytest.mark.parametrize("span", [4, 8])
def test_gate_heavy_tailed_channel_gains(span):
    """Per-channel gains 2^U(-span, span) on conv inputs AND outputs of the dim -> dim blocks."""
    lib = _lib()
    B, H, W = 16, 186, 248
    assert lib.sinddm_debug_infer_path(160, B, H, W) == 8
    sd = _heavy_tailed_state_dict(span)
    net = _net(160, sd)
    x = hash_randn((B, 3, H, W), 777) * 0.9
    t = torch.tensor([(53 * (i + 3)) % 1000 for i in range(B)], dtype=torch.long)
    _gate(net, sd, x, t, 2, [0, B - 1], f"channel gains 2^+-{span}")


def test_gate_one_sample_2e12_louder_than_the_batch():
    """One chain of the batch at 2^12 times the others' amplitude: the activation scales are per SAMPLE, so neither the
    loud chain nor its neighbours may lose bits."""
    lib = _lib()
    B, H, W = 16, 186, 248
    assert lib.sinddm_debug_infer_path(160, B, H, W) == 8
    sd = closed_form_state_dict(160)
    net = _net(160, sd)
    x = hash_randn((B, 3, H, W), 778) * 0.9
    x[5] *= 4096.0
    t = torch.tensor([(53 * (i + 3)) % 1000 for i in range(B)], dtype=torch.long)
    _gate(net, sd, x, t, 1, [4, 5, 6], "sample 5 x 2^12")


def test_gate_after_training_steps_dim160():
    """Weights that an optimiser has touched: 12 Adam steps (lr 1e-3: every weight moves by ~1e-2, the size of a 160 -> 160
    weight itself) of the library's own training step at dim = 160 on the GPU, then frozen and held to the same gate."""
    from sinddm_amd.configs import build_diffusion
    lib = _lib()
    net, d = build_diffusion("C2", dim=160, device=torch.device(DEV))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    s = 4
    H, W = d.image_sizes[s]
    Bt = 8
    x0 = (hash_randn((Bt, 3, H, W), 4001) * 0.5).clamp(-1, 1).to(DEV)
    xr = (hash_randn((Bt, 3, H, W), 4002) * 0.5).clamp(-1, 1).to(DEV)
    torch.manual_seed(3)
    for _ in range(12):
        opt.zero_grad()
        loss = d((x0, xr), s)
        loss.backward()
        opt.step()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    moved = float((sd["l3.net.0.weight"] - closed_form_state_dict(160)["l3.net.0.weight"]).abs().mean())
    assert moved > 2e-3, moved
    B = 16
    assert lib.sinddm_debug_infer_path(160, B, H, W) == 8
    x = hash_randn((B, 3, H, W), 779) * 0.9
    t = torch.tensor([(53 * (i + 3)) % 1000 for i in range(B)], dtype=torch.long)
    _gate(net, sd, x, t, float(s), [0, B - 1], "after 12 Adam steps")


@pytest.mark.parametrize("dim", [80, 240])
def test_conv2_on_binary16_behind_an_fp32_conv1(dim):
    """dim = 80 / 240: block l2 has C_in = dim / 2 (40 / 120: not a multiple of 16), so its conv1 stays on an fp32 kernel
    that does not publish the running max of g while conv2 (dim -> dim) qualifies for conv_wh: the library
    then measures g in one pass.  Inputs scaled so that |g| * 20 would overflow binary16 with a scale of 1."""
    lib = _lib()
    B, H, W = 48, 186, 248
    assert lib.sinddm_debug_infer_path(dim, B, H, W) == 8
    sd = {k: v.clone() for k, v in closed_form_state_dict(dim).items()}
    for k in ("l1.net.2.weight", "l1.net.2.bias", "l1.res_conv.weight", "l1.res_conv.bias"):
        sd[k] = sd[k] * 4096.0                                       # every later tensor ~ 4 000
    for k in ("l2.ds_conv.bias", "l2.time_reshape.weight", "l2.time_reshape.bias"):
        sd[k] = sd[k] * 4096.0
    net = _net(dim, sd)
    x = hash_randn((B, 3, H, W), 780) * 0.9
    t = torch.tensor([(53 * (i + 3)) % 1000 for i in range(B)], dtype=torch.long)
    # (activations of ~4 000 put every fp32 evaluation 1e-5 from float64 -- the depthwise conv and the 1x1 projections cancel large
    # terms --, so two fp32 results may differ by 2e-5: the gate is the ratio)
    _gate(net, sd, x, t, 1, [0, B - 1], f"dim {dim}", tol32=4e-5)
