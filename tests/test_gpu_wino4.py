"""GPU parity of conv_wino4.h (the one-wave-per-SIMD F(2x4,3x3) kernel): every EDGE variant (W % 4 == 0 / == 2 / odd),
tile heights cut by the bottom image edge (H % 8 in {0, 2, 5}), both 80-channel blocks, and the GELU' data-gradient
variant, at launches for which the library picks this kernel -- asked of the library itself (sinddm_debug_conv_path:
4 = conv_wino4, 3 = conv_wino3, 2 = F(2x2) kernels), not re-derived from its threshold.

The forward is compared with the oracle on the first and the last sample (samples are independent; the batch only
has to be large enough to select the kernel).  The backward of a big batch (conv_wino4 data gradients) is compared with
the sum of the backward passes of its eighths (small launches: conv_wino3 / conv_wino2 -- different kernels, same
mathematics) and with oracle autograd.          reference SinDDM/models.py:63,65 (the 3x3 convolutions)
"""
import pytest
import torch

from conftest import rel_l2
from oracle import sinddm_oracle as O
from sinddm_amd.synth import closed_form_state_dict, hash_randn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _fp32_mfma_path():
    """Inference launches of these shapes default to conv_wh (tests/test_gpu_h2.py); conv_wino4 stays the kernel of the
    training forward / data gradients and of the switch-off path: keep it under test."""
    from sinddm_amd.models import SinDDMNet
    prev = SinDDMNet.fp32_convs
    SinDDMNet.fp32_convs = True        # (class attribute: every net these tests build launches with SINDDM_DIM_FP32_CONVS)
    yield
    SinDDMNet.fp32_convs = prev


def _path(B, H, W):
    """Kernel generation the dim -> dim 3x3 convolutions of a launch of this shape take (both forward and data gradient)."""
    from sinddm_amd import _lib
    return _lib.load().sinddm_debug_conv_path(160, B, H, W)


def _net(dim=160):
    from sinddm_amd.models import SinDDMNet
    net = SinDDMNet(dim=dim, multiscale=True, device=DEV).to(DEV)
    net.load_state_dict(closed_form_state_dict(dim))
    return net


@pytest.mark.parametrize("B,H,W", [(12, 133, 177),     # odd width (EDGE = 2), H % 8 = 5
                                    (24, 94, 126),      # W % 4 = 2 (EDGE = 1), H % 8 = 6
                                    (24, 90, 128),      # W % 4 = 0, one tile row cut after 2 rows
                                    (96, 48, 64),       # exact tiles, many samples
                                    (6, 186, 250)])     # C2-sized, W % 4 = 2, last tile column 26 wide
def test_forward_vs_oracle(B, H, W):
    assert _path(B, H, W) == 4                          # the 160-channel launches of the net take conv_wino4
    net = _net()
    sd = closed_form_state_dict(160)
    x = hash_randn((B, 3, H, W), 77 + W) * 0.9
    t = torch.tensor([(53 * (i + 3)) % 1000 for i in range(B)], dtype=torch.long)
    with torch.no_grad():
        got = net(x.to(DEV), t.to(DEV), scale=2).cpu()
    idx = [0, B - 1]
    ref = O.net_forward(sd, x[idx], t[idx], 2)
    assert rel_l2(got[idx], ref) < 1e-5
    # every sample of the batch: the same network evaluated alone (small launch -> conv_wino3 / conv_wino2)
    for i in (1, B // 2):
        yi = net.infer(x[i:i + 1].to(DEV).contiguous(), None, int(t[i]), 2.0).cpu()
        assert rel_l2(got[i:i + 1], yi) < 5e-6, i
    assert torch.isfinite(got).all()


@pytest.mark.parametrize("B,H,W", [(24, 94, 126), (16, 133, 177), (24, 96, 128)])
def test_backward_big_batch_equals_sum_of_eighths(B, H, W):
    """dgrad 3x3 convs of the big batch run on conv_wino4 (ACT = 0 and the GELU' variant ACT = 2, every EDGE variant);
    its eighths are launches the library sends to conv_wino3 / conv_wino2 -- an independent implementation."""
    assert _path(B, H, W) == 4 and _path(B // 8, H, W) in (2, 3)
    net = _net()
    net.bind_grads()
    x = hash_randn((B, 3, H, W), 5)
    gy = hash_randn((B, 3, H, W), 6) / (B * 3 * H * W)
    t = torch.tensor([(91 * (i + 1)) % 1000 for i in range(B)], dtype=torch.long)

    def run(sl):
        net.flat_grads.zero_()
        xd = x[sl].to(DEV).requires_grad_(True)
        y = net(xd, t[sl].to(DEV), scale=3)
        y.backward(gy[sl].to(DEV))
        return y.detach().cpu(), xd.grad.cpu(), net.flat_grads.clone().cpu()

    y_all, gx_all, gp_all = run(slice(0, B))
    q = B // 8
    gp_sum = torch.zeros_like(gp_all)
    for k in range(8):
        y_q, gx_q, gp_q = run(slice(k * q, (k + 1) * q))
        assert rel_l2(y_all[k * q:(k + 1) * q], y_q) < 5e-6
        assert rel_l2(gx_all[k * q:(k + 1) * q], gx_q) < 5e-5
        gp_sum += gp_q
    # per-tensor comparison (the flat buffer mixes magnitudes)
    off = 0
    for name, p in net.named_parameters():
        n = p.numel()
        a, b = gp_all[off:off + n], gp_sum[off:off + n]
        off += n
        cond_path = ".mlp." in name or "time_mlp" in name or "time_reshape" in name or name.endswith("ds_conv.bias")
        # (the condition-path / depthwise-bias gradients are sums over every pixel with heavy cancellation -- e.g. the three
        # entries of l1.time_reshape.bias are ~1e-9 -- so batch vs eighths differ there by summation order alone)
        assert rel_l2(a, b) < (2e-3 if cond_path else 3e-4), name


@pytest.mark.parametrize("B,H,W", [(24, 94, 126), (12, 133, 177)])      # EDGE = 1 and the odd-width EDGE = 2 / ACT = 2 variants
def test_backward_vs_oracle_autograd_on_conv_wino4(B, H, W):
    """One training-shaped batch on conv_wino4 against oracle autograd (input gradient of every sample is independent:
    two samples bound the CPU time; the parameter gradients are covered by the eighths test above)."""
    assert _path(B, H, W) == 4
    net = _net()
    net.bind_grads()
    net.flat_grads.zero_()
    x = hash_randn((B, 3, H, W), 25)
    gy = hash_randn((B, 3, H, W), 26) / (B * 3 * H * W)
    t = torch.tensor([(91 * (i + 1)) % 1000 for i in range(B)], dtype=torch.long)
    xd = x.to(DEV).requires_grad_(True)
    y = net(xd, t.to(DEV), scale=1)
    y.backward(gy.to(DEV))
    idx = [0, B - 1]
    sd = closed_form_state_dict(160)
    xc = x[idx].clone().requires_grad_(True)
    yc = O.net_forward(sd, xc, t[idx], 1)
    yc.backward(gy[idx])
    assert rel_l2(y.detach().cpu()[idx], yc.detach()) < 1e-5
    assert rel_l2(xd.grad.cpu()[idx], xc.grad) < 5e-5
