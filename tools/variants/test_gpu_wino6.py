"""GPU parity of conv_wino6.h (F(4x4,3x3): 36 multiplies per 16 outputs) -- an experimental kernel that is NOT part of the
default build (-DSINDDM_WINO_F44_BUILD=1 adds it, behind the run-time switch sinddm_debug_set_f44; round 4 measured it
parity-green and no faster than conv_wino4: profiles/NOTES_r04.md).  With such a library these tests flip the switch, ask the library that the launch
really takes it (sinddm_debug_conv_path == 6) and compare with the oracle and with the shipped F(2x4) kernel on the
same inputs: exact tiles, tile rows cut by the bottom edge, both 80-channel blocks, the GELU epilogue, padded workspace
rows (odd image widths run through net.infer with a row pitch that is a multiple of 4) and a short fused sampler chain.
reference SinDDM/models.py:63,65 (the 3x3 convolutions), :449-459 (p_sample)
"""
import pytest
import torch

from conftest import rel_l2
from oracle import sinddm_oracle as O
from sinddm_amd.synth import closed_form_state_dict, hash_randn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def f44():
    from sinddm_amd import _lib
    lib = _lib.load()
    prev = lib.sinddm_debug_set_f44(1)
    if prev < 0:
        pytest.skip("library built without the F(4x4) kernel (tools/build_variant.sh <name> -DSINDDM_WINO_F44_BUILD=1)")
    yield lib
    lib.sinddm_debug_set_f44(prev)


def _net(dim=160):
    from sinddm_amd.models import SinDDMNet
    net = SinDDMNet(dim=dim, multiscale=True, device=DEV).to(DEV)
    net.load_state_dict(closed_form_state_dict(dim))
    return net


@pytest.mark.parametrize("B,H,W", [(24, 90, 128),      # one tile row cut after 2 rows
                                    (96, 48, 64),       # exact tiles, many samples
                                    (16, 186, 248),     # C2 finest scale: last tile column 24 wide, tile row cut after 2 rows
                                    (20, 99, 132)])     # H % 8 = 3, last tile column 4 wide
def test_forward_vs_oracle_and_f24(f44, B, H, W):
    assert f44.sinddm_debug_conv_path(160, B, H, W) == 6
    net = _net()
    sd = closed_form_state_dict(160)
    x = hash_randn((B, 3, H, W), 177 + W) * 0.9
    t = torch.tensor([(53 * (i + 3)) % 1000 for i in range(B)], dtype=torch.long)
    with torch.no_grad():
        got = net(x.to(DEV), t.to(DEV), scale=2).cpu()
    idx = [0, B - 1]
    ref = O.net_forward(sd, x[idx], t[idx], 2)
    assert rel_l2(got[idx], ref) < 1e-5
    f44.sinddm_debug_set_f44(0)
    assert f44.sinddm_debug_conv_path(160, B, H, W) == 4
    with torch.no_grad():
        old = net(x.to(DEV), t.to(DEV), scale=2).cpu()
    assert rel_l2(got, old) < 1e-5
    assert torch.isfinite(got).all()


@pytest.mark.parametrize("B,H,W", [(16, 133, 177), (24, 94, 126)])
def test_padded_rows_inference_vs_oracle(f44, B, H, W):
    """net.infer keeps its activations with the row pitch padded to 4 floats: odd widths qualify for the F(4x4) kernel too."""
    net = _net()
    sd = closed_form_state_dict(160)
    x = hash_randn((B, 3, H, W), 31 + W) * 0.9
    with torch.no_grad():
        got = net.infer(x.to(DEV).contiguous(), None, 417, 3.0).cpu()
    idx = [0, B - 1]
    ref = O.net_forward(sd, x[idx], torch.tensor([417, 417]), 3)
    assert rel_l2(got[idx], ref) < 1e-5
    f44.sinddm_debug_set_f44(0)
    with torch.no_grad():
        old = net.infer(x.to(DEV).contiguous(), None, 417, 3.0).cpu()
    assert rel_l2(got, old) < 1e-5


def test_fused_chain_f44_vs_f24(f44):
    """Forty fused reverse steps (sinddm_sample_chain, in-kernel noise) at the C2 finest scale: same seed, both kernels."""
    from sinddm_amd.configs import build_diffusion
    torch.manual_seed(3)
    net, d = build_diffusion("C2", dim=160, device=torch.device(DEV))
    s = 4
    H, W = d.target_size(s, (1, 1), True, s)
    total_t = d.num_timesteps_ideal[s]
    x_tilde = torch.randn(16, 3, H, W, device=DEV).clamp_(-1, 1)
    d.img_prev_upsample = x_tilde
    img0 = d._q_sample_impl(x_tilde, None, total_t, torch.randn_like(x_tilde))
    t_seq = [total_t - 1 - i for i in range(40)]
    outs = []
    for on in (1, 0):
        f44.sinddm_debug_set_f44(on)
        torch.manual_seed(11)
        outs.append(d._run_steps(img0.clone(), s, t_seq).cpu())
    assert torch.isfinite(outs[0]).all()
    assert rel_l2(outs[0], outs[1]) < 1e-4
