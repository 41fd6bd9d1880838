"""The PRODUCTION sampler path (sinddm_sample_chain: fused final conv + reverse step + in-kernel noise, the call bench.py
times) tied to the oracle-verified step-by-step path at the benchmarked shapes:

  * dim = 160 (80-channel final conv), C2 finest scale at batch 16, C3 finest scale at batch 4, and C2's 133x177 scale
    (H*W % 4 != 0: the unfused final-conv + reverse-step fallback), three steps each incl. t = 0, with the SAME normal
    numbers on both paths (sinddm_normal_fill reproduces the in-kernel stream);
  * C3 at the benchmarked batch of 64 (2.15e9-element activations: 32-bit addressing at its limit): one reverse step
    through the library, samples 0 and 63 against the oracle.
reference SinDDM/models.py:449-459 (p_sample), 477-485 / 536-546 (the loops)
"""
import ctypes as C

import pytest
import torch

from conftest import max_abs, rel_l2
from oracle import sinddm_oracle as O
from sinddm_amd.configs import CONFIGS, build_diffusion
from sinddm_amd.synth import closed_form_state_dict, hash_randn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _fill(n, seed, stream):
    from sinddm_amd import _lib
    lib = _lib.load()
    out = torch.empty(n, device=DEV)
    _lib.check(lib.sinddm_normal_fill(_lib.ptr(out), n, seed, stream, _lib.stream_ptr(DEV)), "sinddm_normal_fill")
    return out


@pytest.mark.parametrize("cfg,s,B,ts", [("C2", 4, 16, [227, 100, 0]),      # 186x248, the benchmarked C2 step
                                        ("C3", 5, 4, [118, 1, 0]),         # 411x512
                                        ("C2", 3, 4, [311, 2, 0])])        # 133x177: H*W % 4 = 1
def test_sample_chain_equals_stepwise_dim160(cfg, s, B, ts):
    from sinddm_amd import _lib
    from sinddm_amd.models import _workspace
    lib = _lib.load()
    net, d = build_diffusion(cfg, dim=160, device=DEV)
    H, W = d.image_sizes[s]
    x0 = (hash_randn((B, 3, H, W), 31 + s) * 0.8).to(DEV)
    xt = (hash_randn((B, 3, H, W), 32 + s) * 0.5).clamp(-1, 1).to(DEV)
    d.img_prev_upsample = xt
    seed, n = 424242 + s, len(ts)
    x = x0.clone()
    for i, t in enumerate(ts):                                  # step by step: net forward + reverse-step kernel
        z = _fill(x.numel(), seed, i).view_as(x)
        d.noise_fn = lambda kind, shape, ss, tt, dev, z=z: z
        x = d._p_sample_host_t(x, t, s)
    d.noise_fn = None
    xa, xb, eps = x0.clone(), torch.empty_like(x0), torch.empty_like(x0)
    tab = d._coef_table(s)
    coefs = (_lib.StepCoefs * n)(*[tab[t] for t in ts])
    tl = (C.c_int * n)(*ts)
    ws = _workspace(DEV, lib.sinddm_workspace_bytes(160, B, H, W))
    flag = C.c_int(0)
    _lib.check(lib.sinddm_sample_chain(_lib.ptr(net.flat_params), _lib.ptr(net.packed_weights()), _lib.ptr(xa), _lib.ptr(xb),
                                       _lib.ptr(eps), _lib.ptr(xt), coefs, tl, n, float(s), seed, 0, 160, B, H, W,
                                       ws.data_ptr(), ws.numel(), _lib.stream_ptr(DEV), C.byref(flag)), "sinddm_sample_chain")
    torch.cuda.synchronize()
    y = xb if flag.value else xa
    assert torch.isfinite(y).all()
    assert max_abs(y.cpu(), x.cpu()) <= 4e-6 * max(1.0, float(x.abs().max()))
    # ... and the public fast path produces exactly this run
    torch.manual_seed(11)
    seed_api = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))
    torch.manual_seed(11)
    y_api = d._run_steps(x0.clone(), s, ts)
    xa2, xb2 = x0.clone(), torch.empty_like(x0)
    _lib.check(lib.sinddm_sample_chain(_lib.ptr(net.flat_params), _lib.ptr(net.packed_weights()), _lib.ptr(xa2), _lib.ptr(xb2),
                                       _lib.ptr(eps), _lib.ptr(xt), coefs, tl, n, float(s), seed_api, 0, 160, B, H, W,
                                       ws.data_ptr(), ws.numel(), _lib.stream_ptr(DEV), C.byref(flag)), "sinddm_sample_chain")
    torch.cuda.synchronize()
    assert torch.equal(y_api, xb2 if flag.value else xa2)


def test_c3_benchmarked_batch_64_vs_oracle():
    """C3 finest scale at the batch bench.py times: upsample-free entry (random x_t, x-tilde), ONE reverse step at
    t = 60 for all 64 chains through the library, the first and the last chain against the oracle."""
    cfg = CONFIGS["C3"]
    net, d = build_diffusion("C3", dim=160, device=DEV)
    n = len(cfg["sizes"])
    s = n - 1
    H, W = d.image_sizes[s]
    B = 64
    sched = O.make_schedule(cfg["T"], n, cfg["rescale_losses"], 1, train_full_t=True)
    sd = closed_form_state_dict(160)
    idx = [0, B - 1]
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, 3, H, W, generator=g) * 0.7
    xt = (torch.randn(B, 3, H, W, generator=g) * 0.5).clamp(-1, 1)
    z = torch.randn(B, 3, H, W, generator=g)
    d.img_prev_upsample = xt.to(DEV)
    d.noise_fn = lambda kind, shape, ss, tt, dev: z.to(dev)
    t = 60
    got = d._p_sample_host_t(x.to(DEV), t, s)
    assert torch.isfinite(got).all()
    ref = O.p_sample(sched, sd, x[idx], t, s, z[idx], xt[idx])
    err = rel_l2(got[idx].cpu(), ref)
    assert err < 2e-5, err
    # chains are independent: a middle chain evaluated alone gives the same result as inside the batch of 64
    m = 37
    d.img_prev_upsample = xt[m:m + 1].to(DEV)
    d.noise_fn = lambda kind, shape, ss, tt, dev: z[m:m + 1].to(dev)
    alone = d._p_sample_host_t(x[m:m + 1].to(DEV), t, s)
    assert rel_l2(got[m:m + 1].cpu(), alone.cpu()) < 5e-6
