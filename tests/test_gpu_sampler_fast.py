"""The production sampler path: whole runs of reverse steps inside ONE library call (sinddm_sample_chain) with the
N(0,1) draws of reference SinDDM/models.py:455 generated in the step kernel (Philox4x32-10 + Box-Muller).

The reference never seeds its generator, so the noise STREAM is not contract -- its distribution is, and so is
everything else in the step.  Hence:
  * the generator: moments, tails, independence across streams, determinism, ragged sizes;
  * the fused run == the step-by-step path (sinddm_net_forward + sinddm_reverse_step per step) fed with the SAME
    numbers (sinddm_normal_fill reproduces the in-kernel stream), bit-level up to fma contraction;
  * the public API (sample / sample_via_scale) takes the fused path when no noise is injected and agrees with the
    injected-noise path wherever the reference's own noise factor is 1e-10 (every scale > 0 at omega = 0).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import max_abs, rel_l2
from sinddm_amd.configs import CONFIGS, build_diffusion

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _fill(n, seed, stream):
    from sinddm_amd import _lib
    lib = _lib.load()
    out = torch.empty(n, device=DEV)
    _lib.check(lib.sinddm_normal_fill(_lib.ptr(out), n, seed, stream, _lib.stream_ptr(DEV)), "sinddm_normal_fill")
    torch.cuda.synchronize()
    return out


def test_philox_normal_statistics():
    n = 1 << 22
    z = _fill(n, 1234, 0).double()
    assert abs(float(z.mean())) < 3e-3
    assert abs(float(z.var()) - 1.0) < 4e-3
    assert abs(float((z ** 4).mean()) - 3.0) < 0.03                      # kurtosis of N(0,1)
    assert abs(float((z ** 3).mean())) < 0.02
    assert abs(float((z.abs() > 3).double().mean()) - 0.0026998) < 3e-4  # tail mass
    assert float(z.abs().max()) < 7.0 and torch.isfinite(z).all()
    # neighbouring values / streams / seeds are uncorrelated
    z2 = _fill(n, 1234, 1).double()
    z3 = _fill(n, 1235, 0).double()
    for a, b in ((z[:-1], z[1:]), (z[0::4], z[1::4]), (z[0::4], z[2::4]), (z, z2), (z, z3)):
        assert abs(float((a * b).mean())) < 3e-3
    assert torch.equal(_fill(n, 1234, 0).double(), z)                    # counter-based: reproducible
    # ragged sizes: the tail group of < 4 values comes from the same stream
    for m in (1, 2, 3, 5, 1023):
        assert torch.equal(_fill(m, 77, 5), _fill(1024, 77, 5)[:m])


@pytest.mark.parametrize("s,ts", [(0, [99, 98, 1, 0]), (2, [40, 39, 1, 0])])
def test_sample_chain_equals_stepwise_with_same_numbers(s, ts):
    """sinddm_sample_chain == sinddm_net_forward + sinddm_reverse_step per step with z = sinddm_normal_fill(seed, i)."""
    from sinddm_amd import _lib
    from sinddm_amd.models import _workspace
    lib = _lib.load()
    net, d = build_diffusion("C1", dim=32, device=DEV)
    B = 2
    H, W = d.image_sizes[s]
    g = torch.Generator(device=DEV).manual_seed(3)
    x0 = torch.randn(B, 3, H, W, device=DEV, generator=g)
    xt = torch.randn(B, 3, H, W, device=DEV, generator=g) * 0.5
    d.img_prev_upsample = xt
    seed, n = 987654321, len(ts)
    # step by step, noise injected
    x = x0.clone()
    for i, t in enumerate(ts):
        z = _fill(x.numel(), seed, i).view_as(x)
        d.noise_fn = lambda kind, shape, ss, tt, dev, z=z: z
        x = d._p_sample_host_t(x, t, s)
    d.noise_fn = None
    # fused
    xa, xb, eps = x0.clone(), torch.empty_like(x0), torch.empty_like(x0)
    tab = d._coef_table(s)
    coefs = (_lib.StepCoefs * n)(*[tab[t] for t in ts])
    tl = (C.c_int * n)(*ts)
    ws = _workspace(DEV, lib.sinddm_workspace_bytes(32, B, H, W))
    flag = C.c_int(0)
    _lib.check(lib.sinddm_sample_chain(_lib.ptr(net.flat_params), _lib.ptr(net.packed_weights()), _lib.ptr(xa), _lib.ptr(xb),
                                       _lib.ptr(eps), _lib.ptr(xt), coefs, tl, n, float(s), seed, 0, 32, B, H, W,
                                       ws.data_ptr(), ws.numel(), _lib.stream_ptr(DEV), C.byref(flag)), "sinddm_sample_chain")
    torch.cuda.synchronize()
    y = xb if flag.value else xa
    assert flag.value == n % 2
    assert max_abs(y.cpu(), x.cpu()) <= 2e-6 * max(1.0, float(x.abs().max()))


def test_public_api_takes_fused_path_and_matches_at_fine_scales():
    """sample_via_scale without injected noise (fused run, in-kernel noise) vs the same call with zero noise injected
    at every step: at s > 0 and omega = 0 the reference's noise factor is exp(0.5*log 1e-20) = 1e-10, so
    the two paths must agree up to the amplification of that 1e-10 by the chain -- while the re-noise draw of models.py:518 is shared
    through torch's generator."""
    net, d = build_diffusion("C1", dim=32, device=DEV)
    s = 2
    prev = torch.randn(3, 3, *d.image_sizes[s - 1], device=DEV).clamp(-1, 1)
    calls = {"n": 0}
    orig = d._p_sample_host_t

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)

    d._p_sample_host_t = counting
    torch.manual_seed(5)
    fused = d.sample_via_scale(3, prev, s=s, custom_sample=True, custom_img_size_idx=s, custom_t=d.num_timesteps_ideal[s])
    assert calls["n"] == 0, "the production path must not come back to Python between steps"
    torch.manual_seed(5)
    renoise = torch.randn(fused.shape, device=DEV)                          # what _draw('renoise') consumed
    d.noise_fn = lambda kind, shape, ss, tt, dev: renoise if kind == "renoise" else torch.zeros(shape, device=dev)
    stepwise = d.sample_via_scale(3, prev, s=s, custom_sample=True, custom_img_size_idx=s, custom_t=d.num_timesteps_ideal[s])
    assert calls["n"] == d.num_timesteps_ideal[s]
    # (not bit-level: the 1e-10-scaled draws differ, and 41 chained evaluations of an untrained net amplify that)
    assert max_abs(fused.cpu(), stepwise.cpu()) < 1e-3
    # scale 0: real noise -- distribution-level check of a full T=100 chain: finite, clipped range, not degenerate
    d.noise_fn = None
    img = d.sample(batch_size=4, s=0)
    assert torch.isfinite(img).all() and float(img.abs().max()) < 5 and float(img.std()) > 1e-3
    assert not torch.equal(img[0], img[1])                                  # independent chains


@pytest.mark.parametrize("cfg,s,B", [("C2", 1, 16), ("C2", 1, 13), ("C4", 1, 16), ("C2", 2, 6), ("C3", 1, 16)])
def test_two_stream_half_batches_equal_single_stream(cfg, s, B):
    """sinddm_sample_chain2 with a second stream runs coarse-scale runs as two half-batches whose launches overlap; the
    noise is keyed on the element's index inside the WHOLE batch, so both runs consume the same draws and must agree to
    rounding (not bit for bit: a half-batch launch may take another Winograd kernel than the full batch, like any launch
    size does) -- even and odd batches, plain and padded rows, scale 0 with real noise and finer scales."""
    from sinddm_amd import _lib
    from sinddm_amd.models import _workspace
    lib = _lib.load()
    net, d = build_diffusion(cfg, dim=160, device=DEV)
    H, W = d.image_sizes[s]
    g = torch.Generator(device=DEV).manual_seed(11)
    x0 = torch.randn(B, 3, H, W, device=DEV, generator=g)
    xt = torch.randn(B, 3, H, W, device=DEV, generator=g) * 0.5
    ts = [30, 29, 28, 1, 0]
    n = len(ts)
    tab = d._coef_table(s)
    coefs = (_lib.StepCoefs * n)(*[tab[t] for t in ts])
    tl = (C.c_int * n)(*ts)
    ws = _workspace(DEV, lib.sinddm_workspace_bytes(160, B, H, W))
    aux = torch.cuda.Stream(device=DEV)
    outs = []
    for a in (None, aux.cuda_stream):
        xa, xb, eps = x0.clone(), torch.empty_like(x0), torch.empty_like(x0)
        flag = C.c_int(0)
        _lib.check(lib.sinddm_sample_chain2(_lib.ptr(net.flat_params), _lib.ptr(net.packed_weights()), _lib.ptr(xa), _lib.ptr(xb),
                                            _lib.ptr(eps), _lib.ptr(xt), coefs, tl, n, float(s), 4242, 0, 160, B, H, W,
                                            ws.data_ptr(), ws.numel(), _lib.stream_ptr(DEV), a, C.byref(flag)),
                   "sinddm_sample_chain2")
        torch.cuda.synchronize()
        outs.append((xb if flag.value else xa).clone())
    assert torch.isfinite(outs[0]).all()
    assert rel_l2(outs[1].cpu(), outs[0].cpu()) < 2e-5
    # the second half's draws are the whole batch's: with another key the halves would differ at the noise level (~1)
    assert max_abs(outs[1][B - 1:].cpu(), outs[0][B - 1:].cpu()) < 1e-3
