"""GPU parity at the sizes / timesteps / chain lengths the BASELINE configs actually run.

  * the embedding + conditioning path for EVERY t of the T=1000 schedule (fixture G2, g3:cond_vec_s3) through the
    C ABI (sinddm_cond_embed)                                              reference SinDDM/models.py:39-46,136-141
  * net forward at t in {100, 500, 999} vs the oracle
  * C3 / C4 / C5 finest-scale shapes at their per-GPU batch: upsample -> q_sample -> reverse steps on the HIP path for
    the whole batch, compared with the oracle on the first and the last sample (samples are independent, so two
    samples bound the oracle's CPU time without shrinking the GPU workload)   reference SinDDM/models.py:501-568
  * the full C2 chain: 5 scales, T=1000, B=1, dim=160 = 2 478 chained network evaluations against the images the
    REFERENCE produced for the same hash noise (fixture G14), at the north_star's 1e-4 rel-L2.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import max_abs, rel_l2
from oracle import sinddm_oracle as O
from sinddm_amd.configs import CONFIGS, build_diffusion
from sinddm_amd.synth import closed_form_state_dict, hash_randn, noise_key

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net(dim):
    from sinddm_amd.models import SinDDMNet
    net = SinDDMNet(dim=dim, multiscale=True, device=DEV).to(DEV)
    net.load_state_dict(closed_form_state_dict(dim))
    return net


def _cond_embed(net, t_dev, t_host, scale, B):
    from sinddm_amd import _lib
    lib = _lib.load()
    stride = lib.sinddm_cond_stride(net.dim)
    emb = torch.empty(B, 64, device=DEV)
    cv = torch.empty(B, 32, device=DEV)
    bias = torch.empty(B, stride, device=DEV)
    _lib.check(lib.sinddm_cond_embed(_lib.ptr(net.flat_params), _lib.ptr(t_dev) if t_dev is not None else None,
                                     int(t_host), float(scale), net.dim, B, _lib.ptr(emb), _lib.ptr(cv),
                                     _lib.ptr(bias), _lib.stream_ptr(torch.device(DEV))), "sinddm_cond_embed")
    torch.cuda.synchronize()
    return emb.cpu(), cv.cpu(), bias.cpu()


def test_posemb_every_timestep_golden(golden):
    """G2: SinusoidalPosEmb(32) for t = 0..999 (arguments up to 999 rad) and s = 0..5, bit-level: <= 2 ulp of 1."""
    g = golden("g2_posemb.npz")
    net = _net(32)
    t = torch.arange(1000, device=DEV, dtype=torch.long)
    for s in range(6):
        emb, _, _ = _cond_embed(net, t, 0, float(s), 1000)
        assert max_abs(emb[:, :32], g["t_emb"]) <= 2.4e-7, s
        assert max_abs(emb[:, 32:], np.broadcast_to(g["s_emb"][s], (1000, 32))) <= 2.4e-7, s
    # host-t path (what the sampler uses): same values
    for th in (0, 100, 500, 999):
        emb, _, _ = _cond_embed(net, None, th, 2.0, 2)
        assert max_abs(emb[:, :32], np.broadcast_to(g["t_emb"][th], (2, 32))) <= 2.4e-7, th


def test_cond_vector_golden_and_oracle(golden):
    """g3:cond_vec_s3 (t = 0, 1, 17, 99, 999 at s = 3) from the reference, and every block's per-sample bias vs the
    oracle for t up to 999."""
    g = golden("g3_net.npz")
    net = _net(160)
    sd = closed_form_state_dict(160)
    t = torch.tensor([0, 1, 17, 99, 999], dtype=torch.long)
    _, cv, _ = _cond_embed(net, t.to(DEV), 0, 3.0, 5)
    assert rel_l2(cv, g["cond_vec_s3"]) < 2e-6
    t = torch.tensor([100, 250, 500, 750, 999, 0], dtype=torch.long)
    for s in (0, 4):
        _, cv, bias = _cond_embed(net, t.to(DEV), 0, float(s), len(t))
        cref = O.cond_vector(sd, t, s)
        assert rel_l2(cv, cref) < 2e-6
        off = 0
        for name, (cin, _) in zip(("l1", "l2", "l3", "l4"), O.block_channels(160)):
            ref = O.block_condition(sd, name, cref)
            assert rel_l2(bias[:, off:off + cin], ref) < 5e-6, (s, name)
            off += cin


@pytest.mark.parametrize("H,W", [(40, 70), (67, 90)])
def test_net_forward_large_t_vs_oracle(H, W):
    """Network forward at the timesteps a T=1000 chain visits (the r1 tests stopped at t = 99)."""
    net = _net(160)
    sd = closed_form_state_dict(160)
    x = hash_randn((4, 3, H, W), 300 + H)
    t = torch.tensor([100, 500, 999, 731])
    for s in (0, 3):
        ref = O.net_forward(sd, x, t, s)
        with torch.no_grad():
            y = net(x.to(DEV), t.to(DEV), scale=s)
        assert rel_l2(y.cpu(), ref) < 1e-5, s
        for i in (0, 2):
            yi = net.infer(x[i:i + 1].to(DEV).contiguous(), None, int(t[i]), float(s))
            assert rel_l2(yi.cpu(), ref[i:i + 1]) < 1e-5, (s, i)


def _scale_entry_and_steps(cfg_name, B, steps_t, check=(0, -1)):
    """Arrive at the finest scale of `cfg_name` the way sample_via_scale does (models.py:549-568): bilinear upsample
    of the previous scale's batch, q_sample at total_t, then reverse steps at `steps_t` -- whole batch on the HIP
    path, samples `check` against the oracle."""
    cfg = CONFIGS[cfg_name]
    net, d = build_diffusion(cfg_name, dim=160, device=torch.device(DEV))
    n = len(cfg["sizes"])
    s = n - 1
    mul = cfg.get("scale_mul", (1, 1))
    sched = O.make_schedule(cfg["T"], n, cfg["rescale_losses"], 1, train_full_t=True)
    sd = closed_form_state_dict(160)
    h, w = d.target_size(s - 1, mul, True, s - 1)
    H, W = d.target_size(s, mul, True, s)
    total_t = d.num_timesteps_ideal[s]
    prev = (hash_randn((B, 3, h, w), 901) * 0.6).clamp(-1, 1)
    nz0 = hash_randn((B, 3, H, W), 902)
    up = d.upsample(prev.to(DEV), (H, W))
    x = d._q_sample_impl(up, None, total_t, nz0.to(DEV))
    d.img_prev_upsample = up
    idx = [i % B for i in check]
    up_ref = O.bilinear_upsample(prev[idx], (H, W))
    assert max_abs(up[idx].cpu(), up_ref) < 2e-6
    x_ref = O.q_sample(sched, up_ref, torch.full((len(idx),), total_t, dtype=torch.long), nz0[idx])
    assert rel_l2(x[idx].cpu(), x_ref) < 1e-6
    for j, t in enumerate(steps_t):
        z = hash_randn((B, 3, H, W), 910 + j)
        d.noise_fn = lambda kind, shape, ss, tt, dev, z=z: z.to(dev)
        x = d._p_sample_host_t(x, int(t), s)
        x_ref = O.p_sample(sched, sd, x_ref, int(t), s, z[idx], up_ref)
        err = rel_l2(x[idx].cpu(), x_ref)
        assert err < 2e-5, (cfg_name, j, t, err)
    assert torch.isfinite(x).all()
    return (H, W), total_t


def test_c4_finest_scale_batch16_vs_oracle():
    """C4 starry_night 6-scale: finest 198x252 at the per-GPU batch of 16 (128 over 8 GPUs)."""
    hw, total_t = _scale_entry_and_steps("C4", 16, [196, 195, 0])
    assert hw == (198, 252) and total_t == 197


def test_c5_scale_mul_2x4_batch4_vs_oracle():
    """C5 marinabaysands --scale_mul 2 4: 258x776 -> 364x1092 at the per-GPU batch of 4."""
    hw, total_t = _scale_entry_and_steps("C5", 4, [228, 1], check=(0, -1))
    assert hw == (364, 1092) and total_t == 229


def test_c4_full_global_batch_128_vs_oracle():
    """The shape bench.py's `c4_strong` leg times at N = 1: the WHOLE global batch of 128 chains on one GPU (VERDICT r4 item 6:
    the full-batch legs were covered by isfinite only).  Oracle on the first and the last chain."""
    hw, total_t = _scale_entry_and_steps("C4", 128, [196], check=(0, -1))
    assert hw == (198, 252) and total_t == 197


def test_c5_full_global_batch_32_vs_oracle():
    """... and `c5_strong` at N = 1: 32 chains at 364x1092."""
    hw, total_t = _scale_entry_and_steps("C5", 32, [228], check=(0, -1))
    assert hw == (364, 1092) and total_t == 229


def test_c3_finest_scale_vs_oracle():
    """C3 seascape: 270x336 -> 411x512; batch 8 on the GPU, oracle on two samples."""
    hw, total_t = _scale_entry_and_steps("C3", 8, [118, 0], check=(0, -1))
    assert hw == (411, 512) and total_t == 119


def test_c2_benchmarked_batch_vs_oracle():
    """C2 finest 186x248 at the benchmarked batch of 16."""
    hw, total_t = _scale_entry_and_steps("C2", 16, [227, 100], check=(0, -1))
    assert hw == (186, 248) and total_t == 228


@pytest.mark.parametrize("B,H,W", [(1, 48, 64), (1, 94, 126), (3, 48, 64), (16, 48, 64)])
def test_small_launch_geometries_dim160_vs_oracle(B, H, W):
    """Coarse scales at small batch take different launch geometries of the same kernels: one m-tile per Winograd work
    item, the 1x1 projections with their output channels split over workgroups (16 or 80 per workgroup), the first conv
    with its channel walk split over blockIdx.z.  One network evaluation per geometry against the oracle."""
    net, d = build_diffusion("C2", dim=160, device=torch.device(DEV))
    sd = closed_form_state_dict(160)
    x = hash_randn((B, 3, H, W), 4242) * 0.8
    t = torch.tensor([(37 * (i + 1)) % 1000 for i in range(B)], dtype=torch.long)
    got = net.infer(x.to(DEV), t.to(DEV), 0, 1.0).cpu()
    idx = sorted({0, B - 1})
    ref = O.net_forward(sd, x[idx], t[idx], 1.0)
    err = rel_l2(got[idx], ref)
    assert err < 1e-5, (B, H, W, err)


def test_full_chain_c2_t1000_golden(golden):
    """G14: the headline chain length.  5 scales, T=1000, B=1, dim=160: 2 478 chained network evaluations through
    the public sample()/sample_via_scale() API with hash noise vs the REFERENCE's images (north_star: 1e-4 rel-L2).
    Also per scale in isolation: restarted from the reference's own previous-scale image."""
    g = golden("g14_chain_c2.npz")
    net, d = build_diffusion("C2", dim=160, device=torch.device(DEV))
    n = len(CONFIGS["C2"]["sizes"])
    assert d.num_timesteps_ideal == list(g["ideal"])
    assert int(g["plan_len"]) == 1 + sum(d.num_timesteps_ideal) + (n - 1)
    d.noise_fn = lambda kind, shape, s, t, dev: hash_randn(shape, noise_key(kind, s, t)).to(dev)
    outs = [d.sample(batch_size=1, s=0)]
    for s in range(1, n):
        outs.append(d.sample_via_scale(1, outs[-1], s=s, scale_mul=(1, 1), custom_sample=True,
                                       custom_img_size_idx=s, custom_t=d.num_timesteps_ideal[1:][s - 1]))
    errs = [rel_l2(o.cpu(), g[f"out_s{i}"]) for i, o in enumerate(outs)]
    print("C2 chain rel-L2 per scale (cumulative):", ["%.2e" % e for e in errs])
    iso = []
    for s in range(1, n):
        prev = torch.from_numpy(g[f"out_s{s - 1}"]).to(DEV)
        o = d.sample_via_scale(1, prev, s=s, scale_mul=(1, 1), custom_sample=True, custom_img_size_idx=s,
                               custom_t=d.num_timesteps_ideal[1:][s - 1])
        iso.append(rel_l2(o.cpu(), g[f"out_s{s}"]))
    print("C2 chain rel-L2 per scale (restarted from the reference's previous scale):", ["%.2e" % e for e in iso])
    assert max(errs) < 1e-4, errs
    assert max(iso) < 1e-4, iso
