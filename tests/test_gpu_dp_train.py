"""GPU, world_size=2 on ONE device (gloo rendezvous, gradients staged through the host): data-parallel
MultiscaleTrainer.train() -- 2 ranks x B/2 samples with one gradient all-reduce per step -- must reproduce the
single-process run on the full batch with the same per-sample (t, noise), and keep both ranks' parameters
bit-identical.  (RCCL itself cannot be exercised with one GPU; the collective call site is the same.)"""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from PIL import Image

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
B, STEPS, DIM = 4, 3, 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, tmp, q):
    import torch.distributed as td
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sinddm_amd import dist as sd
        from sinddm_amd.models import MultiScaleGaussianDiffusion, SinDDMNet
        from sinddm_amd.synth import closed_form_state_dict, hash_randn, noise_key
        from sinddm_amd.trainer import MultiscaleTrainer
        dev = "cuda:0"
        with open(os.path.join(GOLDEN, "g11_img_scales.json")) as f:
            meta = json.load(f)["C1"]
        pyr = np.load(os.path.join(GOLDEN, "c1_pyramid.npz"))
        folder = os.path.join(tmp, f"w{world}r{rank}", "balloons") + "/"
        for key in pyr.files:
            os.makedirs(folder + key, exist_ok=True)
            Image.fromarray(pyr[key]).save(folder + key + "/balloons.png")
        net = SinDDMNet(dim=DIM, multiscale=True, device=dev).to(dev)
        net.load_state_dict(closed_form_state_dict(DIM))
        if rank == 1:                       # the trainer must replace this with rank 0's weights
            with torch.no_grad():
                net.flat_params.mul_(1.5)
            net.mark_dirty()
        sizes = [tuple(s) for s in meta["sizes"]]
        d = MultiScaleGaussianDiffusion(net, n_scales=meta["n_scales"], scale_factor=meta["scale_factor"],
                                        image_sizes=sizes, timesteps=meta["T"], train_full_t=True,
                                        scale_losses=meta["rescale_losses"], loss_factor=1, loss_type="l1",
                                        device=dev, reblurring=True, omega=0).to(dev)
        tr = MultiscaleTrainer(d, folder=folder, n_scales=meta["n_scales"], scale_factor=meta["scale_factor"],
                               image_sizes=sizes, train_batch_size=B, train_lr=1e-3, train_num_steps=STEPS,
                               gradient_accumulate_every=1, ema_decay=0.995, fp16=False, step_start_ema=1,
                               update_ema_every=1, save_and_sample_every=10 ** 9, avg_window=1,
                               sched_milestones=[100], results_folder=os.path.join(tmp, f"res{world}{rank}"), device=dev)
        assert tr.data_parallel == (world > 1) and tr.local_batch_size == B // world
        off, lb = sd.shard_offset(B), tr.local_batch_size
        s_seq = [2, 0, 1]
        tr.scale_fn = lambda step: s_seq[step]
        o_randint, o_randn_like = torch.randint, torch.randn_like
        # per-SAMPLE draws keyed by the global sample index, so a shard sees exactly its slice of the full batch
        torch.randint = lambda lo, hi, size, **kw: torch.tensor(
            [(17 * (tr.step + 1) + 29 * (off + i)) % hi for i in range(lb)], dtype=torch.long, device=dev)
        torch.randn_like = lambda x, **kw: hash_randn((B,) + tuple(x.shape[1:]),
                                                      noise_key("train", s_seq[tr.step], tr.step))[off:off + lb].to(x.device)
        try:
            tr.train()
        finally:
            torch.randint, torch.randn_like = o_randint, o_randn_like
        q.put((rank, "ok", net.flat_params.detach().cpu().numpy(), tr.ema_model.denoise_fn.flat_params.detach().cpu().numpy(),
               list(tr.running_loss)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "ERR " + repr(e) + traceback.format_exc(), None, None, None))
    finally:
        if world > 1:
            td.destroy_process_group()


def _launch(world, tmp):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, tmp, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[1]
    return res


def test_data_parallel_train_matches_single_process(tmp_path):
    single = _launch(1, str(tmp_path))[0]
    dp = _launch(2, str(tmp_path))
    # both ranks applied the same reduced gradient: parameters and EMA stay bit-identical
    assert np.array_equal(dp[0][2], dp[1][2]) and np.array_equal(dp[0][3], dp[1][3])
    # the logged loss is the global-batch mean on every rank
    assert np.allclose(dp[0][4], dp[1][4], rtol=0, atol=0)
    assert np.allclose(dp[0][4], single[4], rtol=2e-5), (dp[0][4], single[4])
    # same trajectory as one process on the full batch (summation order differs: 2e-3 relative like G10;
    # Adam's update is sign-like for tiny gradients, so compare in the lr-scaled norm)
    lr = 1e-3
    assert np.abs(dp[0][2] - single[2]).max() < 2 * lr * STEPS
    rel = np.linalg.norm(dp[0][2] - single[2]) / np.linalg.norm(single[2])
    assert rel < 2e-3, rel


# ---- the same data-parallel run against the REFERENCE's own numbers (fixture G10: 20 train() steps of the reference
# trainer at dim=32, batch 2, injected scale / t / noise) -- not against another run of this library ----
def _run_g10(rank, world, port, tmp, q):
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sinddm_amd import dist as sd
        from sinddm_amd.models import MultiScaleGaussianDiffusion, SinDDMNet
        from sinddm_amd.synth import closed_form_state_dict, hash_randn, noise_key
        from sinddm_amd.trainer import MultiscaleTrainer
        dev = "cuda:0"
        g = np.load(os.path.join(GOLDEN, "g10_train.npz"))
        with open(os.path.join(GOLDEN, "g11_img_scales.json")) as f:
            meta = json.load(f)["C1"]
        pyr = np.load(os.path.join(GOLDEN, "c1_pyramid.npz"))
        folder = os.path.join(tmp, f"g10r{rank}", "balloons") + "/"
        for key in pyr.files:
            os.makedirs(folder + key, exist_ok=True)
            Image.fromarray(pyr[key]).save(folder + key + "/balloons.png")
        net = SinDDMNet(dim=32, multiscale=True, device=dev).to(dev)
        net.load_state_dict(closed_form_state_dict(32))
        sizes = [tuple(s) for s in meta["sizes"]]
        d = MultiScaleGaussianDiffusion(net, n_scales=meta["n_scales"], scale_factor=meta["scale_factor"],
                                        image_sizes=sizes, timesteps=meta["T"], train_full_t=True,
                                        scale_losses=meta["rescale_losses"], loss_factor=1, loss_type="l1",
                                        device=dev, reblurring=True, omega=0).to(dev)
        GB = 2
        tr = MultiscaleTrainer(d, folder=folder, n_scales=meta["n_scales"], scale_factor=meta["scale_factor"],
                               image_sizes=sizes, train_batch_size=GB, train_lr=1e-3, train_num_steps=20,
                               gradient_accumulate_every=1, ema_decay=0.995, fp16=False, step_start_ema=6,
                               update_ema_every=2, save_and_sample_every=10 ** 9, avg_window=100,
                               sched_milestones=[5, 12], results_folder=os.path.join(tmp, f"g10res{rank}"), device=dev)
        assert tr.data_parallel and tr.local_batch_size == 1
        off, lb = sd.shard_offset(GB), tr.local_batch_size
        s_seq, t_seq = list(g["s_seq"]), g["t_seq"]
        tr.scale_fn = lambda step: s_seq[step]
        losses, lrs = [], []
        orig_forward = d.forward

        def rec_forward(x, s, *a, **k):
            loss = orig_forward(x, s, *a, **k)
            losses.append(float(loss.detach()))
            lrs.append(tr.opt.param_groups[0]["lr"])
            return loss

        d.forward = rec_forward
        o_randint, o_randn_like = torch.randint, torch.randn_like
        # this rank's slice of the reference's global-batch draws
        torch.randint = lambda lo, hi, size, **kw: torch.tensor(t_seq[tr.step][off:off + lb], dtype=torch.long, device=dev)
        torch.randn_like = lambda x, **kw: hash_randn((GB,) + tuple(x.shape[1:]),
                                                      noise_key("train", s_seq[tr.step], tr.step))[off:off + lb].to(x.device)
        try:
            tr.train()
        finally:
            torch.randint, torch.randn_like = o_randint, o_randn_like
        names = [n for n, _ in tr.model.denoise_fn.named_parameters()]
        pv = torch.cat([p.detach().reshape(-1) for p in tr.model.denoise_fn.parameters()]).cpu().numpy()
        ev = torch.cat([p.detach().reshape(-1) for p in tr.ema_model.denoise_fn.parameters()]).cpu().numpy()
        q.put((rank, "ok", pv, ev, losses, lrs, names))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "ERR " + repr(e) + traceback.format_exc(), None, None, None, None, None))
    finally:
        td.destroy_process_group()


def test_data_parallel_train_matches_reference_run(tmp_path):
    """2 ranks x 1 sample, one flat-gradient all-reduce per step, vs fixture G10 = what the REFERENCE trainer computes
    in one process on the batch of 2: shard-mean losses average to its loss trajectory, the LR schedule is its
    schedule, and the parameters / EMA parameters end where its run ends (same tolerances as the 1-process G10 test)."""
    from sinddm_amd.synth import closed_form_state_dict
    g = np.load(os.path.join(GOLDEN, "g10_train.npz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_g10, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[1]
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])       # ranks stay identical
    losses = (np.array(res[0][4]) + np.array(res[1][4])) / 2                                  # global-batch mean
    assert np.allclose(np.array(res[0][5]), g["lrs"], rtol=0, atol=1e-12)
    assert abs(losses[0] - g["losses"][0]) < 2e-6
    assert (np.abs(losses - g["losses"]) / g["losses"]).max() < 2e-3
    names = res[0][6]
    rp = np.concatenate([g[f"p_{n}"].reshape(-1) for n in names])
    re_ = np.concatenate([g[f"ema_{n}"].reshape(-1) for n in names])
    p0 = np.concatenate([v.reshape(-1).numpy() for v in closed_form_state_dict(32).values()])
    assert np.linalg.norm(res[0][2] - rp) / np.linalg.norm(rp - p0) < 2e-2
    assert np.linalg.norm(res[0][3] - re_) / np.linalg.norm(re_ - p0) < 2e-2
