"""GPU, world_size=2 on ONE device (gloo rendezvous; the all-gather is staged through the host by sinddm_amd.dist):
MultiscaleTrainer.sample_scales shards the sample batch over ranks as independent chains.  Parity
definition of 8(e): rank r's shard of the gathered batch must equal a SINGLE-process run with batch B/R and rank r's
noise -- bit for bit, since the launches are identical.  (RCCL itself needs >= 2 devices; the collective call site is
the same, and the driver's 8-GPU tier exercises it.)"""
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
GB, DIM = 4, 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, tmp, off_single, lb_single, q):
    """world == 2: a rank of the sharded run.  world == 1: the single-process reference for chains
    [off_single, off_single + lb_single)."""
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
        folder = os.path.join(tmp, f"w{world}r{rank}o{off_single}", "balloons") + "/"
        for key in pyr.files:
            os.makedirs(folder + key, exist_ok=True)
            Image.fromarray(pyr[key]).save(folder + key + "/balloons.png")
        net = SinDDMNet(dim=DIM, multiscale=True, device=dev).to(dev)
        net.load_state_dict(closed_form_state_dict(DIM))
        sizes = [tuple(s) for s in meta["sizes"]]
        d = MultiScaleGaussianDiffusion(net, n_scales=meta["n_scales"], scale_factor=meta["scale_factor"],
                                        image_sizes=sizes, timesteps=20, train_full_t=True,
                                        scale_losses=meta["rescale_losses"], loss_factor=1, loss_type="l1",
                                        device=dev, reblurring=True, omega=0).to(dev)
        tr = MultiscaleTrainer(d, folder=folder, n_scales=meta["n_scales"], scale_factor=meta["scale_factor"],
                               image_sizes=sizes, train_batch_size=2, train_num_steps=1,
                               results_folder=os.path.join(tmp, f"res{world}{rank}{off_single}"), device=dev)
        if world > 1:
            off, lb, batch = sd.shard_offset(GB), sd.local_batch(GB), GB
        else:
            off, lb, batch = off_single, lb_single, lb_single

        # chain j of the GLOBAL batch has its own noise stream, whatever rank runs it
        def noise(kind, shape, s, t, device):
            assert shape[0] == lb
            return torch.stack([hash_randn(tuple(shape[1:]), noise_key(kind, s, t) + 7919 * (off + i))
                                for i in range(lb)]).to(device)

        tr.ema_model.noise_fn = noise
        outs = tr.sample_scales(batch_size=batch, custom_sample=True, save_images=False)
        q.put((rank, "ok", [o.cpu().numpy() for o in outs]))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "ERR " + repr(e) + traceback.format_exc(), None))
    finally:
        if world > 1:
            td.destroy_process_group()


def _launch(world, tmp, off=0, lb=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, tmp, off, lb, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[1]
    return res


def test_sharded_sample_scales_equals_single_process_shards(tmp_path):
    sharded = _launch(2, str(tmp_path))
    # every rank holds the same gathered batches
    for a, b in zip(sharded[0][2], sharded[1][2]):
        assert a.shape[0] == GB and np.array_equal(a, b)
    for r, (off, lb) in enumerate(((0, 2), (2, 2))):
        single = _launch(1, str(tmp_path), off, lb)[0][2]
        for s, (got, ref) in enumerate(zip(sharded[0][2], single)):
            assert np.array_equal(got[off:off + lb], ref), (r, s, np.abs(got[off:off + lb] - ref).max())
