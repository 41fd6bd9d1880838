"""CPU, world_size=2 over gloo: the sample-batch sharding / all-gather path of sinddm_amd.dist and
of MultiscaleTrainer.sample_scales' collection step (the diffusion math itself needs the GPU; here the
per-rank 'sampler' is a deterministic stand-in so the collective plumbing is what is tested)."""
import os
import socket

import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, global_batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from sinddm_amd import dist as sd
    try:
        assert sd.is_dist() and sd.rank() == rank and sd.world_size() == world
        sizes = sd.shard_sizes(global_batch, world)
        lb = sd.local_batch(global_batch)
        assert lb == sizes[rank]
        start = sum(sizes[:rank])
        # "samples" that encode their global chain index
        local = torch.stack([torch.full((3, 4, 5), float(start + i)) for i in range(lb)]) if lb else torch.zeros(0, 3, 4, 5)
        full = sd.gather_batch(local, global_batch)
        assert full.shape == (global_batch, 3, 4, 5)
        assert torch.equal(full[:, 0, 0, 0], torch.arange(global_batch, dtype=torch.float32))
        assert sd.seed_for_rank(1234) == 1234 + rank
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("global_batch", [8, 5])
def test_gather_batch_world2(global_batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, global_batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
