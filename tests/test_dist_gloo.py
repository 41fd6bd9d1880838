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


def _dp_worker(rank, world, port, global_batch, q):
    """Data-parallel gradient plumbing on CPU: every rank back-propagates (b_r / B) * mean-loss of its shard,
    one SUM all-reduce -> the gradient of the global-batch mean loss (uneven shards included)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from sinddm_amd import dist as sd
    try:
        g = torch.Generator().manual_seed(7)
        x = torch.randn(global_batch, 6, generator=g)
        y = torch.randn(global_batch, 6, generator=g)
        w0 = torch.randn(6, generator=g)
        # single-process reference: mean |y - w * x| over the whole batch
        wr = w0.clone().requires_grad_(True)
        (y - wr * x).abs().mean().backward()
        off, lb = sd.shard_offset(global_batch), sd.local_batch(global_batch)
        assert off == sum(sd.shard_sizes(global_batch, world)[:rank])
        w = w0.clone().requires_grad_(True)
        loss = (y[off:off + lb] - w * x[off:off + lb]).abs().mean() * (lb / global_batch)
        loss.backward()
        sd.allreduce_sum_(w.grad)
        assert torch.allclose(w.grad, wr.grad, atol=1e-6), (w.grad, wr.grad)
        tot = sd.allreduce_sum_(loss.detach().reshape(1).clone())
        assert abs(float(tot) - float((y - w0 * x).abs().mean())) < 1e-6
        assert sd.broadcast_int(1000 + rank) == 1000          # rank 0's value everywhere
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("global_batch", [8, 5])
def test_data_parallel_gradient_world2(global_batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, global_batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
