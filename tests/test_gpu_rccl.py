"""First execution of backend="nccl" (= RCCL) on hardware.  The GPU boxes of this build have ONE
MI355X, so the collectives of the multi-GPU path run here in a ONE-rank process group: `force=True` makes
sinddm_amd.dist issue the real RCCL calls (all_gather_into_tensor with the uneven-shard padding, all_reduce, broadcast)
on device tensors instead of returning early, and bench.py is run the way the driver's torchrun launches a rank
(RANK / WORLD_SIZE / MASTER_* in the environment) with the process group forced on.  What this does NOT show is data
moving over xGMI between ranks -- that stays unmeasured.   reference: SURVEY.md 8(e)"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, torch
import torch.distributed as td
sys.path.insert(0, %r)
from sinddm_amd import dist as sdist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
td.init_process_group(backend="nccl", device_id=dev)
assert td.get_backend() == "nccl" and sdist.world_size() == 1
x = torch.arange(5 * 3 * 7 * 9, device=dev, dtype=torch.float32).reshape(5, 3, 7, 9)
# identity without force (production single-GPU path makes no collective) ...
assert sdist.gather_batch(x, 5) is x
# ... the real all_gather_into_tensor with force, also through the padding path of an uneven shard
g = sdist.gather_batch(x, 5, force=True)
assert g.data_ptr() != x.data_ptr() and torch.equal(g, x)
g = sdist.gather_batch(x, 5, force=True, pad_to=8)
assert tuple(g.shape) == (5, 3, 7, 9) and torch.equal(g, x)
flat = torch.full((1106772,), 0.25, device=dev)            # the flat gradient buffer of dim = 160
r = sdist.allreduce_sum_(flat, force=True)
assert r is flat and float(flat.sum()) == 0.25 * 1106772
w = torch.randn(1106772, device=dev)
w0 = w.clone()
sdist.broadcast_(w, 0, force=True)
assert torch.equal(w, w0)
assert sdist.broadcast_int(1234567, force=True) == 1234567
td.barrier()
torch.cuda.synchronize()
td.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


def _env(port):
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def _port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_collectives_forced_in_one_rank_group():
    p = subprocess.run([sys.executable, "-c", WORKER % ROOT], env=_env(_port()), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])


def test_bench_torchrun_branch_with_rccl_group():
    """bench.py as the driver's torchrun launches it (env rendezvous), N = 1, process group forced on: init_process_group
    ("nccl"), the barriers, the max/min-over-ranks all-reduces and the all-gather of the full-sample leg all execute."""
    env = _env(_port())
    env["SINDDM_BENCH_FORCE_DIST"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "C1", "--batch", "2",
                        "--steps", "3", "--warmup", "1", "--no-cpu", "--no-train", "--no-strong", "--no-c2"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["comm_world_size"] == 1 and line["value"] > 0
    assert line["full_sample"]["finite"] and line["full_sample"]["all_gather_seconds"] > 0
