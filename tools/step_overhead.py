#!/usr/bin/env python3
"""Host-side cost of one sampler step at the coarse scales: wall time per step when launches are queued
back-to-back (GPU-bound or CPU-bound, whichever is slower) vs the pure host enqueue time (no sync)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sinddm_amd.configs import CONFIGS, build_diffusion
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
net, d = build_diffusion("C2", 160, dev)
for s in range(5):
    H, W = d.image_sizes[s]
    x = torch.randn(B, 3, H, W, device=dev)
    d.img_prev_upsample = torch.randn(B, 3, H, W, device=dev)
    for _ in range(5):
        x = d._p_sample_host_t(x, 50, s)
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for i in range(n):
        x = d._p_sample_host_t(x, 60 + (i % 30), s)
    t_host = (time.perf_counter() - t0) / n          # enqueue only (may already be throttled by the queue depth)
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / n
    # GPU time of the same steps from events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        x = d._p_sample_host_t(x, 60 + (i % 30), s)
    e1.record(); torch.cuda.synchronize()
    print(f"s={s} {H}x{W}: wall {t_wall*1e3:.3f} ms/step, host enqueue {t_host*1e3:.3f} ms/step, gpu span {e0.elapsed_time(e1)/n:.3f} ms/step")
