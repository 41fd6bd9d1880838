#!/usr/bin/env python3
"""What do image widths with W % 4 != 0 cost?  Reverse steps (batch 16, dim 160) at 133x177 (the C2 scale) and at its
neighbours with aligned rows: python tools/edge_cost.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sinddm_amd.configs import build_diffusion
dev = torch.device("cuda:0")
net, d = build_diffusion("C2", 160, dev)
for (H, W) in ((133, 177), (133, 176), (133, 178), (132, 176), (136, 192)):
    x = torch.randn(16, 3, H, W, device=dev)
    d.img_prev_upsample = torch.randn(16, 3, H, W, device=dev)
    for i in range(5):
        x = d._p_sample_host_t(x, 60 + i, 3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 40
    for i in range(n):
        x = d._p_sample_host_t(x, 60 + (i % 30), 3)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{H}x{W}  W%4={W % 4}  {dt * 1e3:7.3f} ms/step   {16 * H * W / dt / 1e6:7.1f} Mpx-steps/s")
