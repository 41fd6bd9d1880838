#!/usr/bin/env python3
"""Kernel-level profile target on the PRODUCTION path: 200 reverse steps of one pyramid scale through sinddm_sample_chain.
python tools/scale_chain_profile.py <config> <scale> <batch>  -- prints wall ms per step (run under rocprofv3 for kernel stats)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sinddm_amd.configs import build_diffusion
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
net, d = build_diffusion(cfg, 160, dev)
d.two_streams = os.environ.get('SINDDM_ONE_STREAM', '0') != '1'
H, W = d.image_sizes[s]
x = torch.randn(B, 3, H, W, device=dev)
d.img_prev_upsample = torch.randn(B, 3, H, W, device=dev).clamp(-1, 1)
ts = [200 - i for i in range(200)]
x = d._run_steps(x, s, ts[:20])
torch.cuda.synchronize()
t0 = time.perf_counter()
x = d._run_steps(x, s, ts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / len(ts)
print(f"{cfg} scale {s} {H}x{W} batch {B} ({'two streams' if d.two_streams else 'one stream'}): {dt * 1e3:.4f} ms/step wall, {B * H * W / dt / 1e6:.1f} Mpx-steps/s")
