#!/usr/bin/env python3
"""One scale of a C2 sample alone (for rocprofv3): python tools/scale0_only.py [scale] [batch]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sinddm_amd.configs import CONFIGS, build_diffusion
s = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
net, d = build_diffusion("C2", 160, dev)
cfg = CONFIGS["C2"]
mul = cfg.get("scale_mul", (1, 1))
if s == 0:
    fn = lambda: d.sample(batch_size=B, scale_0_size=d.target_size(0, mul, True, 0), s=0)
else:
    prev = torch.randn(B, 3, *d.image_sizes[s - 1], device=dev).clamp(-1, 1)
    fn = lambda: d.sample_via_scale(B, prev, s=s, scale_mul=mul, custom_sample=True, custom_img_size_idx=s, custom_t=d.num_timesteps_ideal[s])
fn(); torch.cuda.synchronize()
t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
print("scale", s, "batch", B, "seconds", time.perf_counter() - t0)
