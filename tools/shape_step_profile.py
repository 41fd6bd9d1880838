#!/usr/bin/env python3
"""Kernel-level profile target: 40 reverse steps at an arbitrary image size (batch 16, dim 160, C2 schedule, scale 3):
python tools/shape_step_profile.py H W"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sinddm_amd.configs import build_diffusion
dev = torch.device("cuda:0")
H, W = int(sys.argv[1]), int(sys.argv[2])
net, d = build_diffusion("C2", 160, dev)
x = torch.randn(16, 3, H, W, device=dev)
d.img_prev_upsample = torch.randn(16, 3, H, W, device=dev)
for i in range(40):
    x = d._p_sample_host_t(x, 60 + (i % 30), 3)
torch.cuda.synchronize()
print("done", H, W)
