#!/usr/bin/env python3
"""Kernel-level profile target: 50 reverse steps at one pyramid scale of C2 (batch 16): python tools/scale_step_profile.py <scale>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sinddm_amd.configs import build_diffusion
dev = torch.device("cuda:0")
s = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
net, d = build_diffusion("C2", 160, dev)
H, W = d.image_sizes[s]
x = torch.randn(B, 3, H, W, device=dev)
d.img_prev_upsample = torch.randn(B, 3, H, W, device=dev)
for i in range(55):
    x = d._p_sample_host_t(x, 60 + (i % 30), s)
torch.cuda.synchronize()
print("done", s, H, W)
