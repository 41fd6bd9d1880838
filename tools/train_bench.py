#!/usr/bin/env python3
"""Training-step benchmark (secondary metric of SURVEY 8(d)): MultiScaleGaussianDiffusion.forward + backward +
fused Adam at batch 32 on one pyramid scale of C2 (default: the finest, 186x248)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sinddm_amd.configs import CONFIGS, build_diffusion
from sinddm_amd.optim import FusedAdam

dev = torch.device("cuda:0")
s = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
net, d = build_diffusion("C2", 160, dev)
opt = FusedAdam(net, lr=1e-3)
H, W = d.image_sizes[s]
img = torch.randn(32, 3, H, W, device=dev).clamp(-1, 1)
data = (img, img.clone())
for _ in range(2):
    loss = d(data, s); loss.backward(); opt.step(); opt.zero_grad()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    loss = d(data, s); loss.backward(); opt.step(); opt.zero_grad()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(json.dumps(dict(scale=s, H=H, W=W, batch=32, ms_per_step=round(dt * 1e3, 2), steps_per_s=round(1 / dt, 3),
                      tflops=round(3 * 2150230 * 32 * H * W / dt / 1e12, 1), loss=float(loss))))
