#!/usr/bin/env python3
"""Per-scale timing of the sampler (C2, B=16) and of one training step (B=32) on the GPU."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sinddm_amd.configs import CONFIGS, build_diffusion
from sinddm_amd.optim import FusedAdam

dev = torch.device("cuda:0")
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
net, d = build_diffusion(cfg_name, 160, dev)
cfg = CONFIGS[cfg_name]
FL = 2150230
res = {"sample": [], "train": []}
for s in range(len(cfg["sizes"])):
    H, W = d.target_size(s, cfg.get("scale_mul", (1, 1)), True, s)
    x = torch.randn(B, 3, H, W, device=dev)
    d.img_prev_upsample = torch.randn(B, 3, H, W, device=dev)
    for _ in range(3):
        x = d._p_sample_host_t(x, 50, s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for i in range(n):
        x = d._p_sample_host_t(x, 60 + i, s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    res["sample"].append(dict(s=s, H=H, W=W, ms=round(dt * 1e3, 3), tflops=round(FL * B * H * W / dt / 1e12, 1),
                              steps_at_scale=d.num_timesteps_ideal[s]))
# training step (forward + backward + Adam), batch 32 identical images like the reference trainer
opt = FusedAdam(net, lr=1e-3)
TB = 32
for s in range(len(cfg["sizes"])):
    H, W = d.image_sizes[s]
    img = torch.randn(TB, 3, H, W, device=dev).clamp(-1, 1)
    data = (img, img.clone())
    for _ in range(2):
        loss = d(data, s); loss.backward(); opt.step(); opt.zero_grad()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        loss = d(data, s); loss.backward(); opt.step(); opt.zero_grad()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    res["train"].append(dict(s=s, H=H, W=W, ms=round(dt * 1e3, 2), steps_per_s=round(1 / dt, 2),
                             tflops=round(3 * FL * TB * H * W / dt / 1e12, 1)))
print(json.dumps(res, indent=1))
