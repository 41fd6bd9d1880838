#!/usr/bin/env python3
"""Per-scale wall time of one full multi-scale sample (C2 by default, batch 16): where the imgs/s go."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sinddm_amd.configs import CONFIGS, build_diffusion
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
net, d = build_diffusion(cfgname, 160, dev)
cfg = CONFIGS[cfgname]
n = len(cfg["sizes"])
mul = cfg.get("scale_mul", (1, 1))
for rep in range(2):
    tot = 0.0
    rows = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cur = d.sample(batch_size=B, scale_0_size=d.target_size(0, mul, True, 0), s=0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    rows.append((0, tuple(cur.shape[-2:]), d.num_timesteps_ideal[0], t1 - t0))
    for si in range(1, n):
        t0 = time.perf_counter()
        cur = d.sample_via_scale(B, cur, s=si, scale_mul=mul, custom_sample=True, custom_img_size_idx=si, custom_t=d.num_timesteps_ideal[si])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rows.append((si, tuple(cur.shape[-2:]), d.num_timesteps_ideal[si], t1 - t0))
    tot = sum(r[3] for r in rows)
    if rep == 1:
        for si, hw, T, dt in rows:
            px = hw[0] * hw[1] * B
            print(f"scale {si} {hw[0]:4d}x{hw[1]:<4d} steps {T:5d}  {dt*1e3:9.1f} ms  {dt/T*1e3:8.3f} ms/step  {dt/tot*100:5.1f} %  {px*T/dt/1e9:7.2f} Gpx-steps/s")
        print(f"total {tot:.3f} s -> {B/tot:.3f} img/s")
