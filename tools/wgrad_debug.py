#!/usr/bin/env python3
"""Debug: gradients of a B=24 96x128 backward through the installed library; dumps l3.net.0.weight etc. to gpurun_out."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sinddm_amd.models import SinDDMNet
from sinddm_amd.synth import closed_form_state_dict, hash_randn
tag = sys.argv[1]
dev = torch.device("cuda:0")
B, H, W = 24, 96, 128
net = SinDDMNet(dim=160, multiscale=True, device=dev).to(dev)
net.load_state_dict(closed_form_state_dict(160))
net.bind_grads()
x = hash_randn((B, 3, H, W), 5)
gy = hash_randn((B, 3, H, W), 6) / (B * 3 * H * W)
t = torch.tensor([(91 * (i + 1)) % 1000 for i in range(B)], dtype=torch.long)
outs = []
for rep in range(3):
    net.flat_grads.zero_()
    xd = x.to(dev).requires_grad_(True)
    y = net(xd, t.to(dev), scale=3)
    y.backward(gy.to(dev))
    outs.append({n: p.grad.detach().cpu().clone() for n, p in net.named_parameters()})
names = [n for n in outs[0] if n.endswith("weight") and outs[0][n].dim() == 4 and outs[0][n].shape[-1] == 3]
for n in names:
    a, b, c = outs[0][n], outs[1][n], outs[2][n]
    rl = lambda u, v: float((u - v).norm() / v.norm())
    print(tag, n, tuple(a.shape), "rep01", f"{rl(a, b):.2e}", "rep02", f"{rl(a, c):.2e}")
torch.save({n: outs[0][n] for n in names}, f"gpurun_out/wgrad_debug_{tag}.pt")
