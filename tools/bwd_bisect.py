#!/usr/bin/env python3
"""Which part of the backward tolerances is Winograd?  

One full-size backward (C2 finest scale 186x248, dim 160, B = 2 -- the shape of tests/test_gpu_train.py::
test_net_backward_full_size_vs_oracle_autograd) through whatever libsinddm_hip.so is installed, compared with the
oracle's autograd in float64 (the reference value) and in float32 (what the test compares with).  Run it once per library
variant (tools/bwd_bisect.sh swaps them): F(2x4) data gradients (default), F(2x2) (-DSINDDM_WINO_V4=0 -DSINDDM_WINO_V3=0),
direct convolution (-DSINDDM_CONV_WINO=0), each with the Winograd and the direct weight gradient.
Prints one JSON line: rel-L2 errors against the float64 oracle per gradient class."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import rel_l2
from oracle import sinddm_oracle as O
from sinddm_amd.models import SinDDMNet
from sinddm_amd.synth import closed_form_state_dict, hash_randn

tag = sys.argv[1] if len(sys.argv) > 1 else "installed"
dev = torch.device("cuda:0")
dim, B, H, W = 160, 2, 186, 248
net = SinDDMNet(dim=dim, multiscale=True, device=dev).to(dev)
net.load_state_dict(closed_form_state_dict(dim))
net.bind_grads()
net.flat_grads.zero_()
x = hash_randn((B, 3, H, W), 15)
gy = hash_randn((B, 3, H, W), 16) / (B * 3 * H * W)
t = torch.tensor([731, 12])
xd = x.to(dev).requires_grad_(True)
y = net(xd, t.to(dev), scale=4)
y.backward(gy.to(dev))


def oracle(dtype):
    torch.set_default_dtype(dtype)
    sd = {k: v.to(dtype).clone().requires_grad_(True) for k, v in closed_form_state_dict(dim).items()}
    xc = x.to(dtype).clone().requires_grad_(True)
    yc = O.net_forward(sd, xc, t, 4)
    yc.backward(gy.to(dtype))
    torch.set_default_dtype(torch.float32)
    return yc.detach(), xc.grad, {k: v.grad for k, v in sd.items()}


y64, gx64, g64 = oracle(torch.float64)
y32, gx32, g32 = oracle(torch.float32)


def classes(grads, gx, yy):
    out = {"forward": rel_l2(yy.double(), y64), "grad_input": rel_l2(gx.double(), gx64)}
    worst = {"conv3x3_weight": 0.0, "conv_other_weight": 0.0, "cond_path": 0.0, "bias_other": 0.0}
    for name, g in grads.items():
        e = rel_l2(g.double().cpu(), g64[name])
        cond = ".mlp." in name or "time_mlp" in name or "time_reshape" in name or name.endswith("ds_conv.bias")
        if cond:
            k = "cond_path"
        elif name.endswith("weight") and g.dim() == 4 and g.shape[-1] == 3 and g.shape[1] >= 16:
            k = "conv3x3_weight"
        elif name.endswith("weight"):
            k = "conv_other_weight"
        else:
            k = "bias_other"
        worst[k] = max(worst[k], e)
    out.update(worst)
    return {k: float(f"{v:.3g}") for k, v in out.items()}


gpu = classes({n: p.grad for n, p in net.named_parameters()}, xd.grad.cpu(), y.detach().cpu())
cpu32 = classes(g32, gx32, y32)
print(json.dumps({"variant": tag, "hip_vs_f64_oracle": gpu, "f32_oracle_vs_f64_oracle": cpu32}))
