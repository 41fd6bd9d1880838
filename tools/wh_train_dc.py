"""Is the error of the binary16 conv path white?  Block 2 (160 -> 160) forward + backward at a training shape conv_wh takes,
both paths against a float64 torch evaluation of the same block: rel-L2, and the DC statistic
|sum over pixels of the error| / (sqrt(N) * rms of the error) per (sample, channel) -- ~0.8 for white errors."""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from sinddm_amd import _lib
from sinddm_amd.models import SinDDMNet, _workspace
from sinddm_amd.synth import closed_form_state_dict, hash_randn
DEV = 'cuda:0'
lib = _lib.load()
dim, B, H, W = 160, 8, 186, 248
li = int(sys.argv[1]) if len(sys.argv) > 1 else 2
xs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
net = SinDDMNet(dim=dim, multiscale=True, device=DEV).to(DEV)
sd = closed_form_state_dict(dim)
net.load_state_dict(sd)
st = _lib.stream_ptr(DEV)
ws = _workspace(DEV, lib.sinddm_train_workspace_bytes(dim, B, H, W), tag="train")
cin, cout = [(3, 80), (80, 160), (160, 160), (160, 80)][li]
name = f'l{li + 1}'
x = xs * hash_randn((B, cin, H, W), 5 + li)
cb = 0.1 * hash_randn((B, cin), 50 + li)
gy = hash_randn((B, cout, H, W), 60 + li) / (B * 3 * H * W)
# float64 reference
D = torch.float64
xr = x.to(D).requires_grad_(True); cbr = cb.to(D).requires_grad_(True)
w = {k: sd[f'{name}.{k}'].to(D).requires_grad_(True) for k in ('ds_conv.weight', 'ds_conv.bias', 'net.0.weight', 'net.0.bias', 'net.2.weight', 'net.2.bias')}
h = F.conv2d(xr, w['ds_conv.weight'], w['ds_conv.bias'], padding=2, groups=cin) + cbr[:, :, None, None]
u = F.conv2d(h, w['net.0.weight'], w['net.0.bias'], padding=1)
g = F.gelu(u)
o = F.conv2d(g, w['net.2.weight'], w['net.2.bias'], padding=1) + xr
o.backward(gy.to(D))
ref = dict(y=o.detach(), gx=xr.grad, dcond=cbr.grad, **{k: v.grad for k, v in w.items()})
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
def dc(a, b):
    d = a.double() - b.double()
    n = d.shape[-1] * d.shape[-2]
    return float((d.sum((-1, -2)).abs() / ((d.pow(2).mean((-1, -2)).sqrt() * n ** 0.5) + 1e-300)).mean())
for mode in (3, 0):
    prev = lib.sinddm_debug_set_h2(mode)
    y = torch.empty(B, cout, H, W, device=DEV); gx = torch.empty(B, cin, H, W, device=DEV)
    dcd = torch.zeros(B, cin, device=DEV); gr = torch.zeros_like(net.flat_params)
    xd, cbd, gyd = x.to(DEV), cb.to(DEV), gy.to(DEV)
    _lib.check(lib.sinddm_debug_block_train(_lib.ptr(net.flat_params), _lib.ptr(net.packed_weights()), _lib.ptr(net.packed_weights_bwd()),
               dim, li, _lib.ptr(xd), _lib.ptr(cbd), _lib.ptr(gyd), _lib.ptr(y), _lib.ptr(gx), _lib.ptr(gr), _lib.ptr(dcd), B, H, W,
               ws.data_ptr(), ws.numel(), st), "blk")
    torch.cuda.synchronize(); lib.sinddm_debug_set_h2(prev)
    y, gx, dcd, gr = y.cpu(), gx.cpu(), dcd.cpu(), gr.cpu()
    print(f'mode {mode} path {lib.sinddm_debug_train_path(dim, B, H, W)}: y {rel(y, ref["y"]):.3e} (dc {dc(y, ref["y"]):.2f})  grad_x {rel(gx, ref["gx"]):.3e} (dc {dc(gx, ref["gx"]):.2f})  dcond {rel(dcd, ref["dcond"]):.3e}')
    for pname, p in net.named_parameters():
        if pname.startswith(name + '.') and pname[len(name) + 1:] in ref:
            off = (p.data_ptr() - net.flat_params.data_ptr()) // 4
            print(f'     {pname:24s} {rel(gr[off:off + p.numel()].reshape(p.shape), ref[pname[len(name) + 1:]]):.3e}')
