"""One SinDDMConvBlock forward + backward (sinddm_debug_block_train) at a training shape conv_wh takes: binary16 path (switch 3)
against the fp32-MFMA path (switch 0) -- output, input gradient, per-sample condition gradient, parameter gradients."""
import sys, torch
sys.path.insert(0, '/root/repo')
from sinddm_amd import _lib
from sinddm_amd.models import SinDDMNet, _workspace
from sinddm_amd.synth import closed_form_state_dict, hash_randn
DEV = 'cuda:0'
lib = _lib.load()
dim, B, H, W = 160, 8, 186, 248
net = SinDDMNet(dim=dim, multiscale=True, device=DEV).to(DEV)
net.load_state_dict(closed_form_state_dict(dim))
st = _lib.stream_ptr(DEV)
ws = _workspace(DEV, lib.sinddm_train_workspace_bytes(dim, B, H, W), tag="train")
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
for li in range(4):
    cin, cout = [(3, 80), (80, 160), (160, 160), (160, 80)][li]
    x = hash_randn((B, cin, H, W), 5 + li).to(DEV)
    cb = (0.1 * hash_randn((B, cin), 50 + li)).to(DEV)
    gy = (hash_randn((B, cout, H, W), 60 + li) / (B * 3 * H * W)).to(DEV)
    res = []
    for mode in (3, 0):
        prev = lib.sinddm_debug_set_h2(mode)
        y = torch.empty(B, cout, H, W, device=DEV); gx = torch.empty(B, cin, H, W, device=DEV)
        dc = torch.zeros(B, cin, device=DEV); gr = torch.zeros_like(net.flat_params)
        _lib.check(lib.sinddm_debug_block_train(_lib.ptr(net.flat_params), _lib.ptr(net.packed_weights()), _lib.ptr(net.packed_weights_bwd()),
                   dim, li, _lib.ptr(x), _lib.ptr(cb), _lib.ptr(gy), _lib.ptr(y), _lib.ptr(gx), _lib.ptr(gr), _lib.ptr(dc), B, H, W,
                   ws.data_ptr(), ws.numel(), st), "blk")
        torch.cuda.synchronize(); lib.sinddm_debug_set_h2(prev)
        res.append((y.cpu(), gx.cpu(), dc.cpu(), gr.cpu()))
    (y1, g1, d1, r1), (y0, g0, d0, r0) = res
    print(f'block {li} ({cin}->{cout}): y {rel(y1, y0):.3e}  grad_x {rel(g1, g0):.3e}  dcond {rel(d1, d0):.3e}  all grads {rel(r1, r0):.3e}')
    print('   dcond per sample:', [f'{rel(d1[i], d0[i]):.2e}' for i in range(B)])
    print('   grad_x per sample:', [f'{rel(g1[i], g0[i]):.2e}' for i in range(B)])
    print('   mean(grad_x) per path', float(g1.double().mean()), float(g0.double().mean()), ' sum|.|', float(g0.double().abs().mean()))
    for name, p in net.named_parameters():
        off = (p.data_ptr() - net.flat_params.data_ptr()) // 4
        a, b = r1[off:off + p.numel()], r0[off:off + p.numel()]
        if float(b.abs().max()) > 0:
            print(f'   {name:40s} {rel(a, b):.3e}')
