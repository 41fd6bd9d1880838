"""Where does conv_wino6 differ from conv_wino4?  One 160 -> 160 block (sinddm_debug_block_train) with the switch on / off."""
import sys, torch
sys.path.insert(0, '/root/repo')
from sinddm_amd import _lib
from sinddm_amd.models import SinDDMNet, _workspace
from sinddm_amd.synth import closed_form_state_dict, hash_randn
DEV = 'cuda:0'
lib = _lib.load()
dim, B, H, W = 160, 8, 64, 128
li = int(sys.argv[1]) if len(sys.argv) > 1 else 2
net = SinDDMNet(dim=dim, multiscale=True, device=DEV).to(DEV)
net.load_state_dict(closed_form_state_dict(dim))
cin, cout = [(3, 80), (80, 160), (160, 160), (160, 80)][li]
x = hash_randn((B, cin, H, W), 5).to(DEV)
cb = torch.zeros(B, cin, device=DEV)
gy = torch.zeros(B, cout, H, W, device=DEV)
st = _lib.stream_ptr(DEV)
ws = _workspace(DEV, lib.sinddm_train_workspace_bytes(dim, B, H, W), tag="train")
ys = []
for on in (0, 1):
    lib.sinddm_debug_set_f44(on)
    print('path', lib.sinddm_debug_conv_path(160, B, H, W))
    y = torch.empty(B, cout, H, W, device=DEV); gx = torch.empty(B, cin, H, W, device=DEV)
    dc = torch.zeros(B, cin, device=DEV); gr = torch.zeros_like(net.flat_params)
    _lib.check(lib.sinddm_debug_block_train(_lib.ptr(net.flat_params), _lib.ptr(net.packed_weights()), _lib.ptr(net.packed_weights_bwd()),
               dim, li, _lib.ptr(x), _lib.ptr(cb), _lib.ptr(gy), _lib.ptr(y), _lib.ptr(gx), _lib.ptr(gr), _lib.ptr(dc), B, H, W,
               ws.data_ptr(), ws.numel(), st), "blk")
    torch.cuda.synchronize(); ys.append(y.cpu())
a, b = ys
d = (a - b).abs()
print('rel', float((a - b).norm() / a.norm()), 'max', float(d.max()), 'ref absmax', float(a.abs().max()))
print('by m-tile (16 ch):', [round(float(d[:, m * 16:(m + 1) * 16].max()), 4) for m in range(cout // 16)])
print('by row % 8:', [round(float(d[:, :, r::8].max()), 4) for r in range(8)])
print('by col % 32 (first 16):', [round(float(d[:, :, :, c::32].max()), 3) for c in range(16)])
print('by col % 4:', [round(float(d[:, :, :, c::4].max()), 4) for c in range(4)], 'by row % 4:', [round(float(d[:, :, r::4].max()), 4) for r in range(4)])
print('by sample:', [round(float(d[i].max()), 4) for i in range(B)])
print('interior only (rows 8..55, cols 32..95):', float(d[:, :, 8:56, 32:96].max()))
