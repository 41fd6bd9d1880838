import sys, torch
sys.path.insert(0, '/root/repo')
from sinddm_amd.models import SinDDMNet
from sinddm_amd.synth import closed_form_state_dict, hash_randn
DEV = 'cuda:0'
net = SinDDMNet(dim=160, multiscale=True, device=DEV).to(DEV)
net.load_state_dict(closed_form_state_dict(160))
for (B, H, W) in [(24, 90, 128), (24, 96, 128), (40, 48, 64), (24, 94, 128), (24, 90, 126)]:
    x = hash_randn((B, 3, H, W), 77 + W) * 0.9
    t = torch.tensor([(53 * (i + 3)) % 1000 for i in range(B)], dtype=torch.long)
    with torch.no_grad():
        got = net(x.to(DEV), t.to(DEV), scale=2).cpu()
    worst = 0
    for i in (0, B - 1, B // 2):
        yi = net.infer(x[i:i + 1].to(DEV).contiguous(), None, int(t[i]), 2.0).cpu()
        d = (got[i:i + 1] - yi).abs()
        rel = float(d.norm() / yi.norm())
        worst = max(worst, rel)
        if rel > 1e-5:
            rows = d.amax(dim=(0, 1, 3)); cols = d.amax(dim=(0, 1, 2))
            print((B, H, W), 'sample', i, 'rel', rel, 'bad rows', [int(r) for r in torch.nonzero(rows > 1e-4).flatten()][:40],
                  'bad cols', [int(c) for c in torch.nonzero(cols > 1e-4).flatten()][:40])
    print((B, H, W), 'worst rel', worst)
