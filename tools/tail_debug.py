import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sinddm_amd.models import SinDDMNet
from sinddm_amd.synth import closed_form_state_dict, hash_randn
tag = sys.argv[1]
dev = "cuda:0"
net = SinDDMNet(dim=160, multiscale=True, device=dev).to(dev)
net.load_state_dict(closed_form_state_dict(160))
out = {}
for (B, H, W) in ((1, 5, 7), (1, 5, 9), (1, 5, 6), (2, 9, 33), (1, 13, 21)):
    x = hash_randn((B, 3, H, W), 7).to(dev)
    t = torch.tensor([5] * B, device=dev)
    with torch.no_grad():
        y = net(x, t, scale=1)
    out[(B, H, W)] = y.cpu()
torch.save(out, f"gpurun_out/tail_debug_{tag}.pt")
