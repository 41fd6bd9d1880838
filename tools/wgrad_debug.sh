#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for v in $1; do cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so; python tools/wgrad_debug.py $v 2>&1 | grep -v Warn | tail -8; done
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
python - <<'PY'
import torch
a = torch.load("gpurun_out/wgrad_debug_WIDE.pt"); b = torch.load("gpurun_out/wgrad_debug_NARROW.pt")
for n in a:
    d = (a[n] - b[n]); print("WIDE vs NARROW", n, f"{float(d.norm() / b[n].norm()):.2e}", "max at", int(d.abs().argmax()), "of", d.numel())
    if float(d.norm() / b[n].norm()) > 1e-3:
        idx = (d.abs() > 0.05 * b[n].abs().max()).nonzero()
        print("   bad entries", idx.shape[0], idx[:12].tolist())
PY
