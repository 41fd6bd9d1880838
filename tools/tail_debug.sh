ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for v in NOTAIL TAIL; do cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so; python tools/tail_debug.py $v 2>&1 | grep -v amdgpu | tail -2; done
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
python - <<'PY'
import torch
a=torch.load("gpurun_out/tail_debug_NOTAIL.pt"); b=torch.load("gpurun_out/tail_debug_TAIL.pt")
for k in a:
    d=(a[k]-b[k]).abs()
    print(k, "max diff", float(d.max()), "HW%4", (k[1]*k[2])%4)
    if float(d.max())>1e-5:
        print((d[0,0]>1e-5).int())
PY
