#!/usr/bin/env python3
"""Per-wave s_memtime stamps of the Winograd kernel (needs a -DSINDDM_WINO_TIMING build at tools/ab/libT.so):
for the first two work items of one workgroup print, per 16-channel chunk, the compute span of each wave
(chunk start -> last MFMA issued) and how long it then waited at the chunk barrier."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shutil
shutil.copy(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"), "/tmp/lib_keep.so")
shutil.copy(os.path.join(ROOT, "tools", "ab", "libT.so"), os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
try:
    from sinddm_amd import _lib
    from sinddm_amd.configs import build_diffusion
    lib = _lib.load()
    dev = torch.device("cuda:0")
    net, d = build_diffusion("C2", 160, dev)
    x = torch.randn(16, 3, 186, 248, device=dev)
    for _ in range(3):
        y = net.infer(x, None, 10, 4.0)
    torch.cuda.synchronize()
    n = 2 * 16 * 16 * 4
    buf = (C.c_ulonglong * n)()
    f = C.CDLL(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so")).sinddm_debug_wino_timing
    f.argtypes = [C.c_void_p, C.c_int]
    assert f(buf, n) == 0
    a = np.array(buf, dtype=np.uint64).reshape(2, 16, 16, 4).astype(np.int64)   # [item][chunk][wave][stamp]
    # the last Winograd launch of the forward is l4.conv2 (80->80, 5 chunks); stamps are s_memtime ticks (100 MHz -> x24 shader cycles?)
    for it in range(2):
        print("item", it)
        for c in range(16):
            t0, t1, t2 = a[it, c, :, 0], a[it, c, :, 1], a[it, c, :, 2]
            if t0.max() == 0:
                continue
            base = t0.min()
            t3 = a[it, c, :, 3]
            print(f" chunk {c:2d}: first MFMA group issued after min/mean/max {int((t3-t0).min()):5d} {int((t3-t0).mean()):5d} {int((t3-t0).max()):5d} |", end="")
            print(f" start spread {t0.max()-t0.min():5d}  compute min/mean/max {int((t1-t0).min()):6d} {int((t1-t0).mean()):6d} {int((t1-t0).max()):6d}"
                  f"  barrier wait min/mean/max {int((t2-t1).min()):6d} {int((t2-t1).mean()):6d} {int((t2-t1).max()):6d}  chunk total {int(t2.max()-base):6d}")
        if a[it, :, :, 0].max() > 0:
            cc = [c for c in range(16) if a[it, c, :, 0].max() > 0]
            print("  item span (first chunk start -> last chunk end):", int(a[it, cc[-1], :, 2].max() - a[it, cc[0], :, 0].min()),
                  " next item start - this item end:", int(a[1, 0, :, 0].min() - a[0, cc[-1], :, 2].max()) if it == 0 else "")
finally:
    shutil.copy("/tmp/lib_keep.so", os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
