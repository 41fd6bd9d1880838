#!/usr/bin/env python3
"""Per-wave s_memtime stamps of conv_wino2_kernel (needs a -DW2_TIMING build at tools/ab/libT.so): for the first work
items of workgroups 8 and 8+256 print, per 16-channel chunk, the span of each wave and its barrier wait."""
import ctypes as C, os, sys, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
    n = 2 * 4 * 40 * 4
    buf = (C.c_ulonglong * n)()
    f = C.CDLL(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so")).sinddm_debug_w2_timing
    f.argtypes = [C.c_void_p, C.c_int]
    assert f(buf, n) == 0
    a = np.array(buf, dtype=np.uint64).reshape(2, 4, 40, 4).astype(np.int64)   # [wg][item][slot][wave]
    # the last Winograd launch of the forward is l4.conv2 (80->80, 5 chunks)
    for wg in range(2):
        for it in range(4):
            hw = a[wg, it, 39]
            print(f"wg {wg} item {it}  HW_ID wave_id/simd/cu per wave:", [(int(h) & 15, (int(h) >> 4) & 3, (int(h) >> 8) & 15) for h in hw])
            prev_end = None
            for c in range(12):
                t0, t1, t2 = a[wg, it, 3 * c], a[wg, it, 3 * c + 1], a[wg, it, 3 * c + 2]
                if t0.max() == 0:
                    continue
                print(f"   chunk {c:2d}: start spread {int(t0.max()-t0.min()):5d} compute min/mean/max {int((t1-t0).min()):6d} {int((t1-t0).mean()):6d} "
                      f"{int((t1-t0).max()):6d}  barrier wait min/max {int((t2-t1).min()):5d} {int((t2-t1).max()):5d}  chunk total {int(t2.max()-t0.min()):6d}"
                      + (f"  gap from prev {int(t0.min()-prev_end):6d}" if prev_end is not None else ""))
                prev_end = t2.max()
            e0, e1 = a[wg, it, 36], a[wg, it, 37]
            first = a[wg, it, 0]
            print(f"   main loop {int(e0.max()-first.min()):7d}   epilogue {int(e1.max()-e0.min()):7d}   item total {int(e1.max()-first.min()):7d}"
                  + (f"   next item starts {int(a[wg, it+1, 0].min()-e1.max()):6d} after" if it < 3 and a[wg, it+1, 0].max() > 0 else ""))
finally:
    shutil.copy("/tmp/lib_keep.so", os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
