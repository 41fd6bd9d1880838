#!/usr/bin/env python3
"""k-step internals of conv_wino2_kernel (needs -DW2_TIMING -DW2_KT_KS=<ks> builds at tools/ab/libK<ks>.so): cycles
spent in each m-tile group of one k-step of chunk 2, per wave."""
import ctypes as C, os, sys, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ks = sys.argv[1]
shutil.copy(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"), "/tmp/lib_keep.so")
shutil.copy(os.path.join(ROOT, "tools", "ab", f"libK{ks}.so"), os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
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
    a = np.array(buf, dtype=np.uint64).reshape(2, 4, 40, 4).astype(np.int64)
    for wg in range(2):
        for it in range(1, 4):
            kt = a[wg, it, 20:26]          # [6 stamps][wave]
            if kt.max() == 0:
                continue
            d_ = np.diff(kt, axis=0)
            print(f"ks {ks} wg {wg} item {it}: per-group cycles (rows = m-tile group 0..4, cols = waves):")
            for g in range(5):
                print("    group", g, [int(v) for v in d_[g]])
            print("    k-step total", [int(v) for v in (kt[5] - kt[0])])
            pt = a[wg, it, 30:34]
            if pt.max() > 0:
                print("    piece stamps (after MFMA 1,3,5,7 of the probed group) relative to k-step start:",
                      [[int(pt[i][w] - kt[0][w]) for w in range(4)] for i in range(4)])
    sys.stdout.flush()
    os._exit(0)
finally:
    shutil.copy("/tmp/lib_keep.so", os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
