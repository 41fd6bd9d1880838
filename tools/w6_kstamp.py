#!/usr/bin/env python3
"""Where inside a k-step does a conv_wino6 wave lose time?  Needs a -DW4_KSTAMP build at tools/ab/lib<name>.so (argv[1]):
s_memtime at slots 0, 5, .. 40, 44 of the four k-steps of chunk 2 of every workgroup's item 4 and around the transform
burst.  Five MFMA slots = 160 cycles of matrix pipe."""
import ctypes as C, os, sys, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1]
shutil.copy(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"), "/tmp/lib_keep.so")
shutil.copy(os.path.join(ROOT, "tools", "ab", f"lib{name}.so"), os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
try:
    from sinddm_amd import _lib
    from sinddm_amd.configs import build_diffusion
    lib = _lib.load()
    lib.sinddm_debug_set_f44(1)
    dev = torch.device("cuda:0")
    net, d = build_diffusion("C3", 160, dev)
    x = torch.randn(8, 3, 411, 512, device=dev)
    for _ in range(2):
        y = net.infer(x, None, 10, 5.0)
    torch.cuda.synchronize()
    n = 8 * 256 * 4 * 64
    buf = (C.c_ulonglong * n)()
    f = C.CDLL(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so")).sinddm_debug_w4_ks
    f.argtypes = [C.c_void_p, C.c_int]
    assert f(buf, n) == 0
    a = np.array(buf, dtype=np.uint64).reshape(8, 256, 4, 4, 16).astype(np.int64)
    for row in range(8):
        r = a[row]
        ok = r[:, :, 0, 0] > 0
        t = r[ok]
        if t.shape[0] == 0:
            continue
        print(f"stamp row {row}: {t.shape[0]} waves; cycles per 5-slot segment (160 = matrix-pipe bound)")
        for ks in range(4):
            seg = [(t[:, ks, i + 1] - t[:, ks, i]).mean() for i in range(8)]
            tail = (t[:, ks, 9] - t[:, ks, 8]).mean()
            burst = (t[:, ks, 11] - t[:, ks, 10]).mean()
            nxt = (t[:, (ks + 1) % 4, 0] - t[:, ks, 9]).mean() if ks < 3 else float('nan')
            tot = (t[:, ks, 9] - t[:, ks, 0]).mean()
            print(f"   k-step {ks}: " + " ".join(f"{v:5.0f}" for v in seg) + f" | slots 40-44 {tail:4.0f} (128) | burst + 1 MFMA {burst:4.0f}"
                  f" | slot 0 -> 44: {tot:5.0f} (ideal 1408) | to next k-step {nxt:4.0f}")
finally:
    shutil.copy("/tmp/lib_keep.so", os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
