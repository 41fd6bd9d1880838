#!/usr/bin/env python3
"""Where inside a k-step does a conv_wino4 wave lose time?  Needs a -DW4_KSTAMP build at tools/ab/lib<name>.so (argv[1],
default K): s_memtime at slots 0, 8, .. 56 of the four k-steps of chunk 2 of every workgroup's item 4, in front of and
behind the transform burst, and at slot 59.  Eight MFMA slots = 256 cycles of matrix pipe."""
import ctypes as C, os, sys, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else "K"
shutil.copy(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"), "/tmp/lib_keep.so")
shutil.copy(os.path.join(ROOT, "tools", "ab", f"lib{name}.so"), os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
try:
    from sinddm_amd import _lib
    from sinddm_amd.configs import build_diffusion
    lib = _lib.load()
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
    names = ["80->80", "80->160 GELU", "160->160", "160->160 GELU", "160->160", "160->80 GELU", "80->80"]
    for j in (0, 2, 3):
        r = a[(7 + j) % 8]
        ok = r[:, :, 0, 0] > 0
        t = r[ok]                                   # (n, 4 k-steps, 16)
        print(f"launch {j} {names[j]}: {t.shape[0]} waves; cycles per 8-slot segment (256 = matrix-pipe bound)")
        for ks in range(4):
            seg = [(t[:, ks, i + 1] - t[:, ks, i]).mean() for i in range(7)]
            burst = (t[:, ks, 9] - t[:, ks, 8]).mean()
            tail = (t[:, ks, 10] - t[:, ks, 7]).mean()
            nxt = (t[:, (ks + 1) % 4, 0] - t[:, ks, 10]).mean() if ks < 3 else float('nan')
            tot = (t[:, ks, 10] - t[:, ks, 0]).mean()
            fine = [(t[:, ks, 11] - t[:, ks, 9]).mean()] + [(t[:, ks, 12 + i] - t[:, ks, 11 + i]).mean() for i in range(4)] + [(t[:, ks, 6] - t[:, ks, 15]).mean()]
            print("      slots 41->42, 42->43 .. 45->46, 46->48: " + " ".join(f"{v:5.0f}" for v in fine))
            print(f"   k-step {ks}: segments " + " ".join(f"{v:6.0f}" for v in seg) + f" | slots 56-59 {tail:5.0f} | burst+1 MFMA {burst:5.0f}"
                  f" | slot 0 -> 59: {tot:6.0f} (ideal 1888) | to next k-step {nxt:5.0f}")
finally:
    shutil.copy("/tmp/lib_keep.so", os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
