#!/usr/bin/env python3
"""Where does a conv_wh work item spend its time?  Needs a -DWH_TIMING build at tools/ab/lib<name>.so (argv[1], default
WT): s_memtime stamps (100 MHz constant clock -> x clk/100MHz) of every workgroup's third item, per launch of one network
evaluation (C3 finest scale, batch 8), for a multiplying wave (0) and a service wave (6)."""
import ctypes as C, os, sys, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else "WT"
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
    n = 8 * 256 * 8 * 32
    buf = (C.c_ulonglong * n)()
    f = C.CDLL(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so")).sinddm_debug_wh_seg
    f.argtypes = [C.c_void_p, C.c_int]
    assert f(buf, n) == 0
    a = np.array(buf, dtype=np.uint64).reshape(8, 256, 8, 32).astype(np.int64)
    names = ["80->80", "80->160 GELU", "160->160", "160->160 GELU", "160->160", "160->80 GELU", "80->80"]
    for jl in range(7):
        r = a[(7 + jl) % 8]
        print(f"== launch {jl} {names[jl]}  (units: shader clock cycles)")
        for w in (0, 5, 6):
            t = r[:, w, :]
            ok = (t[:, 23] > 0) & (t[:, 0] > 0)
            t = t[ok]
            if len(t) == 0:
                print("  wave", w, "no stamps"); continue
            d0 = lambda i, k: np.median(t[:, i] - t[:, k])
            nch = 10 if jl in (2, 3, 4, 5) else 5
            chunks = [d0(3 + c, 2 + c) for c in range(nch)]
            print(f"  wave {w}: item {d0(23, 0):.0f} | prologue wait {d0(1, 0):.0f} T0 {d0(2, 1):.0f} | chunks {' '.join('%.0f' % v for v in chunks)} |"
                  f" -> epilogue start {d0(17, 2 + nch):.0f} | passes {' '.join('%.0f' % d0(18 + k, 17 + k) for k in range(5))} | tail {d0(23, 22):.0f}"
                  + (f" | chunk3: stage+wait {d0(24, 5):.0f} T {d0(25, 24):.0f} barrier {d0(6, 25):.0f}" if w == 6 else f" | chunk3: multiply {d0(25, 5):.0f} barrier {d0(6, 25):.0f}"))
finally:
    shutil.copy("/tmp/lib_keep.so", os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
