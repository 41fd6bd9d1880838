#!/usr/bin/env python3
"""Where does a conv_wino4 work item spend its time?  Needs a -DW4_TIMING build at tools/ab/lib<name>.so (argv[1],
default T): per Winograd launch of one network evaluation (C3 finest scale, batch 8) the s_memtime stamps of every
workgroup's item 3: main loop, and per epilogue pass (column transform + LDS write | barrier | LDS reads + row
transform | barrier | activation / residual / stores)."""
import ctypes as C, os, sys, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else "T"
NW = 4
shutil.copy(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"), "/tmp/lib_keep.so")
shutil.copy(os.path.join(ROOT, "tools", "ab", f"lib{name}.so"), os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
try:
    from sinddm_amd import _lib
    from sinddm_amd.configs import build_diffusion
    lib = _lib.load()
    if len(sys.argv) > 2 and sys.argv[2] == "f44":
        lib.sinddm_debug_set_f44(1)
    dev = torch.device("cuda:0")
    net, d = build_diffusion("C3", 160, dev)
    x = torch.randn(8, 3, 411, 512, device=dev)
    for _ in range(2):
        y = net.infer(x, None, 10, 5.0)
    torch.cuda.synchronize()
    n = 8 * 256 * NW * 32
    buf = (C.c_ulonglong * n)()
    f = C.CDLL(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so")).sinddm_debug_w4_seg
    f.argtypes = [C.c_void_p, C.c_int]
    assert f(buf, n) == 0
    a = np.array(buf, dtype=np.uint64).reshape(8, 256, NW, 32).astype(np.int64)
    names = ["80->80", "80->160 GELU", "160->160", "160->160 GELU", "160->160", "160->80 GELU", "80->80"]
    # 7 launches per evaluation, 2 evaluations = 14 launches -> rows (launch % 8); the second evaluation's launches 7..13
    # overwrite rows 7, 0..5: row of launch j of the 2nd evaluation = (7 + j) % 8
    # stamps: 0 item start, 1 main loop done, per pass p (= m-tile): 2+3p pass start (dump of p issued), 3+3p behind the
    # barrier, 4+3p transforms done (finish: bias / residual / GELU / stores follows), 20 end of the epilogue
    for j in range(7):
        r = a[(7 + j) % 8]
        ok = r[:, :, 1] > 0
        t = r[ok]
        main = t[:, 1] - t[:, 0]
        epi = t[:, 20] - t[:, 1]
        print(f"launch {j} {names[j]:14s}: main loop {main.mean():8.0f}  epilogue {epi.mean():7.0f} ({100 * epi.mean() / (epi.mean() + main.mean()):4.1f} % of the item)")
        for p in range(5):
            b = 2 + 3 * p
            prev = t[:, 1] if p == 0 else t[:, b - 1]
            nxt = t[:, b + 3] if p < 4 else t[:, 20]
            print(f"     pass {p}: [prev finish + dump] {(t[:, b] - prev).mean():6.0f}  [fetch + barrier] {(t[:, b + 1] - t[:, b]).mean():5.0f}  "
                  f"[LDS reads + transforms] {(t[:, b + 2] - t[:, b + 1]).mean():5.0f}  [finish -> next pass] {(nxt - t[:, b + 2]).mean():5.0f}")
finally:
    shutil.copy("/tmp/lib_keep.so", os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
