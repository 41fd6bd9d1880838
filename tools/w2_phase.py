#!/usr/bin/env python3
"""Are the two workgroups of a CU in the epilogue at the same time?  Needs a -DW2_PHASE build at tools/ab/lib<name>.so
(argv[1], default P): per workgroup the s_memtime stamps of its first 9 epilogues (160->160 conv of the C2 forward) and
HW_ID; prints per-CU pairs: item period, epilogue length, and the phase offset between the two workgroups."""
import ctypes as C, os, sys, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else "P"
shutil.copy(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"), "/tmp/lib_keep.so")
shutil.copy(os.path.join(ROOT, "tools", "ab", f"lib{name}.so"), os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
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
    n = 1024 * 20
    buf = (C.c_ulonglong * n)()
    f = C.CDLL(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so")).sinddm_debug_w2_phase
    f.argtypes = [C.c_void_p, C.c_int]
    assert f(buf, n) == 0
    a = np.array(buf, dtype=np.uint64).reshape(1024, 20).astype(np.int64)
    used = [w for w in range(1024) if a[w, 1] != 0]
    print("workgroups with stamps:", len(used))
    cus = {}
    for w in used:
        hw = int(a[w, 0]) & 0xffff
        xcc = (int(a[w, 0]) >> 32) & 0xf
        key = (xcc, hw >> 8)              # cu/sh/se bits
        cus.setdefault(key, []).append(w)
    print("distinct CUs:", len(cus), "workgroups per CU histogram:", np.bincount([len(v) for v in cus.values()]))
    t_all = a[used][:, 1:19].reshape(len(used), 9, 2)
    per = np.diff(t_all[:, :, 0], axis=1)                       # item period
    epi = t_all[:, :, 1] - t_all[:, :, 0]
    print("item period   mean/min/max:", per.mean(), per.min(), per.max(), " (s_memtime ticks)")
    print("epilogue len  mean/min/max:", epi.mean(), epi.min(), epi.max())
    overl = []
    shown = 0
    for key, ws in sorted(cus.items()):
        if len(ws) != 2:
            continue
        A, B = a[ws[0], 1:19].reshape(9, 2), a[ws[1], 1:19].reshape(9, 2)
        # phase of B's epilogue starts relative to A's item period
        P = float(np.diff(A[:, 0]).mean())
        ph = [((B[i, 0] - A[0, 0]) % P) / P for i in range(1, 8)]
        # overlap: total time both are in the epilogue / total epilogue time of A (items 1..7)
        ov = 0
        for i in range(1, 8):
            for j in range(0, 9):
                ov += max(0, min(A[i, 1], B[j, 1]) - max(A[i, 0], B[j, 0]))
        overl.append(ov / max(1, (A[1:8, 1] - A[1:8, 0]).sum()))
        if shown < 12:
            shown += 1
            print(key, "wgs", ws, "HW slots", int(a[ws[0], 0]) & 15, int(a[ws[1], 0]) & 15, "period", int(P), "epi", int((A[1:8, 1] - A[1:8, 0]).mean()),
                  "phase of B in A's period", np.round(ph, 2), "epilogue overlap frac", round(overl[-1], 2))
    n2 = 1024 * 4 * 16
    buf2 = (C.c_ulonglong * n2)()
    f2 = C.CDLL(os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so")).sinddm_debug_w2_seg
    f2.argtypes = [C.c_void_p, C.c_int]
    assert f2(buf2, n2) == 0
    sg = np.array(buf2, dtype=np.uint64).reshape(1024, 4, 16).astype(np.int64)[used]
    names = ["start"] + [f"p{p_}:{n_}" for p_ in range(3) for n_ in ("writes+prefetch issued", "barrier1 passed", "LDS reads done(issued)", "barrier2 passed", "math+stores issued")]
    d = np.diff(sg, axis=2)
    print("epilogue segments of item 3 (ticks; mean over workgroups of the per-wave mean / max over the 4 waves):")
    for i in range(15):
        print(f"   {names[i + 1]:32s} {d[:, :, i].mean():8.0f} {d[:, :, i].max(axis=1).mean():8.0f}")
    print("   total", (sg[:, :, 15] - sg[:, :, 0]).mean(), " wave start spread", (sg[:, :, 0].max(axis=1) - sg[:, :, 0].min(axis=1)).mean())
    overl = np.array(overl)
    print("epilogue overlap fraction over CUs: mean", overl.mean(), "quartiles", np.percentile(overl, [25, 50, 75]))
finally:
    shutil.copy("/tmp/lib_keep.so", os.path.join(ROOT, "sinddm_amd", "libsinddm_hip.so"))
