#!/usr/bin/env python3
"""Numpy model of the two binary16 hi/lo 3x3 kernels (conv_h2.h, conv_wh.h) -- the arithmetic in the order the device uses,
on network-shaped data -- against float64 and against an fp32 evaluation of the same convolution.  VERDICT r4 item 1(a):
establish on the CPU that the scheme is not narrower than fp32 before (and beside) the GPU gate of tests/test_gpu_h2.py.

  split(a): s = 2^e a (e: max |a| of the tensor / channel -> [2^13, 2^14) direct, [2^10, 2^11) Winograd input),
            hi = rn16(s), lo = rn16(s - hi)                                   (binary16 = numpy float16, round to nearest even)
  direct  : sum over (tap, ci) of  hi*hi + hi*lo + lo*hi       -- per MFMA (16 ci of one tap) the exact products are summed
            and added to the fp32 accumulator with ONE rounding (the model's assumption about v_mfma_f32_32x32x16_f16; the GPU
            test measures the real thing)
  winograd: V = B^T d B in fp32 (vertical differences first, then the six horizontal combinations, as the kernel), U = G g G^T in
            float64 rounded once to fp32; both split; per frequency  sum over ci of all four terms, 16 ci per MFMA; inverse
            transform A^T M A in fp32
Run: python tools/h2_model.py     (prints rel-L2 errors vs float64)
"""
import numpy as np

F16, F32, F64 = np.float16, np.float32, np.float64
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], F64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], F64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], F64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], F64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], F64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], F64)


def shift_for(amax, target):
    if amax == 0 or not np.isfinite(amax):
        return 0
    return int(target - np.floor(np.log2(amax)))


def split(a32, shift):
    """fp32 array -> (hi, lo) binary16 pieces of a * 2^shift, as float64 arrays of the binary16 values."""
    s = (a32.astype(F32) * F32(2.0 ** shift)).astype(F32)
    hi = s.astype(F16)
    lo = (s - hi.astype(F32)).astype(F16)
    return hi.astype(F64), lo.astype(F64)


def mfma_accumulate(terms, acc32):
    """One MFMA: exact sum of the products (float64 holds 22-bit products of binary16 exactly) + accumulator, one fp32 rounding."""
    return (acc32.astype(F64) + terms).astype(F32)


def conv_direct64(x, w):
    C, H, W = x.shape
    xp = np.zeros((C, H + 2, W + 2), F64)
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W), F64)
    for a in range(3):
        for b in range(3):
            out += np.einsum("oc,chw->ohw", w[:, :, a, b].astype(F64), xp[:, a:a + H, b:b + W])
    return out


def conv_fp32(x, w):
    """fp32 FMA chain over (ci, tap) -- what an fp32 kernel does (order: ci outer, taps inner)."""
    C, H, W = x.shape
    xp = np.zeros((C, H + 2, W + 2), F32)
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W), F32)
    for c in range(C):
        for a in range(3):
            for b in range(3):
                prod = (w[:, c, a, b].astype(F64)[:, None, None] * xp[c, a:a + H, b:b + W].astype(F64)[None])
                out = (out.astype(F64) + prod).astype(F32)          # fused multiply-add: one rounding
    return out


def conv_h2(x, w):
    """conv_h2.h: direct, three terms, 16 ci per MFMA, per-channel weight scale, per-tensor activation scale."""
    C, H, W = x.shape
    O = w.shape[0]
    xs = shift_for(np.abs(x).max(), 13)
    xh, xl = split(x, xs)
    xph = np.zeros((C, H + 2, W + 2), F64); xph[:, 1:-1, 1:-1] = xh
    xpl = np.zeros((C, H + 2, W + 2), F64); xpl[:, 1:-1, 1:-1] = xl
    ws = np.array([shift_for(np.abs(w[o]).max(), 13) for o in range(O)])
    wh = np.zeros(w.shape, F64); wl = np.zeros(w.shape, F64)
    for o in range(O):
        wh[o], wl[o] = split(w[o], int(ws[o]))
    acc = np.zeros((O, H, W), F32)
    for c0 in range(0, C, 16):
        for a in range(3):
            for b in range(3):
                sl = slice(c0, c0 + 16)
                ph, pl = xph[sl, a:a + H, b:b + W], xpl[sl, a:a + H, b:b + W]
                for (u, v) in ((wh, pl), (wl, ph), (wh, ph)):          # lo terms first, as the kernel issues them
                    acc = mfma_accumulate(np.einsum("oc,chw->ohw", u[:, sl, a, b], v), acc)
    return acc.astype(F64) * (2.0 ** -(xs + ws))[:, None, None]


def conv_wh(x, w):
    """conv_wh.h: Winograd F(2x4,3x3), frequency GEMMs on binary16 hi/lo pieces, all four terms."""
    C, H, W = x.shape
    O = w.shape[0]
    assert H % 2 == 0 and W % 4 == 0 and C % 16 == 0
    U = np.einsum("ia,ocab,jb->ocij", G2, w.astype(F64), G4).astype(F32)              # float64 -> fp32 once
    ws = np.array([shift_for(np.abs(U[o]).max(), 13) for o in range(O)])
    Uh = np.zeros(U.shape, F64); Ul = np.zeros(U.shape, F64)
    for o in range(O):
        Uh[o], Ul[o] = split(U[o], int(ws[o]))
    xs = shift_for(np.abs(x).max(), 10)
    xp = np.zeros((C, H + 2, W + 2), F32); xp[:, 1:-1, 1:-1] = x
    th, tw = H // 2, W // 4
    # patches (C, th, tw, 4, 6) -> V (fp32: vertical +-1 combinations, then horizontal)
    d = np.stack([np.stack([xp[:, 2 * i:2 * i + 4, 4 * j:4 * j + 6] for j in range(tw)], 1) for i in range(th)], 1)
    r = np.einsum("ir,ctura->ctuia", BT2, d.astype(F64)).astype(F32)
    V = np.einsum("ctuia,ja->ctuij", r.astype(F64), BT4).astype(F32)                  # (each entry: <= 5 fp32 operations; modelled as one rounding)
    Vh, Vl = split(V, xs)
    M = np.zeros((O, th, tw, 4, 6), F32)
    for c0 in range(0, C, 16):
        sl = slice(c0, c0 + 16)
        terms = (np.einsum("ocij,ctuij->otuij", Uh[:, sl], Vl[sl]) + np.einsum("ocij,ctuij->otuij", Ul[:, sl], Vl[sl]))
        M = mfma_accumulate(terms, M)                                                  # MFMA 1: [V_lo | V_lo] x [U_hi | U_lo]
        terms = (np.einsum("ocij,ctuij->otuij", Uh[:, sl], Vh[sl]) + np.einsum("ocij,ctuij->otuij", Ul[:, sl], Vh[sl]))
        M = mfma_accumulate(terms, M)                                                  # MFMA 2: [V_hi | V_hi] x [U_hi | U_lo]
    q = np.einsum("pi,otuij->otupj", AT2, M.astype(F64)).astype(F32)
    y = np.einsum("otupj,qj->otupq", q.astype(F64), AT4).astype(F32)
    y = y.astype(F64) * (2.0 ** -(xs + ws))[:, None, None, None, None]
    return y.transpose(0, 1, 3, 2, 4).reshape(O, H, W)


def rel(a, b):
    return float(np.linalg.norm(a.astype(F64) - b) / np.linalg.norm(b))


def network_like(C, O, H, W, seed=0):
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((C, H, W)) * 1.3
    from math import erf
    x = (0.5 * g * (1 + np.vectorize(erf)(g / np.sqrt(2)))).astype(F32)          # GELU of a Gaussian: what conv2 of a block reads
    w = (rng.standard_normal((O, C, 3, 3)) / np.sqrt(9 * C)).astype(F32)
    return x, w


if __name__ == "__main__":
    for (C, O, H, W) in ((32, 16, 8, 16), (160, 32, 8, 16)):
        x, w = network_like(C, O, H, W)
        ref = conv_direct64(x, w)
        print(f"C_in={C} C_out={O} {H}x{W}:  fp32 FMA chain {rel(conv_fp32(x, w), ref):.3e}   binary16 direct (3 terms) {rel(conv_h2(x, w), ref):.3e}"
              f"   binary16 Winograd F(2x4) (4 terms) {rel(conv_wh(x, w), ref):.3e}")
