#!/usr/bin/env python3
"""fp32 rounding error of the Winograd variants a next kernel generation could use, on data shaped like the network's
(C_in = 160 unit-variance activations, default-init weights), against a float64 direct convolution.  CPU only (numpy).

    python tools/winograd_numerics.py            # prints rel-L2 errors of one 3x3 conv output

F(m x n) = F(m,3) down the rows x F(n,3) along the columns; F(2,3) uses the points {0, 1, -1, inf}, F(4,3) {0, +-1, +-2, inf}
(Lavin & Gray).  Products are accumulated over C_in in float32 in the order an MFMA k-loop would
(blocks of 4 channels).  The shipped kernels are F(2x2) (conv_wino2.h) and F(2x4) (conv_wino3/4/5.h)."""
import numpy as np

rng = np.random.default_rng(0)


def mats(kind):
    if kind == "F2":
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
    elif kind == "F4":
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                       [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                      [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], float)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)
    return BT, G, AT


def wino(x, w, row, col, dt=np.float32):
    """x (C,H,W), w (K,C,3,3) -> (K,H-2,W-2) through F(row) x F(col) tiles, arithmetic in dt."""
    BTr, Gr, ATr = mats(row)
    BTc, Gc, ATc = mats(col)
    mr, mc = ATr.shape[0], ATc.shape[0]
    nr, nc = BTr.shape[0], BTc.shape[0]
    C, H, W = x.shape
    K = w.shape[0]
    U = np.einsum("ia,kcab,jb->ijkc", Gr, w.astype(np.float64), Gc).astype(dt)       # weights transformed offline
    th, tw = (H - 2) // mr, (W - 2) // mc
    out = np.zeros((K, th * mr, tw * mc), dt)
    BTr32, BTc32, ATr32, ATc32 = (m.astype(dt) for m in (BTr, BTc, ATr, ATc))
    for ty in range(th):
        for tx in range(tw):
            d = x[:, ty * mr:ty * mr + nr, tx * mc:tx * mc + nc].astype(dt)
            V = np.einsum("ia,cab->cib", BTr32, d).astype(dt)
            V = np.einsum("cib,jb->ijc", V, BTc32).astype(dt)
            M = np.zeros((nr, nc, K), dt)
            for c0 in range(0, C, 4):                                                   # MFMA-like fp32 k-loop
                M += np.einsum("ijkc,ijc->ijk", U[..., c0:c0 + 4], V[..., c0:c0 + 4]).astype(dt)
            Y = np.einsum("ai,ijk->ajk", ATr32, M).astype(dt)
            Y = np.einsum("ajk,bj->kab", Y, ATc32).astype(dt)
            out[:, ty * mr:(ty + 1) * mr, tx * mc:(tx + 1) * mc] = Y
    return out


def direct(x, w, dt):
    C, H, W = x.shape
    out = np.zeros((w.shape[0], H - 2, W - 2), dt)
    xs, ws = x.astype(dt), w.astype(dt)
    for c0 in range(0, C, 4):
        for a in range(3):
            for b in range(3):
                out += np.einsum("kc,chw->khw", ws[:, c0:c0 + 4, a, b], xs[c0:c0 + 4, a:a + H - 2, b:b + W - 2]).astype(dt)
    return out


def main():
    C = K = 160
    H, W = 2 + 24, 2 + 24                  # 24 is a multiple of 2 and 4
    x = rng.standard_normal((C, H, W))
    w = rng.uniform(-1, 1, (K, C, 3, 3)) / np.sqrt(C * 9)
    ref = direct(x, w, np.float64)
    rel = lambda y: float(np.linalg.norm(y.astype(np.float64) - ref[:, :y.shape[1], :y.shape[2]]) / np.linalg.norm(ref[:, :y.shape[1], :y.shape[2]]))
    print(f"direct fp32                     {rel(direct(x, w, np.float32)):.2e}")
    for name, r, c, mult in (("F(2x2)  shipped, small launches", "F2", "F2", 16 / 4), ("F(2x4)  shipped, dominant kernel", "F2", "F4", 24 / 8),
                             ("F(4x4)  points 0,+-1,+-2,inf    ", "F4", "F4", 36 / 16)):
        print(f"{name} {rel(wino(x, w, r, c)):.2e}   ({mult:.2f} multiplies per output; float64 check {rel(wino(x, w, r, c, np.float64)):.1e})")


if __name__ == "__main__":
    main()
