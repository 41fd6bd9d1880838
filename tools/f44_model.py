#!/usr/bin/env python3
"""Numpy model of the F(4x4,3x3) kernel planned as conv_wino6.h -- the arithmetic in the exact order the device will use,
checked against a float64 direct convolution (CPU only).

Work split: a wave (a, b), a, b in {0, 1}, owns the 3 x 3 frequency block i in {3a..3a+2}, j in {3b..3b+2} of the 6 x 6
grid: nine of the 36 frequency GEMMs, all five m-tiles, one n-tile of sixteen 4x4 tiles (an 8 x 32 pixel item).
  input transform:  rows i of B^T d over patch rows a..a+4, then columns j over patch columns b..b+4 (25 of the 36 patch
                    values; 6 + 6 operations per row / column triple)
  products:         M[i][j] += U[i][j] (co x ci) V[i][j] (ci x tile), fp32 accumulation in k-steps of 4 channels
  output transform: writer half  T_ab[p][jj] = sum_{i in block} A^T[p][i] M[i][3b+jj]     (4 x 3 per tile and channel)
                    reader half  T[p][j] = T_0b[p][jj] + T_1b[p][jj];  Y[p][q] = sum_j T[p][j] A^T[q][j]
"""
import numpy as np

f = np.float32
BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
               [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
              [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], float)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)


def bt3(d, a):
    """rows 3a..3a+2 of B^T applied along axis 0 of d (6 x ...), device operation order, fp32."""
    d = d.astype(f)
    if a == 0:
        s = (d[4] - f(4) * d[2]).astype(f)
        t = (d[3] - f(4) * d[1]).astype(f)
        r0 = (f(4) * d[0] + (d[4] - f(5) * d[2]).astype(f)).astype(f)
        return np.stack([r0, (s + t).astype(f), (s - t).astype(f)])
    u = (d[4] - d[2]).astype(f)
    w = (d[3] - d[1]).astype(f)
    r5 = (f(4) * d[1] + (d[5] - f(5) * d[3]).astype(f)).astype(f)
    return np.stack([(u + f(2) * w).astype(f), (u - f(2) * w).astype(f), r5])


def at_rows(m, a):
    """sum_{i in block a} A^T[p][i] m[i]  for p = 0..3 (m: 3 x ...), device order."""
    m = m.astype(f)
    if a == 0:
        s, d = (m[1] + m[2]).astype(f), (m[1] - m[2]).astype(f)
        return np.stack([(m[0] + s).astype(f), d, s, d])
    s, d = (m[0] + m[1]).astype(f), (m[0] - m[1]).astype(f)
    return np.stack([s, (f(2) * d).astype(f), (f(4) * s).astype(f), (f(8) * d + m[2]).astype(f)])


def at_cols(t):
    """Y[q] = sum_j t[j] A^T[q][j] over all six j (t: 6 x ...), reader order."""
    t = t.astype(f)
    s12, d12 = (t[1] + t[2]).astype(f), (t[1] - t[2]).astype(f)
    s34, d34 = (t[3] + t[4]).astype(f), (t[3] - t[4]).astype(f)
    return np.stack([((t[0] + s12).astype(f) + s34).astype(f), (d12 + f(2) * d34).astype(f),
                     (s12 + f(4) * s34).astype(f), ((d12 + f(8) * d34).astype(f) + t[5]).astype(f)])


def conv_f44(x, w):
    """x (C, H, W) with H - 2, W - 2 multiples of 4; w (K, C, 3, 3); returns (K, H - 2, W - 2) float32."""
    C, H, W = x.shape
    K = w.shape[0]
    U = np.einsum("ia,kcab,jb->ijkc", G, w.astype(np.float64), G).astype(f)        # offline, float64, rounded once
    out = np.zeros((K, H - 2, W - 2), f)
    for ty in range((H - 2) // 4):
        for tx in range((W - 2) // 4):
            d = x[:, ty * 4:ty * 4 + 6, tx * 4:tx * 4 + 6].astype(f)              # (C, 6, 6)
            T = np.zeros((2, 2, 4, 3, K), f)                                        # [a][b][p][jj][co]
            for a in range(2):
                v1 = bt3(np.moveaxis(d, 1, 0), a)                                   # (3, C, 6): rows i, columns still raw
                for b in range(2):
                    V = bt3(np.moveaxis(v1, 2, 0), b)                               # (3 jj, 3 ii, C)
                    M = np.zeros((3, 3, K), f)
                    for c0 in range(0, C, 4):                                       # MFMA k-loop, fp32 accumulation
                        for ii in range(3):
                            for jj in range(3):
                                M[ii, jj] += (U[3 * a + ii, 3 * b + jj][:, c0:c0 + 4].astype(np.float64)
                                              @ V[jj, ii, c0:c0 + 4].astype(np.float64)).astype(f)
                    T[a, b] = at_rows(M, a)                                         # (4 p, 3 jj, K)
            Tsum = (T[0] + T[1]).astype(f)                                          # [b][p][jj][co]
            t6 = np.concatenate([Tsum[0], Tsum[1]], axis=1)                         # (4 p, 6 j, K)
            Y = at_cols(np.moveaxis(t6, 1, 0))                                      # (4 q, 4 p, K)
            out[:, ty * 4:ty * 4 + 4, tx * 4:tx * 4 + 4] = np.transpose(Y, (2, 1, 0))
    return out


def direct64(x, w):
    C, H, W = x.shape
    out = np.zeros((w.shape[0], H - 2, W - 2))
    for a in range(3):
        for b in range(3):
            out += np.einsum("kc,chw->khw", w[:, :, a, b], x[:, a:a + H - 2, b:b + W - 2])
    return out


def main():
    rng = np.random.default_rng(0)
    rel = lambda y, r: float(np.linalg.norm(y.astype(np.float64) - r) / np.linalg.norm(r))
    x = rng.standard_normal((16, 10, 14))
    w = rng.uniform(-1, 1, (8, 16, 3, 3)) / 12
    print("exactness of the split (small case):", f"{rel(conv_f44(x, w), direct64(x, w)):.2e}")
    C = K = 160
    x = rng.standard_normal((C, 2 + 12, 2 + 16))
    w = rng.uniform(-1, 1, (K, C, 3, 3)) / np.sqrt(C * 9)
    print("C_in = C_out = 160, unit-variance input, default-init weights:", f"{rel(conv_f44(x, w), direct64(x, w)):.2e}")


if __name__ == "__main__":
    main()
