#!/usr/bin/env python3
"""The rational GELU of csrc/common.h (gelu_erf2): coefficients of Eigen's / XLA's fp32 erf (x P(x^2) / Q(x^2), |x| <= 4)
rescaled to the GELU argument v = x sqrt 2 and to Q(0) = 1, and the accuracy of the exact fp32 operation sequence
against float64 (no GPU).  Prints the constants as they stand in the source and the error figures quoted there."""
import numpy as np
from scipy.special import erf

AL = [-1.60960333262415e-02, -2.95459980854025e-03, -7.34990630326855e-04, -5.69250639462346e-05,
      -2.10102402082508e-06, 2.77068142495902e-08, -2.72614225801306e-10]          # x^1, x^3, .. x^13
BE = [-1.42647390514189e-02, -7.37332916720468e-03, -1.68282697438203e-03, -2.13374055278905e-04,
      -1.45660718464996e-05]                                                         # x^0, x^2, .. x^8
k = 1 / np.sqrt(2.0)
A = [a * k ** (2 * i + 1) / BE[0] for i, a in enumerate(AL)]
B = [b * k ** (2 * i) / BE[0] for i, b in enumerate(BE)]
f = np.float32
print("P:", ", ".join(f"{f(a):.9e}f" for a in A))
print("Q:", ", ".join(f"{f(b):.9e}f" for b in B))
print("clamp:", f(4 * np.sqrt(2)))


def gelu2(v):
    v = v.astype(f)
    c = np.clip(v, f(-5.656854249), f(5.656854249))
    s = (c * c).astype(f)
    p = f(A[6])
    for a in A[5::-1]:
        p = (p.astype(np.float64) * s + f(a)).astype(f)       # fma: one rounding
    q = f(B[4])
    for b in B[3::-1]:
        q = (q.astype(np.float64) * s + f(b)).astype(f)
    n = (p.astype(np.float64) * c + q).astype(f)
    r = (f(1) / q).astype(f)
    h = (f(0.5) * v).astype(f)
    return ((h * n).astype(f) * r).astype(f)


v = np.linspace(-12, 12, 3000001)
g = gelu2(v).astype(np.float64)
gx = 0.5 * v * (1 + erf(v / np.sqrt(2)))
e = np.abs(g - gx)
print(f"max abs error {e.max():.3e} at v = {v[e.argmax()]:.3f}; inside |v| < 4: {e[np.abs(v) < 4].max():.3e}; "
      f"rel-L2 over [-12, 12]: {np.linalg.norm(g - gx) / np.linalg.norm(gx):.3e}")
