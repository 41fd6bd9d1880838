"""CPU: the Winograd identities the conv kernels are built on, with the exact matrices (and the kernels' algebraic
shortcuts) written in sinddm_amd/csrc/conv_wino2.h / conv_wino3.h / sinddm_fwd.hip (pack kinds 3 and 4):

  F(2x2,3x3):  Y = A2^T [ (G2 g G2^T) (.) (B2^T d B2) ] A2          d: 4x4 patch,  Y: 2x2
  F(2x4,3x3):  Y = A2^T [ (G2 g G4^T) (.) (B2^T d B4) ] A4          d: 4x6 patch,  Y: 2x4

checked in exact rational arithmetic against the direct 3x3 correlation (reference SinDDM/models.py:63,65 = nn.Conv2d),
including the conventions the kernels use: vertical frequency row 2 is evaluated as d1 - d2 with U_2j stored negated,
the F(4,3) input transform through the (r4 - 4 r2) +- (r3 - 4 r1) / (r4 - r2) +- 2 (r3 - r1) pairs, and the output
transform through the s12 / d12 / s34 / d34 pairs."""
from fractions import Fraction as Fr

import numpy as np

G2 = [[1, 0, 0], [Fr(1, 2), Fr(1, 2), Fr(1, 2)], [Fr(1, 2), Fr(-1, 2), Fr(1, 2)], [0, 0, 1]]
G4 = [[Fr(1, 4), 0, 0], [Fr(-1, 6), Fr(-1, 6), Fr(-1, 6)], [Fr(-1, 6), Fr(1, 6), Fr(-1, 6)],
      [Fr(1, 24), Fr(1, 12), Fr(1, 6)], [Fr(1, 24), Fr(-1, 12), Fr(1, 6)], [0, 0, 1]]
B2T = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
B4T = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
       [0, 4, 0, -5, 0, 1]]
A2T = [[1, 1, 1, 0], [0, 1, -1, -1]]
A4T = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]


def F(m):
    return np.array([[Fr(v) for v in row] for row in m], dtype=object)


def direct(d, g, oh, ow):
    return np.array([[sum(d[y + a][x + b] * g[a][b] for a in range(3) for b in range(3)) for x in range(ow)]
                     for y in range(oh)], dtype=object)


def rand(shape, rng):
    return np.array(rng.integers(-9, 10, size=shape).tolist(), dtype=object).reshape(shape) * Fr(1)


def test_f2x2_identity():
    rng = np.random.default_rng(0)
    for _ in range(20):
        d, g = rand((4, 4), rng), rand((3, 3), rng)
        U = F(G2) @ g @ F(G2).T
        V = F(B2T) @ d @ F(B2T).T
        Y = F(A2T) @ (U * V) @ F(A2T).T
        assert (Y == direct(d, g, 2, 2)).all()


def test_f2x4_identity_with_kernel_conventions():
    rng = np.random.default_rng(1)
    for _ in range(20):
        d, g = rand((4, 6), rng), rand((3, 3), rng)
        U = F(G2) @ g @ F(G4).T                                   # 4 x 6 frequencies (pack kind 4)
        U[2] = -U[2]                                              # row 2 stored negated ...
        # ... because wave 2 evaluates d1 - d2:  vertical rows  0: d0 - d2   1: d1 + d2   2: d1 - d2   3: d1 - d3
        rows = [d[0] - d[2], d[1] + d[2], d[1] - d[2], d[1] - d[3]]
        V = np.empty((4, 6), dtype=object)
        for i, r in enumerate(rows):                              # the kernel's factorised F(4,3) input transform
            s24, s13 = r[4] - 4 * r[2], r[3] - 4 * r[1]
            u24, u13 = r[4] - r[2], 2 * (r[3] - r[1])
            V[i] = [4 * r[0] - 5 * r[2] + r[4], s24 + s13, s24 - s13, u24 + u13, u24 - u13, 4 * r[1] - 5 * r[3] + r[5]]
        assert (np.array([list(F(B4T) @ r) for r in rows], dtype=object) == V).all()
        M = U * V
        T = np.empty((4, 4), dtype=object)
        for i in range(4):                                        # column half of the output transform (in registers)
            m = M[i]
            s12, d12, s34, d34 = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
            T[i] = [m[0] + s12 + s34, d12 + 2 * d34, s12 + 4 * s34, d12 + 8 * d34 + m[5]]
        assert (np.array([list(F(A4T) @ M[i]) for i in range(4)], dtype=object) == T).all()
        Y = np.array([list(T[0] + T[1] + T[2]), list(T[1] - T[2] - T[3])], dtype=object)   # row half (through LDS)
        assert (Y == direct(d, g, 2, 4)).all()


def test_multiplies_per_output():
    assert 16 / 4 == 4 and 24 / 8 == 3 and 36 / 16 == 2.25        # F(2x2), F(2x4), F(4x4) against 9 of the direct form
