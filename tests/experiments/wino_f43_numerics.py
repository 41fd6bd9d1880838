#!/usr/bin/env python3
"""Numerical feasibility of Winograd F(4x4,3x3) for this network (NOT a test, not collected by pytest; lives under
tests/ because it drives the oracle).  Question for DESIGN.md section 8: F(4x4,3x3) needs 2.25 instead of 4 multiplies
per output (1.78x fewer MFMAs than the shipped F(2x2,3x3) kernel) -- does its fp32 rounding error fit the parity
budget (1e-5 rel-L2 per network evaluation in the tests, 1e-4 on a full sampling chain)?

Everything is emulated on the CPU in fp32 exactly as a kernel would do it: U = G g G^T (computed in float64, rounded
once -- the pack kernel can afford that), V = B^T d B in fp32, the channel contraction in fp32, Y = A^T M A in fp32.

    python tests/experiments/wino_f43_numerics.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sinddm_oracle as O          # noqa: E402
from sinddm_amd.synth import closed_form_state_dict, hash_randn, noise_key   # noqa: E402

MATS = {
    2: dict(
        BT=[[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],
        G=[[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]],
        AT=[[1, 1, 1, 0], [0, 1, -1, -1]]),
    4: dict(
        BT=[[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
            [0, 4, 0, -5, 0, 1]],
        G=[[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
           [0, 0, 1]],
        AT=[[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]),
}


def wino_conv(x, w, b, m):
    """3x3 conv, padding 1, as Winograd F(m x m, 3x3) with fp32 arithmetic."""
    M = MATS[m]
    BT = torch.tensor(M["BT"], dtype=torch.float32)
    AT = torch.tensor(M["AT"], dtype=torch.float32)
    G = torch.tensor(M["G"], dtype=torch.float64)
    a = m + 2
    Bn, C, H, W = x.shape
    K = w.shape[0]
    U = (G @ w.double() @ G.T).float()                       # [K][C][a][a], rounded once
    th, tw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, 1 + tw * m - W, 1, 1 + th * m - H))
    d = xp.unfold(2, a, m).unfold(3, a, m)                    # [B][C][th][tw][a][a]
    V = BT @ d @ BT.T                                         # fp32
    Mm = torch.einsum("kcij,bcyxij->bkyxij", U, V)            # fp32 contraction over channels
    Y = AT @ Mm @ AT.T                                        # [B][K][th][tw][m][m]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, K, th * m, tw * m)[:, :, :H, :W]
    return y + b.view(1, -1, 1, 1)


def patched(m):
    orig = F.conv2d

    def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if m and w.shape[-1] == 3 and groups == 1 and w.shape[1] >= 16 and padding == 1:
            if m < 0:        # "truth": the same conv evaluated in float64, rounded once
                return orig(x.double(), w.double(), b.double(), stride, padding, dilation, groups).float()
            return wino_conv(x, w, b, m)
        return orig(x, w, b, stride, padding, dilation, groups)
    return conv2d


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    torch.manual_seed(0)
    dim = 32
    sd = closed_form_state_dict(dim)
    # ---- one conv, network-like magnitudes
    x = torch.randn(2, 160, 48, 64)
    w = torch.randn(160, 160, 3, 3) * (1.0 / np.sqrt(160 * 9))
    bz = torch.zeros(160)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    print("single 160->160 conv, rel-L2 vs float64:  direct fp32 %.2e   F(2x2) %.2e   F(4x4) %.2e" % (
        rel(F.conv2d(x, w, None, padding=1), ref), rel(wino_conv(x, w, bz, 2), ref), rel(wino_conv(x, w, bz, 4), ref)))
    # ---- one network evaluation
    xin = torch.randn(2, 3, 40, 56)
    t = torch.tensor([37, 99])
    orig = F.conv2d
    res = {}
    for m in (-1, 0, 2, 4):
        O.F.conv2d = patched(m)
        try:
            res[m] = O.net_forward(sd, xin, t, 2.0)
        finally:
            O.F.conv2d = orig
    truth = res[-1]
    print("net_forward (dim=%d), rel-L2 vs float64 convs:  direct fp32 %.2e   F(2x2) %.2e   F(4x4) %.2e   (test budget 1e-5)" % (
        dim, rel(res[0], truth), rel(res[2], truth), rel(res[4], truth)))
    # ---- a sampling chain: T=100 at one scale from pure noise, same injected noise in every variant
    sched = O.make_schedule(100, 1, None)
    H, W = 32, 40
    outs = {}
    for m in (-1, 0, 2, 4):
        O.F.conv2d = patched(m)
        try:
            img = hash_randn((2, 3, H, W), noise_key("init", 0, 0))
            for tt in reversed(range(100)):
                z = hash_randn((2, 3, H, W), noise_key("step", 0, tt))
                img = O.p_sample(sched, sd, img, tt, 0, z, None)
            outs[m] = img
        finally:
            O.F.conv2d = orig
    print("100-step chain (scale 0), rel-L2 vs float64-conv chain:  direct fp32 %.2e   F(2x2) %.2e   F(4x4) %.2e   (chain budget 1e-4)" % (
        rel(outs[0], outs[-1]), rel(outs[2], outs[-1]), rel(outs[4], outs[-1])))


if __name__ == "__main__":
    main()
