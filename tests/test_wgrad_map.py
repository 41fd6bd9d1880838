"""Host logic of the weight-gradient launch: the workgroup -> (slab, pixel split) table of wgrad_wino_kernel
(sinddm_amd/csrc/wgrad_wino.h, ww_build_map).  No device call: the table is built on the host.

The kernel computes autograd's 3x3 weight gradients (reference SinDDM/models.py:63,65 through functions.py:97-102 /
trainer.py:200-209); a (80 co x <=48 ci) slab with n 16-channel ci tiles costs n MFMAs per k-step and tile, so the table
plus a slab-independent part,
and the table must hand every workgroup about the same cost."""
import ctypes as C

import pytest

from sinddm_amd import _lib

WW_CI, WW_CO = 48, 80


def _map(cin, cout, ntiles, ncu):
    lib = _lib.load()
    wg = (C.c_uint32 * 512)()
    sp = (C.c_int32 * 256)()
    n = lib.sinddm_debug_wgrad_map(cin, cout, ntiles, ncu, wg, 512, sp, 256)
    return n, list(wg[:max(n, 0)]), list(sp)


@pytest.mark.parametrize("cin,cout", [(80, 80), (80, 160), (160, 160), (160, 80), (16, 80), (32, 160), (320, 320)])
def test_every_split_once_and_balanced(cin, cout):
    ntiles = 32 * 47 * 16                 # C2 finest scale at batch 32: 186 x 248 in 4 x 16 tiles
    n, wg, sp = _map(cin, cout, ntiles, 256)
    ciblks = (cin + WW_CI - 1) // WW_CI
    slabs = (cout // WW_CO) * ciblks
    assert 0 < n <= 256 or slabs > 256
    seen = set()
    for e in wg:
        q, s = e >> 16, e & 0xFFFF
        assert q < slabs and s < sp[q]
        assert (q, s) not in seen
        seen.add((q, s))
    assert len(seen) == sum(sp[:slabs]) == n
    # cost of a workgroup = (n-tiles of its slab + 1 for the slab-independent part of a tile: DMA issue, barrier;
    # WW_SPLIT_C0 in wgrad_wino.h) * ceil(tiles / splits)
    work = []
    for q in range(slabs):
        nci = min(WW_CI, cin - (q % ciblks) * WW_CI)
        work.append(((nci + 15) // 16 + 1) * -(-ntiles // sp[q]))
    total = sum(w * sp[q] for q, w in enumerate(work))
    if slabs * 3 <= 256:
        assert max(work) * 256 <= 1.08 * total, (work, sp[:slabs])


def test_xcd_runs_are_contiguous_in_tile_order():
    """Workgroup ids id, id+8, id+16 ... (one XCD) walk a contiguous run of the tile-ordered workgroup list."""
    n, wg, sp = _map(160, 160, 24064, 256)
    key = lambda e: (e & 0xFFFF) / sp[e >> 16]
    for x in range(8):
        run = [key(e) for e in wg[x::8]]
        assert run == sorted(run)
    firsts = [key(wg[x]) for x in range(8)]
    assert firsts == sorted(firsts)


def test_small_launches_and_bad_arguments():
    n, wg, sp = _map(160, 160, 3, 256)          # fewer tiles than CUs: no slab gets more splits than tiles
    assert n > 0 and all(v <= 3 for v in sp[:8])
    assert _map(160, 100, 100, 256)[0] < 0      # Cout not a multiple of the 80-channel slab
    assert _map(48 * 70, 80, 100, 256)[0] > 0   # 70 slabs: one workgroup each at least
    assert _map(48 * 300, 80, 100, 256)[0] < 0  # more slabs than the table holds
    # the caller's buffers are sized by the caller: too small is reported, never overrun
    lib = _lib.load()
    wg = (C.c_uint32 * 8)()
    sp = (C.c_int32 * 256)()
    assert lib.sinddm_debug_wgrad_map(160, 160, 32 * 47 * 16, 256, wg, 8, sp, 256) == -1
    wg = (C.c_uint32 * 512)()
    sp = (C.c_int32 * 2)()
    assert lib.sinddm_debug_wgrad_map(160, 160, 32 * 47 * 16, 256, wg, 512, sp, 2) == -1
