"""The inline-asm contract of conv_wino4.h, checked on the compiler's own output (no GPU): the kernel addresses its 240
accumulator registers as AGPRs a0..a239 BY NUMBER, so the compiler must not allocate a single AGPR itself, must not
spill, and the kernel descriptor must reserve 240 AGPRs.  Compiles a one-kernel translation unit to assembly (~40 s)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

TU = """
#include "conv_wino5.h"
namespace sinddm {
ConvProfiler& conv_profiler() { static ConvProfiler p; return p; }
int touch(const ConvArgs& a, hipStream_t st) { return conv_wino4_launch(a, st); }
int touch5(const ConvArgs& a, hipStream_t st) { return conv_wino5_launch(a, st); }
}
"""


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_conv_wino4_owns_the_agprs():
    from sinddm_amd import build
    tmp = tempfile.mkdtemp(prefix="w4isa")
    try:
        src = os.path.join(tmp, "t.hip")
        with open(src, "w") as f:
            f.write(TU)
        flags = [f for f in build.FLAGS if f not in ("-fPIC", "-shared")]
        subprocess.check_call([HIPCC, *flags, "-I", os.path.join(ROOT, "include"), "-I", build.CSRC,
                               "--cuda-device-only", "-S", src, "-o", os.path.join(tmp, "t.s")],
                              stderr=subprocess.DEVNULL)
        s = open(os.path.join(tmp, "t.s")).read()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kernels = list(re.finditer(r"^(_ZN6sinddm17conv_wino4_kernel\w+):", s, re.M))
    assert len(kernels) == 9                      # ACT 0..2 x EDGE 0..2
    for m in kernels:
        end = s.index(".Lfunc_end", m.start())
        body = s[m.start():end].split("\n")
        inasm, outside, mfma = False, [], 0
        for line in body:
            if "#ASMSTART" in line:
                inasm = True
            elif "#ASMEND" in line:
                inasm = False
            elif not line.strip().startswith(";"):
                if inasm and "v_mfma_f32_16x16x4_f32" in line:
                    mfma += 1
                if not inasm and re.search(r"\ba\d+\b|a\[\d+:\d+\]|accvgpr|v_mfma", line):
                    outside.append(line.strip())
        assert not outside, (m.group(1), outside[:5])
        assert mfma == 240, (m.group(1), mfma)    # one 16-channel chunk = 4 k-steps x 60 MFMAs, nothing duplicated
        assert not any("scratch_" in l for l in body), m.group(1)
        meta = s[end:end + 8000]
        assert re.search(r"NumAgprs:\s+240\b", meta), m.group(1)
        assert re.search(r"ScratchSize:\s+0\b", meta), m.group(1)
        assert int(re.search(r"NumVgprs:\s+(\d+)", meta).group(1)) <= 248, m.group(1)
    # conv_wino5.h: two waves per SIMD, accumulators a8..a127 by number; the compiler may park values in a0..a7 only
    kernels = list(re.finditer(r"^(_ZN6sinddm17conv_wino5_kernel\w+):", s, re.M))
    assert len(kernels) == 9
    for m in kernels:
        end = s.index(".Lfunc_end", m.start())
        inasm, own, mine, mfma = False, set(), set(), 0
        for line in s[m.start():end].split("\n"):
            if "#ASMSTART" in line:
                inasm = True
            elif "#ASMEND" in line:
                inasm = False
            elif not line.strip().startswith(";"):
                regs = [int(x) for x in re.findall(r"\ba(\d+)\b", line)] + [int(x) for x in re.findall(r"a\[(\d+):", line)]
                (mine if inasm else own).update(regs)
                mfma += inasm and "v_mfma_f32_16x16x4_f32" in line
                assert inasm or "v_mfma" not in line, m.group(1)
                assert "scratch_" not in line, m.group(1)
        assert mfma == 240, (m.group(1), mfma)       # two halves x 4 k-steps x 30 MFMAs
        assert min(mine) >= 8 and max(mine) <= 127, (m.group(1), min(mine), max(mine))
        assert not own or max(own) < 8, (m.group(1), sorted(own))
        meta = s[end:end + 8000]
        assert re.search(r"NumAgprs:\s+128\b", meta) and re.search(r"ScratchSize:\s+0\b", meta), m.group(1)
        assert re.search(r"Occupancy:\s+2\b", meta), m.group(1)
