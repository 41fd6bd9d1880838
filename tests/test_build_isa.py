"""The inline-asm contract of conv_wino4.h, checked on the compiler's own output (no GPU): the kernel addresses its 240
accumulator registers as AGPRs a0..a239 BY NUMBER, so the compiler must not allocate a single AGPR itself, must not
spill, and the kernel descriptor must reserve 240 AGPRs.  Round 4: the accumulators are never zeroed -- the first k-step
of a work item runs with C = 0 -- and each is read exactly once (by the epilogue's column transform).  Compiles a
one-kernel translation unit to assembly (~50 s)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

TU = """
#include "conv_wino4.h"
namespace sinddm {
ConvProfiler& conv_profiler() { static ConvProfiler p; return p; }
int touch(const ConvArgs& a, hipStream_t st) { return conv_wino4_launch(a, st); }
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
        inasm, outside, mfma, mfma0, reads = False, [], 0, 0, []
        for line in body:
            if "#ASMSTART" in line:
                inasm = True
            elif "#ASMEND" in line:
                inasm = False
            elif not line.strip().startswith(";"):
                if inasm and "v_mfma_f32_16x16x4_f32" in line:
                    mfma += 1
                    mfma0 += line.strip().endswith(", 0")
                if inasm and "v_accvgpr_read_b32" in line:
                    reads.append(int(re.search(r"\ba(\d+)\b", line).group(1)))
                assert "v_accvgpr_write" not in line, m.group(1)
                if not inasm and re.search(r"\ba\d+\b|a\[\d+:\d+\]|accvgpr|v_mfma", line):
                    outside.append(line.strip())
        assert not outside, (m.group(1), outside[:5])
        # the chunk body exists twice: the item's first chunk (its k-step 0 with C = 0: 60 MFMAs) and the loop body
        assert mfma == 480 and mfma0 == 60, (m.group(1), mfma, mfma0)
        assert sorted(reads) == list(range(240)), m.group(1)          # every accumulator is read once, none is zeroed
        assert not any("scratch_" in l for l in body), m.group(1)
        meta = s[end:end + 8000]
        assert re.search(r"NumAgprs:\s+240\b", meta), m.group(1)
        assert re.search(r"ScratchSize:\s+0\b", meta), m.group(1)
        assert int(re.search(r"NumVgprs:\s+(\d+)", meta).group(1)) <= 248, m.group(1)


WGRAD_TU = """
#include "conv_mfma.h"
#include "wgrad_wino.h"
#include "wgrad_wh.h"
namespace sinddm {
ConvProfiler& conv_profiler() { static ConvProfiler p; return p; }
void touch(const WwArgs& w, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_wino_wide_kernel, dim3(1), dim3(WW_THREADS), 0, st, w);
    hipLaunchKernelGGL(wgrad_wh_kernel, dim3(1), dim3(WW_THREADS), 0, st, w);
}
}
"""


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_wgrad_wide_kernel_reads_lds_with_plain_b64_and_dmas_16_bytes():
    """wgrad_wino.h's wide kernel relies on two things the compiler could silently undo: its operand reads must stay
    single ds_read_b64 (fused into ds_read2_b64 / ds_read2st64_b64 the 16-byte-slot LDS image is 2-way bank-conflicted),
    and the tile loop body of every (frequency, n-tile count) copy must be ONE basic block (60 / 40 / 20 MFMAs), i.e. free
    of branches -- the reads are scheduled a k-step ahead inside it."""
    from sinddm_amd import build
    tmp = tempfile.mkdtemp(prefix="wwisa")
    try:
        src = os.path.join(tmp, "t.hip")
        with open(src, "w") as f:
            f.write(WGRAD_TU)
        flags = [f for f in build.FLAGS if f not in ("-fPIC", "-shared")]
        subprocess.check_call([HIPCC, *flags, "-I", os.path.join(ROOT, "include"), "-I", build.CSRC,
                               "--cuda-device-only", "-S", src, "-o", os.path.join(tmp, "t.s")],
                              stderr=subprocess.DEVNULL)
        s = open(os.path.join(tmp, "t.s")).read()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    m = re.search(r"^(_ZN6sinddm22wgrad_wino_wide_kernel\w+):", s, re.M)
    assert m
    end = s.index(".Lfunc_end", m.start())
    body = [l.strip() for l in s[m.start():end].split("\n") if l.strip() and not l.strip().startswith(";")]
    assert not [l for l in body if l.startswith("ds_read2")], "fused LDS reads"
    assert sum(l.startswith("ds_read_b64") for l in body) > 2000
    assert any(l.startswith("buffer_load_dwordx4") and " lds" in l for l in body)
    assert not [l for l in body if l.startswith("buffer_load_dword ") and " lds" in l]
    assert sum(l.startswith("v_mfma_f32_16x16x4_f32") for l in body) == 16 * (60 + 40 + 20)
    # basic blocks by label: the 48 loop bodies hold 60, 40 or 20 MFMAs each, nothing else holds any
    counts, cur = [], 0
    for l in body:
        if re.match(r"^\.LBB\d+_\d+:", l):
            counts.append(cur)
            cur = 0
        elif l.startswith("v_mfma"):
            cur += 1
    counts.append(cur)
    held = sorted(c for c in counts if c)
    assert held == sorted([60] * 16 + [40] * 16 + [20] * 16), held
    meta = s[end:end + 8000]
    assert re.search(r"Occupancy:\s+4\b", meta)
    scratch = int(re.search(r"ScratchSize:\s+(\d+)", meta).group(1))
    assert scratch <= 64, scratch                  # (a few spilled address registers in the 4-term copies; none holds data)

    # the binary16 sibling (wgrad_wh.h): one k-step per tile -- 30 / 20 / 10 MFMAs of the binary16 shape in each of the 48 loop
    # bodies, operands read as ds_read_b128, the split's remainder as ONE v_fma_mix_f32 per value (16 per fragment pair: 8 / 7 / 6
    # fragments per tile), tiles by 16-byte LDS-DMA, four waves per SIMD
    m = re.search(r"^(_ZN6sinddm15wgrad_wh_kernel\w+):", s, re.M)
    assert m
    end = s.index(".Lfunc_end", m.start())
    body = [l.strip() for l in s[m.start():end].split("\n") if l.strip() and not l.strip().startswith(";")]
    assert sum(l.startswith("v_mfma_f32_16x16x32_f16") for l in body) == 16 * (30 + 20 + 10)
    assert not [l for l in body if l.startswith("v_mfma") and "16x16x32_f16" not in l]
    assert sum(l.startswith("v_fma_mix_f32") for l in body) == 16 * 4 * (8 + 7 + 6)
    assert not [l for l in body if l.startswith("v_cvt_f32_f16")], "the remainder went back to conversion + subtract"
    assert sum(l.startswith("ds_read_b128") for l in body) > 600 and not [l for l in body if l.startswith("ds_read2")]
    assert any(l.startswith("buffer_load_dwordx4") and " lds" in l for l in body)
    counts, cur = [], 0
    for l in body:
        if re.match(r"^\.LBB\d+_\d+:", l):
            counts.append(cur)
            cur = 0
        elif l.startswith("v_mfma"):
            cur += 1
    counts.append(cur)
    assert sorted(c for c in counts if c) == sorted([30] * 16 + [20] * 16 + [10] * 16)
    meta = s[end:end + 8000]
    assert re.search(r"Occupancy:\s+4\b", meta)
    assert int(re.search(r"ScratchSize:\s+(\d+)", meta).group(1)) <= 64


TU_WH = """
#include "conv_wh.h"
namespace sinddm {
ConvProfiler& conv_profiler() { static ConvProfiler p; return p; }
int touch_wh(const ConvArgs& a, hipStream_t st) { return conv_wh_launch(a, st); }
}
"""


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_binary16_kernels_fit_their_register_budget():
    """conv_wh runs two waves per SIMD (512-thread workgroups): 256 registers per lane, and sits within a
    handful of that limit (160 accumulators + operand rings).  A spill would not fail any parity test -- it would put
    accumulators in scratch memory and cost tens of percent -- so the compiler's own resource summary is checked: no scratch,
    no VGPR spill, the MFMA counts of the main loops (conv_wh: 4 frequencies x 5 column tiles x 4 = 80 per chunk in the
    multiplying instantiation, none in the service one's loop)."""
    from sinddm_amd import build
    tmp = tempfile.mkdtemp(prefix="whisa")
    try:
        src = os.path.join(tmp, "t.hip")
        with open(src, "w") as f:
            f.write(TU_WH)
        flags = [f for f in build.FLAGS if f not in ("-fPIC", "-shared")]
        subprocess.check_call([HIPCC, *flags, "-I", os.path.join(ROOT, "include"), "-I", build.CSRC,
                               "--cuda-device-only", "-S", src, "-o", os.path.join(tmp, "t.s")],
                              stderr=subprocess.DEVNULL)
        s = open(os.path.join(tmp, "t.s")).read()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    meta = {}
    for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size: +\d+", s, re.S):
        name = re.search(r"\.name: +(\S+)", blk).group(1)
        meta[name] = {k: int(re.search(r"\.%s: +(\d+)" % k, blk).group(1))
                      for k in ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "max_flat_workgroup_size")}
    wh = [k for k in meta if "conv_wh_kernel" in k]
    assert len(wh) == 1 and not [k for k in meta if "conv_h2_kernel" in k], sorted(meta)     # (conv_h2: archived in tools/variants since round 6)
    for k in wh:
        m = meta[k]
        assert m["max_flat_workgroup_size"] == 512 and m["vgpr_count"] <= 256, (k, m)
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, (k, m)
    body = s[s.index(wh[0] + ":"):]
    body = body[:body.index(".Lfunc_end")]
    assert body.count("v_mfma_f32_16x16x32_f16") == 80
    # the input transform's patch windows come from DPP row shifts, its lo pieces from v_fma_mixlo/hi_f16 (two copies of the
    # transform: the item's first chunk and the chunk loop)
    assert body.count("v_mov_b32_dpp") == 64 and body.count("v_fma_mixlo_f16") == 96 and body.count("v_fma_mixhi_f16") == 96
