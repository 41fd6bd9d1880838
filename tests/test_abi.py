"""CPU: the C-ABI library loads and exports every symbol include/sinddm_hip.h declares; the
introspection entry points (no GPU needed) agree with the Python-side layout."""
import os
import re

import numpy as np

from sinddm_amd import _lib
from sinddm_amd.synth import net_param_shapes

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    """Every function include/*.h declares (the boundary header and the debug-hook header)."""
    syms = set()
    for name in sorted(os.listdir(os.path.join(REPO, "include"))):
        if not name.endswith(".h"):
            continue
        txt = open(os.path.join(REPO, "include", name)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        syms |= set(re.findall(r"\b(sinddm_[a-z0-9_]+)\s*\(", txt))
    return sorted(syms)


def test_boundary_header_has_no_global_state_hooks():
    """The measurement hooks (process-global event table) are declared in sinddm_hip_debug.h only, and the library
    sources read no environment variable."""
    txt = open(os.path.join(REPO, "include", "sinddm_hip.h")).read()
    assert "sinddm_prof_" not in re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    csrc = os.path.join(REPO, "sinddm_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".h", ".hip")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_header_and_binding_agree():
    hdr = _header_symbols()
    assert hdr, "no declarations parsed"
    assert sorted(_lib.ABI_SYMBOLS) == hdr
    # one ABI version everywhere: the header's #define, the binding's constant (which __graft_entry__.build() checks the
    # built library against) and the option bit of `dim`
    import re
    txt = open(os.path.join(REPO, "include", "sinddm_hip.h")).read()
    assert int(re.search(r"#define SINDDM_ABI_VERSION (\d+)", txt).group(1)) == _lib.ABI_VERSION
    assert int(re.search(r"#define SINDDM_DIM_FP32_CONVS (0x[0-9a-fA-F]+)", txt).group(1), 16) == _lib.DIM_FP32_CONVS
    assert "_lib.ABI_VERSION" in open(os.path.join(REPO, "__graft_entry__.py")).read()


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _header_symbols():
        assert hasattr(lib, name), name
    assert lib.sinddm_abi_version() == _lib.ABI_VERSION == 3
    assert not _lib.missing_symbols()


def test_param_layout_matches_reference_key_order():
    lib = _lib.load()
    for dim in (16, 32, 160):
        shapes = net_param_shapes(dim)
        assert lib.sinddm_param_tensors(dim) == len(shapes) == 52
        off = 0
        for i, (k, s) in enumerate(shapes.items()):
            assert lib.sinddm_param_offset(dim, i) == off, (dim, k)
            off += int(np.prod(s))
        assert lib.sinddm_param_count(dim) == off
    assert lib.sinddm_param_count(160) == 1106772          # SURVEY.md 2.1 [probe]
    assert lib.sinddm_param_count(3) == -1                 # odd dim rejected
    assert lib.sinddm_packed_count(160) > 0 and lib.sinddm_packed_bwd_count(160) > 0


def test_workspace_sizes_and_error_codes():
    lib = _lib.load()
    a = lib.sinddm_workspace_bytes(160, 1, 48, 64)
    b = lib.sinddm_workspace_bytes(160, 2, 48, 64)
    assert 0 < a < b
    assert lib.sinddm_train_workspace_bytes(160, 1, 48, 64) > a
    assert lib.sinddm_workspace_bytes(160, 0, 48, 64) == 0
    # argument validation happens before any device work -> safe without a GPU
    assert lib.sinddm_pack_weights(None, None, 160, None) == -1
    assert lib.sinddm_net_forward(None, None, None, None, 0, 0.0, None, 160, 1, 8, 8, None, 0, None) == -1
    assert lib.sinddm_reverse_step(None, None, None, None, None, None, 10, None) == -1
    assert lib.sinddm_q_sample(None, None, None, None, None, None, None, None, 0, 1, 10, None) == -1
    assert lib.sinddm_upsample_bilinear(None, None, 1, 2, 2, 4, 4, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.SinddmError as e:
        assert "no CPU" in str(e)
    else:
        raise AssertionError("load() must raise when the HIP library is missing")


def test_cpu_tensors_are_rejected():
    import torch
    from sinddm_amd.models import SinDDMNet
    net = SinDDMNet(dim=16, multiscale=True, device="cpu")
    try:
        net(torch.zeros(1, 3, 8, 8), torch.zeros(1, dtype=torch.long), scale=0)
    except _lib.SinddmError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("CPU forward must raise (no fallback path exists)")
