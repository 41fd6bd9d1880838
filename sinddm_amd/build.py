"""Builds libsinddm_hip.so (gfx950) in-tree with hipcc.  `python -m sinddm_amd.build [--force]`.

The built library is git-ignored but travels to the GPU box with the source snapshot, so a stale binary is a real
hazard: every build writes the SHA-256 of its inputs (all of csrc/, include/ and the compiler flags) next to the
library, `needs_build()` / `verify()` compare it with the current sources, and `_lib.load()` refuses a library whose
stamp does not match the sources it sits beside.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(HERE, "..", "include")
LIB = os.path.join(HERE, "libsinddm_hip.so")
STAMP = LIB + ".sha256"
SOURCES = ["sinddm_fwd.hip", "sinddm_bwd.hip"]
# -amdgpu-spill-vgpr-to-agpr=0: conv_wino4.h owns the AGPRs a0..a239 by number (inline asm); a VGPR spill must never land there
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0",
         "-fPIC", "-shared"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _inputs():
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hip"))]
    files += [os.path.join(INCLUDE, f) for f in sorted(os.listdir(INCLUDE)) if f.endswith(".h")]
    return files


def source_hash() -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in _inputs():
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def stamp() -> str:
    try:
        with open(STAMP) as f:
            return f.read().strip()
    except OSError:
        return ""


def needs_build() -> bool:
    return not os.path.exists(LIB) or stamp() != source_hash()


def verify() -> None:
    """Raise if the library is missing or was built from other sources than the ones beside it.  A deployment that ships
    the library without csrc/ (nothing to compare with) is accepted as it is."""
    if not os.path.exists(LIB):
        raise RuntimeError(f"{LIB} is missing: run `python -m sinddm_amd.build`")
    if not os.path.isdir(CSRC) or not os.path.isdir(INCLUDE):
        return
    try:
        current = source_hash()
    except OSError as e:
        raise RuntimeError(f"cannot read the sources beside {LIB}: {e}") from None
    if stamp() != current:
        raise RuntimeError(f"{LIB} is stale (csrc/ or include/ changed since it was built): run `python -m sinddm_amd.build`")


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    if os.path.exists(STAMP):
        os.remove(STAMP)
    # the two translation units compile side by side (the backward one carries 96 specialised copies of the
    # weight-gradient loop: ~100 s on its own), then one link
    import tempfile
    cflags = [f for f in FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="sinddm_build") as tmp:
        procs, objs = [], []
        for src in SOURCES:
            obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
            cmd = [_hipcc(), *cflags, "-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print("[sinddm_amd.build]", " ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, stderr=None if verbose else subprocess.DEVNULL, start_new_session=True)))
            objs.append(obj)
        failed = None
        try:
            for cmd, pr in procs:
                if pr.wait() != 0 and failed is None:
                    failed = (pr.returncode, cmd)
                    break
        finally:
            # never leave a compiler writing into the temporary directory that is about to be deleted
            for _, pr in procs:
                if pr.poll() is None:
                    try:                     # (hipcc is a driver: its clang children live in the same process group)
                        os.killpg(pr.pid, 9)
                    except OSError:
                        pr.kill()
                pr.wait()
        if failed:
            raise subprocess.CalledProcessError(*failed)
        link = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", LIB]
        if verbose:
            print("[sinddm_amd.build]", " ".join(link), flush=True)
        subprocess.check_call(link)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
