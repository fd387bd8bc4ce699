"""Build libmmada_mi355x.so (the C-ABI of include/mmada_mi355x.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the resulting .so sits next to this file so it travels with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "libmmada_mi355x.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
SOURCES = ["gemm.hip", "gemm8.hip", "attention.hip", "attention64.hip", "elementwise.hip", "sampler.hip", "vq_decoder.hip", "graph.hip", "tp_comm.hip", "probe.hip", "api.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_epilogue.h", "handle.h", "attention.h", os.path.join("..", "..", "include", "mmada_mi355x.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 into one shared library; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
