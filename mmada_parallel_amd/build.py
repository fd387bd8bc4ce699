"""Build libmmada_mi355x.so (the C-ABI of include/mmada_mi355x.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the resulting .so sits next to this file so it travels with the repo snapshot.
Translation units are compiled in parallel (one hipcc per .hip, objects under csrc/_obj/) and linked once.

`verify=True` (what __graft_entry__.build() passes) also checks the COMPILED code of the hand-scheduled kernels with
tools/isa_check.py (an assembly listing made with the SAME flags as the objects) and fails the build on a violation — removing
the objects, so that a later unverified build cannot link them: the LDS-DMA requests of gemm8.hip / attention.hip are
asm statements that write M0 themselves, which is only correct while hipcc keeps nothing live in M0 across them and while
nothing but LDS-DMA sits in the vector-memory queue in front of a counted wait — properties of the compiler's output, not
of the source, so a compiler upgrade must not be able to break them silently.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB_NAME = "libmmada_mi355x.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
SOURCES = ["gemm.hip", "gemm8.hip", "attention.hip", "elementwise.hip", "sampler.hip", "vq_decoder.hip", "graph.hip", "tp_comm.hip", "probe.hip", "api.hip"]
HEADERS = ["exports.map", "common.h", "kernels.h", "gemm_epilogue.h", "handle.h", "attention.h", os.path.join("..", "..", "include", "mmada_mi355x.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wno-unused-result"]
# per-unit flags.  attention: hipcc's SLP vectoriser pairs the fp32 row-sum adds and rescale multiplies into v_pk_*_f32, which
# cost more issue time beside MFMAs than the two scalar operations they replace (MI355X guide, per-instruction constants);
# without NaN semantics fmaxf() on MFMA results needs no canonicalising v_max x,x and nests into v_max3_f32
UNIT_FLAGS = {"attention.hip": ["-fno-slp-vectorize", "-fno-honor-nans"]}


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


def _run(cmd):
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)
    return proc.stdout


def compile_objects(out_dir: str, extra_flags=(), force: bool = False, verbose: bool = False, jobs: int = 0):
    """One object per translation unit, in parallel; only units older than their sources / headers are recompiled."""
    os.makedirs(out_dir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    todo, objs = [], []
    for src in SOURCES:
        obj = os.path.join(out_dir, src.replace(".hip", ".o"))
        objs.append(obj)
        sp = os.path.join(CSRC, src)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_t, os.path.getmtime(sp)):
            todo.append([_hipcc(), *FLAGS, *UNIT_FLAGS.get(src, []), *extra_flags, "-c", sp, "-o", obj])
    if verbose:
        for c in todo:
            print(" ".join(c))
    with ThreadPoolExecutor(max_workers=jobs or min(len(todo) or 1, os.cpu_count() or 4)) as ex:
        list(ex.map(_run, todo))
    return objs


def verify_isa() -> None:
    """tools/isa_check.py on freshly compiled device code: raises on any violation (see the module docstring)."""
    tools = os.path.join(ROOT, "tools")
    if not os.path.exists(os.path.join(tools, "isa_check.py")):
        print("build: tools/isa_check.py not found, compiled-code checks skipped", file=sys.stderr)
        return
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import isa_check

    with ThreadPoolExecutor(max_workers=2) as ex:
        asm8, asma = ex.map(isa_check.device_asm, ["gemm8.hip", "attention.hip"])
    errors = []
    errors += isa_check.check_gemm8(asm8)[1]          # counted waits, M0 contract, no spill, 16-byte epilogue stores
    errors += isa_check.check_attention(asma)[1]      # M0 contract, counted LDS waits, no spill
    if errors:
        raise RuntimeError("compiled-code checks failed (tools/isa_check.py):\n" + "\n".join(errors))


def build(force: bool = False, verbose: bool = False, verify: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 into one shared library; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    with ThreadPoolExecutor(max_workers=2) as ex:
        chk = ex.submit(verify_isa) if verify else None
        objs = compile_objects(OBJ, force=force, verbose=verbose)
        _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), *objs,
              "-o", LIB_PATH + ".tmp"])
        if chk is not None:
            try:
                chk.result()
            except Exception:
                # the checked code IS what these objects hold (same flags): do not leave them for a later unverified link
                for o in objs + [LIB_PATH + ".tmp"]:
                    if os.path.exists(o):
                        os.remove(o)
                raise
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, verify="--verify" in sys.argv))
