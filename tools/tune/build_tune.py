"""Build tools/libmmada_tune.so: the GEMM tuning variants (tools/tune/gemm_var.hip) + a copy of the production GEMM
for side-by-side timing.  Not part of the product; used by tools/gemm_sweep.py only."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
LIB = os.path.join(os.path.dirname(HERE), "libmmada_tune.so")
CSRC = os.path.join(ROOT, "mmada_parallel_amd", "csrc")


def build(force=False):
    from mmada_parallel_amd.build import _hipcc

    srcs = [os.path.join(HERE, "gemm_var.hip"), os.path.join(HERE, "gemm8.hip"), os.path.join(HERE, "tune_api.hip"), os.path.join(CSRC, "gemm.hip"), os.path.join(CSRC, "gemm8.hip")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(s) < os.path.getmtime(LIB) for s in srcs):
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DMMADA_TUNE", "-I", CSRC] + srcs + ["-o", LIB]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
