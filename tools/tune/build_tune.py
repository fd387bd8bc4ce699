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


PRODUCT_TUNE_LIB = os.path.join(os.path.dirname(HERE), "libmmada_mi355x_tune.so")


def build_product_tune(force=False):
    """The whole product library compiled with -DMMADA_TUNE (extra GEMM configurations, diagnostic attention variants):
    load it instead of the product with MMADA_MI355X_LIB=<path> (tools/attn_sweep.py --forms does)."""
    from mmada_parallel_amd import build as pb

    srcs = [os.path.join(CSRC, s) for s in pb.SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in pb.HEADERS]
    if not force and os.path.exists(PRODUCT_TUNE_LIB) and all(os.path.getmtime(d) < os.path.getmtime(PRODUCT_TUNE_LIB) for d in deps):
        return PRODUCT_TUNE_LIB
    cmd = [pb._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DMMADA_TUNE", "-I", CSRC,
           "-I", os.path.join(ROOT, "include")] + srcs + ["-o", PRODUCT_TUNE_LIB]
    subprocess.run(cmd, check=True)
    return PRODUCT_TUNE_LIB


if __name__ == "__main__":
    print(build(force=True))
    print(build_product_tune(force=True))
