"""Build tools/libmmada_mi355x_tune.so: the PRODUCT sources (mmada_parallel_amd/csrc) compiled with -DMMADA_TUNE, which adds
diagnostic GEMM configurations (gemm8.hip: no-MFMA / no-DMA / no-ds_read builds, the other read schedule per tile) and the
diagnostic attention variants to the same library.  Load it instead of the product with MMADA_MI355X_LIB=<path>; every sweep
tool (tools/gemm_sweep.py, tools/attn_sweep.py, tools/fit_gemm8_cost.py's input) measures THESE kernels — there is no fork of
any kernel source outside csrc/."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PRODUCT_TUNE_LIB = os.path.join(ROOT, "tools", "libmmada_mi355x_tune.so")


def build_product_tune(force=False):
    from mmada_parallel_amd import build as pb

    deps = [os.path.join(pb.CSRC, s) for s in pb.SOURCES + pb.HEADERS]
    if os.environ.get("MMADA_TUNE_PREBUILT") == "1" and os.path.exists(PRODUCT_TUNE_LIB):
        return PRODUCT_TUNE_LIB   # a snapshot on the GPU box: file times are those of the copy, the library travelled with it
    if not force and os.path.exists(PRODUCT_TUNE_LIB) and all(os.path.getmtime(d) < os.path.getmtime(PRODUCT_TUNE_LIB) for d in deps):
        return PRODUCT_TUNE_LIB
    objs = pb.compile_objects(os.path.join(ROOT, "tools", "_obj_tune"), extra_flags=["-DMMADA_TUNE"], force=force)
    pb._run([pb._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", PRODUCT_TUNE_LIB])
    return PRODUCT_TUNE_LIB


if __name__ == "__main__":
    print(build_product_tune(force=True))
