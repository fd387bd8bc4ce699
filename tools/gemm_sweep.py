#!/usr/bin/env python
"""Time the product GEMM's configurations (the -DMMADA_TUNE build of mmada_parallel_amd/csrc: tools/build_tune.py ->
tools/libmmada_mi355x_tune.so — the same sources as the product, plus diagnostic configurations) on the projection shapes of
the 8B block.  Variant codes: 100 = the planner's pick; 300 + c = the 8-phase kernel's configuration c (0..3 ship; 4.. are
tuning-build extras, 9..15 DIAGNOSTIC builds with wrong results: csrc/gemm8.hip launch_epi8); 1000 + BM = the 16-wave kernel.

    python tools/gemm_sweep.py [--variants 100,300,301,302,303] [--m 2440,4880]
Random bf16 operands (zero-filled operands clock ~20 % higher: never bench on zeros).  Interleaved rounds, median.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.build_tune import build_product_tune  # noqa: E402

SHAPES = {"qkv": (12288, 4096), "o": (4096, 4096), "gateup": (24576, 4096), "down": (4096, 12288)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="100,300,301,302,303")
    ap.add_argument("--m", default="2440,4880")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--shapes", default=None, help="name:N:K,... instead of the four TP=1 projection shapes")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--fill", default="randn", choices=["randn", "zeros", "ones"],
                    help="operand data: power (hence clock) depends on it — never quote a zero-fill number")
    ap.add_argument("--cold", type=int, default=6, help="rotate over this many operand copies (defeats the 256 MB MALL)")
    args = ap.parse_args()
    shapes = SHAPES
    if args.shapes:
        shapes = {n: (int(a), int(b)) for n, a, b in (x.split(":") for x in args.shapes.split(","))}
    import ctypes

    raw = ctypes.CDLL(build_product_tune())
    lib = raw
    lib.mmada_gemm_bt.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    lib.mmada_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.mmada_last_error.restype = ctypes.c_char_p

    class _V:
        @staticmethod
        def mmada_gemm_variant(v, A, W, C, M, N, K, st):
            raw.mmada_set_option(b"gemm_config", -1 if v == 100 else (v - 300 if 300 <= v < 1000 else v))
            rc = raw.mmada_gemm_bt(A, W, C, M, N, K, st)
            raw.mmada_set_option(b"gemm_config", -1)
            return rc

    lib = _V
    dev = "cuda:0"
    variants = [int(v) for v in args.variants.split(",")]
    st = torch.cuda.current_stream().cuda_stream
    print(f"{'shape':8s} {'M':>5s} " + " ".join(f"v{v:<6d}" for v in variants) + "   (TFLOP/s, median)")
    for M in [int(m) for m in args.m.split(",")]:
        for name, (N, K) in shapes.items():
            A = torch.randn(M, K, device=dev).to(torch.bfloat16)
            W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
            if args.fill != "randn":
                A.fill_(0.0 if args.fill == "zeros" else 1.0)
                W.fill_(0.0 if args.fill == "zeros" else 1.0)
            C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            As = [A] + [A.clone() for _ in range(args.cold - 1)]
            Ws = [W] + [W.clone() for _ in range(args.cold - 1)]
            res = {v: [] for v in variants}
            ok = {}
            cprod = None
            for v in variants:
                if lib.mmada_gemm_variant(v, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, st) != 0:
                    res[v] = None
                    continue
                if args.check:
                    ref = A[:256].float() @ W.float().t()
                    ok[v] = bool(((C[:256].float() - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-3 * ref.abs().max()).all())
                    if v == 100:
                        cprod = C.clone()
                    elif cprod is not None and not torch.equal(C, cprod):  # same K order, same MFMA: must be bit-identical
                        ok[v] = False
                        print(f"  v{v} differs from production: max |d| {(C.float() - cprod.float()).abs().max().item():.3e}, "
                              f"{(C != cprod).float().mean().item():.2e} of elements")
            for _ in range(args.rounds):
                for v in variants:
                    if res[v] is None:
                        continue
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    nrep = max(5, args.cold)
                    for i in range(nrep):
                        lib.mmada_gemm_variant(v, As[i % args.cold].data_ptr(), Ws[i % args.cold].data_ptr(), C.data_ptr(),
                                               M, N, K, st)
                    e1.record()
                    e1.synchronize()
                    res[v].append(e0.elapsed_time(e1) / nrep)
            cells = []
            for v in variants:
                if res[v] is None:
                    cells.append("  n/a  ")
                else:
                    ms = sorted(res[v])[len(res[v]) // 2]
                    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
                    cells.append(f"{tf:6.0f}{'' if ok.get(v, True) else '!'} ")
            print(f"{name:8s} {M:5d} " + " ".join(cells))


if __name__ == "__main__":
    main()
