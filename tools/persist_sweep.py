#!/usr/bin/env python
"""Persistent against plain launches of the 8-phase GEMM on multi-round products (csrc/gemm8.hip: gemm8_persist_kernel).

For every shape and row count: the planner's pick and each forced configuration, with `gemm_persistent` off / on, interleaved
rounds, median; the outputs of every (configuration, persistent) pair must equal the first one bit for bit.

    python tools/persist_sweep.py [--shapes o8:4096:512,down8:4096:1536] [--m 9760,19520]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmada_parallel_amd import abi  # noqa: E402

SHAPES = "o8:4096:512,down8:4096:1536,o4:4096:1024,down4:4096:3072,o2:4096:2048,down2:4096:6144,qkv8:1536:4096,gateup8:3072:4096,qkv1:12288:4096,o1:4096:4096"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=SHAPES, help="name:N:K,...")
    ap.add_argument("--m", default="2440,4880,9760,19520")
    ap.add_argument("--configs", default="-1,0,1,2,3")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cold", type=int, default=4)
    args = ap.parse_args()
    lib = abi.lib()
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    cfgs = [int(c) for c in args.configs.split(",")]
    forms = [(c, p) for c in cfgs for p in (0, 1)]
    print(f"{'shape':9s} {'M':>6s}  " + "  ".join(f"c{c:>2d}{'P' if p else ' '}" for c, p in forms) + "   (TFLOP/s, median;  P = persistent;  c-1 = planner)")
    for name, N, K in ((n, int(a), int(b)) for n, a, b in (x.split(":") for x in args.shapes.split(","))):
        for M in [int(m) for m in args.m.split(",")]:
            if 2.0 * M * N * K < 1e10:
                continue
            As = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(args.cold)]
            Ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(args.cold)]
            C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

            def run(c, p, n, i0=0):
                abi.check(lib.mmada_set_option(b"gemm_config", c), "opt")
                abi.check(lib.mmada_set_option(b"gemm_persistent", p), "opt")
                for i in range(n):
                    j = (i0 + i) % args.cold
                    abi.check(lib.mmada_gemm_bt(As[j].data_ptr(), Ws[j].data_ptr(), C.data_ptr(), M, N, K, st), "gemm")

            ref = None
            same = True
            for c, p in forms:
                C.zero_()
                run(c, p, 1)
                torch.cuda.synchronize()
                if ref is None:
                    ref = C.clone()
                elif not torch.equal(ref, C):
                    same = False
            ms = {f: [] for f in forms}
            for f in forms:
                run(*f, 3)
            for _ in range(args.rounds):
                for f in forms:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run(*f, args.iters)
                    e1.record()
                    torch.cuda.synchronize()
                    ms[f].append(e0.elapsed_time(e1) / args.iters)
            fl = 2.0 * M * N * K
            cells = []
            for f in forms:
                t = sorted(ms[f])[len(ms[f]) // 2]
                cells.append(f"{fl / t / 1e9:5.0f}")
            plan = lib.mmada_gemm_plan(M, N, K) if hasattr(lib, "mmada_gemm_plan") else -9
            print(f"{name:9s} {M:6d}  " + "  ".join(cells) + f"   bits equal: {same}  planner: {plan}", flush=True)
    lib.mmada_set_option(b"gemm_config", -1)
    lib.mmada_set_option(b"gemm_persistent", -1)


if __name__ == "__main__":
    main()
