#!/usr/bin/env python
"""Launch the gate/up GEMM (SwiGLU epilogue, 8B shapes) under each tile order, a few launches per order, in a fixed sequence —
run it under `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv` and summarise the CSV with --summarise: fabric-side
bytes per launch (gfx950: 2 x FETCH_SIZE KiB) and duration per (M, order), dispatches matched by their order in the file.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o t -- python tools/tile_order_fetch.py
    python tools/tile_order_fetch.py --summarise out/*/t_counter_collection.csv
"""
import csv
import sys

ORDERS = [0, 408, 216, 804, 1602, 602]
MS = [2440, 4880]
REPS = 4


def run():
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from mmada_parallel_amd import abi

    lib = abi.lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    N, K = 24576, 4096
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    for M in MS:
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        C = torch.empty(M, N // 2, dtype=torch.bfloat16, device="cuda")
        for _ in range(3):   # untimed: table fill, clocks
            abi.check(lib.mmada_gemm_swiglu_bt(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, st), "gemm")
        torch.cuda.synchronize()
        for order in ORDERS:
            abi.check(lib.mmada_set_option(b"gemm_tile_order", order), "set_option")
            for _ in range(REPS):
                abi.check(lib.mmada_gemm_swiglu_bt(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, st), "gemm")
            torch.cuda.synchronize()
    lib.mmada_set_option(b"gemm_tile_order", 0)


def summarise(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if "gemm8_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                rows.append((int(r["Start_Timestamp"]), float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    rows.sort()
    i = 0
    alg = {2440: (2440 * 4096 + 24576 * 4096 + 2440 * 12288) * 2, 4880: (4880 * 4096 + 24576 * 4096 + 4880 * 12288) * 2}
    for M in MS:
        i += 3
        for order in ORDERS:
            grp = rows[i:i + REPS]
            i += REPS
            if len(grp) < REPS:
                print("short CSV")
                return
            fetch = sum(g[1] for g in grp[1:]) / (REPS - 1) * 2 * 1024
            dur = sorted(g[2] for g in grp[1:])[(REPS - 1) // 2]
            print(f"M={M} tile order {order:5d}: fetch {fetch / 1e6:8.1f} MB per launch = {fetch / alg[M]:.2f} x algorithmic (reads only), median {dur:7.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        run()
