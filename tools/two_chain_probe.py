#!/usr/bin/env python
"""Do two independent chains of kernels on two streams recover the launch-boundary tax?  (~9 us of cold start,
store burst and kernel boundary per one-round GEMM launch.)

The two sequences of a batch-2 forward (the two unconditional branches of an image step) never exchange data: they can run as
ONE batch-2 chain of launches (today) or as TWO batch-1 chains on two streams, so that one chain's ramps, tails and boundaries
fall under the other's main loops.  This probe times both forms on 8B blocks at L = 2438, interleaved in one process:

    python tools/two_chain_probe.py [--layers 4] [--prio 0|1]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmada_parallel_amd import LLaDAForMultiModalGeneration, abi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--L", type=int, default=2438)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warm", type=int, default=40)
    ap.add_argument("--prio", type=int, default=1, help="1: the second chain runs on a low-priority stream")
    args = ap.parse_args()
    dev = "cuda:0"
    lib = abi.lib()
    cfg = dict(synth.CFG_8B, n_layers=args.layers)
    sd = synth.synthetic_state_dict(cfg, seed=3, device="cpu")
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=dev, max_batch=2)
    g = torch.Generator().manual_seed(5)
    ids2 = torch.randint(0, 126000, (2, args.L), generator=g).to(dev)
    parts = [ids2[:1].contiguous(), ids2[1:].contiguous()]
    h = [model._lane_handle(0), model._lane_handle(1)]
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    s_main = torch.cuda.Stream(priority=-1)
    s_side = torch.cuda.Stream(priority=0 if args.prio else -1)
    streams = [s_main, s_side]

    def one_chain():
        with torch.cuda.stream(s_main):
            model.forward_body(ids2)

    def two_chains():
        for j in (0, 1):
            model._ensure_ws(1, args.L, lane=j)
        s_side.wait_stream(s_main)
        for j in (0, 1):
            abi.check(lib.mmada_embed(h[j], parts[j].data_ptr(), 1, args.L, streams[j].cuda_stream), "embed")
        for i in range(args.layers):
            for seg in (lib.mmada_attn_partial, lib.mmada_mlp_partial):
                for j in (0, 1):
                    abi.check(seg(h[j], i, streams[j].cuda_stream), "partial")
        s_main.wait_stream(s_side)

    forms = {"one batch-2 chain": one_chain, "two batch-1 chains on two streams": two_chains}
    for f in forms.values():
        for _ in range(args.warm):
            f()
    torch.cuda.synchronize()
    ms = {k: [] for k in forms}
    for _ in range(args.rounds):
        for k, f in forms.items():
            f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s_main):
                e0.record()
            for _ in range(args.iters):
                f()
            with torch.cuda.stream(s_main):
                e1.record()
            torch.cuda.synchronize()
            ms[k].append(e0.elapsed_time(e1) / args.iters)
    base = None
    for k in forms:
        t = sorted(ms[k])[len(ms[k]) // 2]
        base = base or t
        print(f"L={args.L} {args.layers} blocks, 2 sequences: {k:36s} {t:8.3f} ms per forward pair  ({t / base:.3f} x)", flush=True)


if __name__ == "__main__":
    main()
