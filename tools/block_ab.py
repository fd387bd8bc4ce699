#!/usr/bin/env python
"""In-process A/B of library options on the kernels of 8B blocks at BASELINE configs[1] shapes (L = 2438, B = 1 and 2): option
settings are interleaved round by round in ONE process (same clock, same heat), per-kernel time is the library's live hipEvent
timing (mmada_profile_begin / _end around one block).  Minutes instead of one bench run per setting.

    python tools/block_ab.py "attention_form=1" "attention_form=2"
    python tools/block_ab.py --layers 2 --batches 1,2 --rounds 6 "gemm_tile_order=0" "gemm_tile_order=1"
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmada_parallel_amd import LLaDAForMultiModalGeneration, abi, synth  # noqa: E402

KINDS = ["qkv", "attn", "attn_out", "gate_up", "down"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+", help='comma lists of option=value (mmada_set_option names), one per variant')
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--batches", default="1,2")
    ap.add_argument("--L", type=int, default=2438)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=12, help="forwards per (round, variant)")
    ap.add_argument("--warm", type=int, default=60)
    args = ap.parse_args()
    dev = "cuda:0"
    lib = abi.lib()
    cfg = dict(synth.CFG_8B, n_layers=args.layers)
    sd = synth.synthetic_state_dict(cfg, seed=3, device="cpu")
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=dev, max_batch=2)
    h = model._handle
    variants = []
    for sset in args.settings:
        variants.append([(kv.split("=")[0].encode(), int(kv.split("=")[1])) for kv in sset.split(",") if kv])
    names = set(n for v in variants for n, _ in v)
    g = torch.Generator().manual_seed(5)
    for B in [int(b) for b in args.batches.split(",")]:
        ids = torch.randint(0, 126000, (B, args.L), generator=g).to(dev)
        for _ in range(args.warm):
            model.forward_body(ids)
        torch.cuda.synchronize()
        acc = [[[] for _ in KINDS] for _ in variants]
        wall = [[] for _ in variants]
        for _ in range(args.rounds):
            for vi, v in enumerate(variants):
                for n, val in v:
                    abi.check(lib.mmada_set_option(n, val), "set_option")
                model.forward_body(ids)   # the first launch under a new setting is not timed
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                cnt, ms, fl = (C.c_int32 * 5)(), (C.c_double * 5)(), (C.c_double * 5)()
                tot = [0.0] * 5
                n = [0] * 5
                e0.record()
                for _ in range(args.iters):
                    abi.check(lib.mmada_profile_begin(h, args.layers // 2 if args.layers > 2 else 0), "profile_begin")
                    model.forward_body(ids)
                    abi.check(lib.mmada_profile_end(h, cnt, ms, fl), "profile_end")
                    for k in range(5):
                        tot[k] += ms[k]
                        n[k] += cnt[k]
                e1.record()
                torch.cuda.synchronize()
                wall[vi].append(e0.elapsed_time(e1) / args.iters)
                for k in range(5):
                    acc[vi][k].append(tot[k] / max(1, n[k]))
                for nme in names:   # back to defaults between variants
                    lib.mmada_set_option(nme, -1 if nme in (b"gemm_config", b"attention_form", b"gemm_tile_order") else 1)
        for vi, sset in enumerate(args.settings):
            med = [sorted(a)[len(a) // 2] * 1e3 for a in acc[vi]]
            w = sorted(wall[vi])[len(wall[vi]) // 2]
            flops = [2.0 * B * args.L * 12288 * 4096, 4.0 * 32 * B * args.L * args.L * 128, 2.0 * B * args.L * 4096 * 4096,
                     2.0 * B * args.L * 24576 * 4096, 2.0 * B * args.L * 4096 * 12288]
            cells = "  ".join(f"{KINDS[k]} {med[k]:7.1f} us {flops[k] / med[k] / 1e6:5.0f} TF" for k in range(5))
            print(f"B={B} [{sset}]  {cells}  | forward {w:.3f} ms ({args.layers} blocks)", flush=True)


if __name__ == "__main__":
    main()
