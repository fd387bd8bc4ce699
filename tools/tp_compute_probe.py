#!/usr/bin/env python
"""Per-rank compute cost of the tensor-parallel shapes on ONE GPU (no collective): rank 0 of TP=n with the weak-scaling
batch B=n of bench.py, timed without the all-reduce.  Perfect weak scaling would keep ms/block equal to TP=1, B=1;
the ratio is the compute-side ceiling of the N-GPU scaling efficiency (the xGMI all-reduce comes on top).
Measurement tool only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import torch  # noqa: E402

from mmada_parallel_amd import LLaDAForMultiModalGeneration, abi, synth  # noqa: E402

KINDS = ["qkv", "attn", "o", "gateup", "down"]


def main(layers=4, reps=5):
    dev = "cuda:0"
    cfg = dict(synth.CFG_8B, n_layers=layers)
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    base = None
    for tp in (1, 2, 4, 8):
        sd = synth.synthetic_state_dict(cfg, seed=0, device=dev)
        model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=dev, tp_rank=0,
                                                             tp_size=tp, max_batch=2 * tp)
        del sd
        for B in (tp, 2 * tp):
            ids = job["input_ids"].repeat(B, 1).to(dev)
            for _ in range(2):
                model.forward_body(ids)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                model.forward_body(ids)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / reps / layers * 1e3
            abi.check(model._lib.mmada_profile_begin(model._handle, layers // 2), "profile_begin")
            model.forward_body(ids)
            torch.cuda.synchronize()
            cnt, ms, fl = (C.c_int32 * 5)(), (C.c_double * 5)(), (C.c_double * 5)()
            abi.check(model._lib.mmada_profile_end(model._handle, cnt, ms, fl), "profile_end")
            kinds = "  ".join(f"{k} {ms[i] / max(cnt[i], 1):.3f}ms/{fl[i] / max(ms[i], 1e-9) / 1e9:.0f}TF"
                              for i, k in enumerate(KINDS))
            per_job = t / (B / tp)
            if base is None:
                base = per_job
            print(f"TP={tp} B={B}: {t:.3f} ms/block on rank 0 = {per_job:.3f} ms per (job/rank)  "
                  f"compute-side efficiency {base / per_job:.3f}\n    lane 0: {kinds}", flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
