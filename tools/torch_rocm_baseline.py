#!/usr/bin/env python
"""Like-for-like GPU baseline: the oracle's plain-PyTorch forward (oracle/llada_oracle.py — the reference's op
sequence: F.linear / F.scaled_dot_product_attention / fp32 RoPE / RMSNorm as separate torch ops) run under
PyTorch-ROCm on the same MI355X, next to the HIP path.  Measurement tool only (not product, not bench.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmada_parallel_amd import LLaDAForMultiModalGeneration, synth  # noqa: E402
from oracle import llada_oracle  # noqa: E402


def main(layers=4, reps=5):
    dev = "cuda:0"
    cfg = dict(synth.CFG_8B, n_layers=layers)
    sd = synth.synthetic_state_dict(cfg, seed=0, device=dev)
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids = job["input_ids"].to(dev)
    L = ids.shape[1]
    with torch.no_grad():
        for _ in range(2):
            llada_oracle.forward_hidden(sd, cfg, ids)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            llada_oracle.forward_hidden(sd, cfg, ids)
        torch.cuda.synchronize()
        t_torch = (time.perf_counter() - t0) / reps / layers
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=dev)
    for _ in range(2):
        model.forward_body(ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model.forward_body(ids)
    torch.cuda.synchronize()
    t_hip = (time.perf_counter() - t0) / reps / layers
    fl = (2.0 * L * (4 * 4096 * 4096 + 3 * 4096 * 12288) + 4.0 * L * L * 4096)
    print(f"per block at L={L}: torch-ROCm eager {t_torch*1e3:.3f} ms ({fl/t_torch/1e12:.0f} TFLOP/s)  "
          f"libmmada_mi355x {t_hip*1e3:.3f} ms ({fl/t_hip/1e12:.0f} TFLOP/s)  speed-up {t_torch/t_hip:.2f}x")


if __name__ == "__main__":
    main()
