#!/usr/bin/env python
"""Time the MAGVITv2 token -> pixel decode (csrc/vq_decoder.hip) at the MMaDA-Parallel-M shape: 32x32 codes -> 512x512.
Measurement tool only (the headline bench is bench.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmada_parallel_amd import MAGVITv2, synth  # noqa: E402


def conv_flops(cfg, hz):
    """2*M*N*K of every convolution / attention GEMM of the decoder (fp32)."""
    shapes = synth.vq_decoder_param_shapes(cfg)
    L = len(cfg["ch_mult"])
    res = {lvl: hz * 2 ** (L - 1 - lvl) for lvl in range(L)}  # resolution of the res blocks of a level
    total = 0.0
    for name, s in shapes.items():
        if len(s) != 4:
            continue
        if name.startswith("up."):
            lvl = int(name.split(".")[1])
            r = res[lvl] * (2 if "upsample" in name else 1)
        elif name.startswith("conv_out"):
            r = res[0]
        else:
            r = hz
        total += 2.0 * r * r * s[0] * s[1] * s[2] * s[3]
    T, C = hz * hz, cfg["ch"] * cfg["ch_mult"][-1]
    return total + 4.0 * T * T * C


def main(B=1, hz=32, reps=10):
    cfg = synth.VQ_CFG_M
    vq = MAGVITv2.from_state_dict(synth.synthetic_vq_state_dict(cfg, 8), cfg, device="cuda:0")
    idx = torch.randint(0, 8192, (B, hz * hz), device="cuda:0")
    for _ in range(2):
        vq.decode_code(idx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        vq.decode_code(idx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    fl = conv_flops(cfg, hz) * B
    print(f"decode_code B={B} {hz}x{hz} codes -> {hz*16}x{hz*16}: {dt*1e3:.2f} ms, {fl/1e12:.3f} TFLOP fp32, "
          f"{fl/dt/1e12:.1f} TFLOP/s")


if __name__ == "__main__":
    main(B=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
