#!/usr/bin/env python
"""CPU baseline from the UNMODIFIED reference: imports LLaDAForMultiModalGeneration from
/root/reference/MMaDA-Parallel-A and times its forward(infer=True) at the real 8B block shapes, L = 2438, bf16, on this
host's cores.  The reference tree exists only in the build container (never on the GPU box), so this runs HERE and its
result is committed as profiles/r02_cpu_reference.json; bench.py reports it next to the oracle port it times on the GPU
box's own cores.  Bounded sample: `--layers` blocks (default 4) + the full [L, V] LM head the reference always computes,
extrapolated to 32 blocks and to the 256 forwards of one BASELINE configs[1] image.

    python tools/cpu_reference_baseline.py --end-to-end      (BASELINE.md §3, configs[0])
runs the unmodified `generate_ti2ti` (generators/parallel_generator.py:102-368) on the unmodified, full-depth
LLaDAForMultiModalGeneration ONCE, end to end: 256x256, text_steps 32, timesteps 16, temperature 0, L = 1654, 64 forwards of
the 8B model with seeded synthetic weights, and writes seconds per image next to the core count
(profiles/r03_cpu_reference_e2e.json; `python bench.py --config 0` prints the same job on the GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/MMaDA-Parallel-A"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--end-to-end", action="store_true", help="configs[0]: one whole generate_ti2ti job on 32 layers")
    args = ap.parse_args()
    if args.end_to_end:
        args.layers = 32
    if args.out is None:
        args.out = os.path.join(ROOT, "profiles", "r03_cpu_reference_e2e.json" if args.end_to_end else "r02_cpu_reference.json")
    from mmada_parallel_amd import synth
    from model import LLaDAForMultiModalGeneration          # the reference's own class (unmodified)
    from model.configuration_llada import LLaDAConfig

    if args.threads:
        torch.set_num_threads(args.threads)
    flags = open("/proc/cpuinfo").read()
    cfg = dict(synth.CFG_8B, n_layers=args.layers)
    sd = synth.synthetic_state_dict(cfg, seed=0, device="cpu")
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        model = LLaDAForMultiModalGeneration(LLaDAConfig(**synth.full_config(cfg)))
    finally:
        torch.set_default_dtype(old)
    model.load_state_dict(sd, strict=True)
    model = model.to(torch.bfloat16).eval()
    del sd
    if args.end_to_end:
        return end_to_end(model, cfg, flags, args)
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids = job["input_ids"]
    L = ids.shape[1]
    # time the blocks and the head separately: hooks on the reference modules, its forward untouched
    marks = {}
    blocks = model.model.transformer.blocks
    blocks[0].register_forward_pre_hook(lambda *_: marks.__setitem__("b0", time.perf_counter()))
    blocks[-1].register_forward_hook(lambda *_: marks.__setitem__("b1", time.perf_counter()))
    t_blocks, t_total = [], []
    with torch.no_grad():
        for _ in range(args.reps + 1):  # first call = warm-up (oneDNN primitive creation)
            t0 = time.perf_counter()
            out = model(ids, infer=True, use_cache=False).logits
            t1 = time.perf_counter()
            t_blocks.append((marks["b1"] - marks["b0"]) / args.layers)
            t_total.append(t1 - t0)
            assert out.shape == (1, L, cfg["embedding_size"])
    per_block = min(t_blocks[1:])
    rest = min(t - b * args.layers for t, b in zip(t_total[1:], t_blocks[1:]))   # embedding + ln_f + dense [L, V] head
    per_forward = 32 * per_block + rest
    res = {"kind": "reference", "what": "unmodified LLaDAForMultiModalGeneration.forward(infer=True), bf16, device=cpu",
           "where": "build container (the reference tree does not exist on the GPU box)",
           "cores": os.cpu_count(), "threads": torch.get_num_threads(), "amx_bf16": "amx_bf16" in flags,
           "avx512_bf16": "avx512_bf16" in flags, "L": L, "sample_layers": args.layers, "reps": args.reps,
           "seconds_per_block": per_block, "seconds_embed_lnf_dense_head": rest, "seconds_per_forward_32_blocks": per_forward,
           "forwards_per_image": 256, "images_per_sec": 1.0 / (256 * per_forward), "unit": "images/sec"}
    print(json.dumps(res))
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


def end_to_end(model, cfg, flags, args):
    from generators.parallel_generator import generate_ti2ti   # the reference's own sampler (unmodified)
    from mmada_parallel_amd import synth

    job = synth.synthetic_job(256, 256, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids = job["input_ids"]
    calls = []
    inner = model.forward

    def timed_forward(*a, **kw):
        t0 = time.perf_counter()
        out = inner(*a, **kw)
        calls.append((int(a[0].shape[0]) if a else int(kw["input_ids"].shape[0]), time.perf_counter() - t0))
        return out

    model.forward = timed_forward
    torch.manual_seed(0)
    t0 = time.perf_counter()
    with torch.no_grad():
        vq, text = generate_ti2ti(model, ids, job["text_start"], job["text_end"], job["image_start"], job["seq_len"],
                                  job["newline_every"], text_steps=32, timesteps=16, temperature=0.0, text_temperature=0.0,
                                  cfg_scale=0.0, cfg_img=4.0, uncon_text=job["uncon_text"], uncon_image=job["uncon_image"],
                                  tokenizer=None)
    wall = time.perf_counter() - t0
    seq_forwards = sum(b for b, _ in calls)
    t_model = sum(t for _, t in calls)
    res = {"kind": "reference", "what": "unmodified generate_ti2ti + LLaDAForMultiModalGeneration, bf16, device=cpu, one whole job",
           "workload": "BASELINE configs[0]: 256x256, text_steps=32, timesteps=16, cfg_img=4.0, temperature=0",
           "where": "build container (the reference tree does not exist on the GPU box)", "L": int(ids.shape[1]),
           "cores": os.cpu_count(), "threads": torch.get_num_threads(), "amx_bf16": "amx_bf16" in flags,
           "avx512_bf16": "avx512_bf16" in flags, "model_calls": len(calls), "sequence_forwards": seq_forwards,
           "seconds_per_image": wall, "seconds_in_model_forward": t_model, "seconds_sampler_and_rest": wall - t_model,
           "seconds_per_sequence_forward": t_model / max(seq_forwards, 1), "images_per_sec": 1.0 / wall, "unit": "images/sec",
           "vq_tokens": len(vq), "text_tokens": len(text),
           "note": "the host was shared with compile jobs of the build session while this ran"}
    print(json.dumps(res))
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
