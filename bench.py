#!/usr/bin/env python
"""Headline benchmark: images/sec of the MMaDA-Parallel-A 8B parallel text+image sampler on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one complete TI2TI job of BASELINE.json configs[1]: 512x512 output, text_steps=128, timesteps=64,
cfg_scale=0, cfg_img=4.0, temperature=0, batch 1 per tensor-parallel group, L = 2438 tokens, 256 forwards of the 8B
denoiser (128 conditional + 64 x 2 unconditional), synthetic weights / tokens (no checkpoint offline).
With N GPUs the model is tensor-parallel over N ranks (RCCL all-reduce over xGMI) and the batch is N jobs (weak
scaling, BASELINE configs[2]).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
KIND_NAMES = ["qkv_rope_gemm", "flash_attention", "attn_out_gemm", "gate_up_swiglu_gemm", "down_gemm"]


def body_flops(cfg, L):
    d, F, nl = cfg["d_model"], cfg["mlp_hidden_size"], cfg["n_layers"]
    kv = (cfg.get("n_kv_heads") or cfg["n_heads"]) * (d // cfg["n_heads"])
    lin = 2.0 * L * (d * d + 2 * d * kv + d * d + 3 * d * F)
    return nl * (lin + 4.0 * L * L * d)


def last_block_saving(cfg, L, lo, hi):
    """FLOPs the last block does not spend when only rows [lo, hi) are consumed (mmada_set_consumed_rows): attention
    queries, attn_out and the MLP run on the window (start rounded down to 32 rows); QKV still covers every row."""
    d, F = cfg["d_model"], cfg["mlp_hidden_size"]
    w = hi - (lo & ~31)
    if w >= L:
        return 0.0
    return (L - w) * (2.0 * (d * d + 3 * d * F) + 4.0 * L * d)


def job_flops(cfg, L, T, N, text_steps, n_img_steps, V, CB, job=None):
    """FLOPs of one image actually required: LM head on the consumed rows/columns only and, when `job`
    is given, the last block on the consumed rows only — skipped work is not counted as achieved."""
    fb = body_flops(cfg, L)
    head_text = 2.0 * T * cfg["d_model"] * V
    head_img = 2.0 * N * cfg["d_model"] * CB
    total = text_steps * (fb + head_text) + n_img_steps * 2 * fb + n_img_steps * 3 * head_img
    if job is not None:
        img = (job["image_start"], job["text_start"] - 1)  # image span incl. newlines, text follows after <eoi>
        total -= (text_steps - n_img_steps) * last_block_saving(cfg, L, job["text_start"], job["text_end"])
        total -= n_img_steps * last_block_saving(cfg, L, img[0], job["text_end"])
        total -= n_img_steps * 2 * last_block_saving(cfg, L, img[0], img[1])
    return total


def cpu_baseline(cfg, job, sample_layers=4, reps=2):
    """Oracle (CPU restatement of the reference forward) timed on the host cores on a bounded sample:
    `sample_layers` blocks + consumed LM-head rows at the full shape, extrapolated to 256 forwards / image."""
    from mmada_parallel_amd import synth
    from oracle import llada_oracle

    small = dict(cfg, n_layers=sample_layers)
    sd = synth.synthetic_state_dict(small, seed=0, device="cpu")
    ids = job["input_ids"]
    L = ids.shape[1]
    T, N = job["text_end"] - job["text_start"], job["seq_len"]
    t_layers, t_head = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        x = llada_oracle.forward_hidden(sd, small, ids)
        t1 = time.perf_counter()
        llada_oracle.head(sd, small, x[:, job["text_start"]:job["text_end"]])
        llada_oracle.head(sd, small, x[:, :N], synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK)
        t2 = time.perf_counter()
        t_layers.append((t1 - t0) / sample_layers)
        t_head.append(t2 - t1)
    per_forward = min(t_layers) * cfg["n_layers"]
    n_fwd = 128 + 2 * 64
    per_image = n_fwd * per_forward + 128 * min(t_head)
    return {"value": 1.0 / per_image, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/llada_oracle.py forward of {sample_layers} of {cfg['n_layers']} blocks + consumed LM-head "
                      f"rows at L={L} (best of {reps}), extrapolated to {n_fwd} forwards/image: "
                      f"{per_forward:.2f} s/forward"}


class SmiSampler:
    """Samples `rocm-smi` (sclk, socket power) about once a second in a host thread while the timed region runs; the
    medians go into the JSON line as evidence of the clock the MFMA peak has to be read against."""

    def __init__(self, device_index):
        import threading

        self.dev, self.rows, self._stop = device_index, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import subprocess

        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showclocks", "--showpower"], capture_output=True,
                                     text=True, timeout=10).stdout
                c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
                p = re.search(r"Power \(W\): ([\d.]+)", out)
                if c and p:
                    self.rows.append((int(c.group(1)), float(p.group(1))))
            except Exception:
                return
            self._stop.wait(1.0)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._t.join(timeout=15)

    def summary(self):
        if not self.rows:
            return None
        sc = sorted(r[0] for r in self.rows)
        pw = sorted(r[1] for r in self.rows)
        return {"samples": len(self.rows), "sclk_mhz_median": sc[len(sc) // 2], "socket_power_w_median": pw[len(pw) // 2]}


def measured_traffic(kernel):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/traffic.json:
    FETCH_SIZE / WRITE_SIZE collected separately, gfx950 correction applied), averaged over the launch mix of one
    image.  PMC collection cannot run inside the timed bench; null if the kernel was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        if t["kernel"] != kernel:
            return None
        mix = t["launch_mix"]
        return sum(t["per_launch_bytes"][k] * mix[k] for k in mix) / sum(mix.values())
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--text-steps", type=int, default=128, help="debug only: any other value marks the line reduced")
    ap.add_argument("--timesteps", type=int, default=64, help="debug only")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (marks the line reduced)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parallelism", choices=["tp", "dp"], default="tp",
                    help="tp (default, BASELINE configs[2]): the model is tensor-parallel over all ranks, batch = N jobs; "
                         "dp: every rank holds the full 16 GB model and runs its own job, no data-path collective")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # MMADA_BENCH_ONE_GPU=1 (test rigs with a single GPU): every rank uses cuda:0 and the collective runs over gloo, so
    # the multi-process tensor-parallel path can be exercised end to end; such a line is marked and is not a measurement.
    one_gpu = os.environ.get("MMADA_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "RANK" in os.environ  # launched by torch.distributed.run: one rank per GPU over RCCL
    if use_dist:
        import torch.distributed as dist

        # the collective's kernels must not queue behind a whole round of GEMM workgroups: high-priority RCCL stream
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from mmada_parallel_amd import LLaDAForMultiModalGeneration, abi, generate_ti2ti, synth

    cfg = dict(synth.CFG_8B)
    if args.layers:
        cfg["n_layers"] = args.layers
    reduced = args.text_steps != 128 or args.timesteps != 64 or args.layers is not None or one_gpu
    full = synth.full_config(cfg)
    tp = world if args.parallelism == "tp" else 1
    B = tp  # weak scaling: one job per rank-equivalent; tp: model sharded over all ranks, dp: replicas
    sd = synth.synthetic_state_dict(cfg, seed=0, device=str(dev))
    model = LLaDAForMultiModalGeneration.from_state_dict(full, sd, device=dev, tp_rank=rank if tp > 1 else 0, tp_size=tp,
                                                         max_batch=2 * B)
    del sd
    torch.cuda.empty_cache()

    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids = job["input_ids"].repeat(B, 1).to(dev)
    L = ids.shape[1]
    T, N = job["text_end"] - job["text_start"], job["seq_len"]

    def run_once():
        return generate_ti2ti(model, ids, job["text_start"], job["text_end"], job["image_start"], job["seq_len"],
                              job["newline_every"], text_steps=args.text_steps, timesteps=args.timesteps, temperature=0.0,
                              text_temperature=0.0, cfg_scale=0.0, cfg_img=4.0, uncon_text=job["uncon_text"],
                              uncon_image=job["uncon_image"], return_state=True)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_once()
    lib, h = model._lib, model._handle
    barrier()
    abi.check(lib.mmada_profile_begin(h, cfg["n_layers"] // 2), "profile_begin")
    smi = SmiSampler(local if not one_gpu else 0) if rank == 0 else None
    if smi:
        smi.__enter__()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vq, _, final_ids = run_once()
    barrier()
    dt = time.perf_counter() - t0
    if smi:
        smi.__exit__()
    cnt, ms, fl = (C.c_int32 * 5)(), (C.c_double * 5)(), (C.c_double * 5)()
    abi.check(lib.mmada_profile_end(h, cnt, ms, fl), "profile_end")
    ranks_agree = None
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
        # outside the timed region: under tensor parallelism every rank must have sampled the same image tokens
        allv = [None] * world
        dist.all_gather_object(allv, final_ids.tolist())  # all B jobs, before the one random fill of the read-out
        ranks_agree = all(a == allv[0] for a in allv) if tp > 1 else None
    ar_probe = None
    if use_dist and tp > 1:
        # outside the timed region: the collective the forward issues 128 times per micro-batch (one micro-batch's residual
        # stream, bf16), timed alone, so a scaling run also records what the fabric delivered for that message size
        lane_rows = ((B + 1) // 2) * ((L + 7) // 8 * 8)
        buf = torch.zeros(lane_rows * cfg["d_model"], dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        ar_ms = (time.perf_counter() - t1) / 10 * 1e3
        nbytes = buf.numel() * 2
        ar_probe = {"bytes": nbytes, "ms": ar_ms, "busbw_GBps": 2.0 * (world - 1) / world * nbytes / (ar_ms * 1e-3) / 1e9,
                    "calls_per_forward": 2 * cfg["n_layers"] * 2}

    if rank == 0:
        from mmada_parallel_amd.generators.parallel_generator import image_step_indices

        n_img = len(set(image_step_indices(args.text_steps, args.timesteps)))
        windowed = os.environ.get("MMADA_NO_WINDOW") != "1"
        fl_img = job_flops(cfg, L, T, N, args.text_steps, n_img, cfg["embedding_size"], synth.CODEBOOK,
                           job if windowed else None)
        images = args.steps * world  # tp: B = world jobs in one group; dp: one job on each of `world` replicas
        value = images / dt
        kinds = {}
        for i, nm in enumerate(KIND_NAMES):
            if cnt[i]:
                kinds[nm] = {"launches": cnt[i], "avg_ms": ms[i] / cnt[i], "tflops": fl[i] / (ms[i] * 1e-3) / 1e12}
        dom = max(range(5), key=lambda i: ms[i]) if any(cnt) else 3
        ach = fl[dom] / (ms[dom] * 1e-3) / 1e12 if cnt[dom] else 0.0
        out = {
            "metric": "images/sec (512x512, 64 img + 128 text steps) MMaDA-Parallel-A 8B",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: MMaDA-Parallel-A 8B, 512x512, timesteps=64, text_steps=128, "
                                   "cfg_img=4.0, temperature=0, L=2438, 256 forwards/image" + (" [REDUCED DEBUG RUN]" if reduced else ""),
                       "global_batch": world, "seq_len": L, "parallelism": f"{args.parallelism}{world}",
                       "text_steps": args.text_steps, "timesteps": args.timesteps, "n_layers": cfg["n_layers"],
                       "algorithmic_pflop_per_image": fl_img / 1e15,
                       "job_mfma_frac": value * fl_img / 1e12 / (world * MFMA_BF16_PEAK_TFLOPS),
                       "tp_ranks_agree": ranks_agree, "allreduce_probe": ar_probe,
                       "rocm_smi_during_run": smi.summary() if smi else None,
                       "kernels": kinds},
            "roofline": {"bound": "mfma", "kernel": KIND_NAMES[dom], "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": measured_traffic(KIND_NAMES[dom])},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, job)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
