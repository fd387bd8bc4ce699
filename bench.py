#!/usr/bin/env python
"""Headline benchmark: images/sec of the MMaDA-Parallel 8B parallel text+image sampler on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {0,1,3,4}] [--scaling {weak,strong}]
        N = 1, 2, 4, 8 as a PLAIN command: for N > 1 bench.py starts its own N ranks (torch.distributed.run on 127.0.0.1,
        one rank per GPU over RCCL) and prints rank 0's JSON line as the last line of stdout
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W   (same thing)

--config 1 (default, the headline): one "step" = one complete TI2TI job of BASELINE.json configs[1]: MMaDA-Parallel-A,
  512x512 output, text_steps=128, timesteps=64, cfg_scale=0, cfg_img=4.0, temperature=0, L = 2438 tokens, 256 forwards
  of the 8B denoiser (128 conditional + 64 x 2 unconditional).  With N GPUs the model is tensor-parallel over N ranks and
  the batch is N jobs (weak scaling, BASELINE configs[2]).
--config 0: BASELINE configs[0], the reference's CPU-runnable plumbing case, on the GPU: 1 prompt, 256x256, text_steps=32,
  timesteps=16, temperature=0, L = 1654, 64 forwards; the unmodified reference's end-to-end time for the same job
  (tools/cpu_reference_baseline.py --end-to-end, build container) is attached to the line.
--scaling strong (config 1): ONE job whatever the rank count (batch 1, tensor-parallel over all ranks) — the regime in which
  every kernel shrinks with N while the launch count does not; use with --graph on.  Default: weak (batch = N jobs).
--config 3: BASELINE configs[3] in its single-GPU form: MMaDA-Parallel-M (MAGVITv2 tokenizer), a batch of 4 edit jobs,
  each pixels in -> pixels out: MAGVITv2.get_code -> interleave_generate (text_steps 128, image_steps 30, text_cfg 2.5,
  image_cfg 4.0: 128 batch-2 forwards at L = 2349) -> decode_code.  One "step" = the 4 jobs.
--config 4: BASELINE configs[4] in its single-GPU form: MMaDA-Parallel-A editing, ONE batch of 16 jobs through
  generate_ti2ti with cfg_scale=3.0 and cfg_img=4.0 (all three CFG branches contribute), L = 2438: batch-16 conditional
  and batch-32 unconditional forwards.  One "step" = the 16 jobs.
Synthetic weights / tokens / pixels (no checkpoint offline).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the tensor-parallel exchange stream parks a (one-wave) wait kernel until the peers arrive: it must own a hardware queue,
# not share one with the compute stream (the HIP runtime multiplexes streams over 4 queues by default)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# the bench records what BOTH transports deliver (config.allreduce_probe): connect RCCL next to the pull transport
os.environ.setdefault("MMADA_TP_PROBE_RCCL", "1")

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
    w = ((hi + 7) & ~7) - (lo & ~31)   # start rounded down to the 32-query granule, end up to 8 rows
    if w >= L:
        return 0.0
    return (L - w) * (2.0 * (d * d + 3 * d * F) + 4.0 * L * d)


def job_flops(cfg, L, T, N, text_steps, n_img_steps, V, CB, job=None):
    """FLOPs of one A image actually required: LM head on the consumed rows/columns only and, when `job`
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


def m_job_flops(cfg, L, T, N, text_steps, n_img_steps, V, CB, img_start, text_start, windowed=True):
    """FLOPs of one M edit (interleave_generate): every step is ONE batch-2 forward (cond + uncond); text head on the
    2T text rows every step, image head on the 2N image rows of image steps; last block on the consumed window."""
    fb = body_flops(cfg, L)
    per_seq = text_steps * fb
    if windowed:
        per_seq -= (text_steps - n_img_steps) * last_block_saving(cfg, L, text_start, L)
        per_seq -= n_img_steps * last_block_saving(cfg, L, img_start, L)
    heads = text_steps * 2 * (2.0 * T * cfg["d_model"] * V) + n_img_steps * 2 * (2.0 * N * cfg["d_model"] * CB)
    return 2 * per_seq + heads


def cpu_baseline(cfg, ids, text_rows, img_rows, seq_forwards, text_heads, img_heads, sample_layers=4, reps=2):
    """Oracle (CPU restatement of the reference forward, same torch ops as the reference model) timed on THIS box's host
    cores on a bounded sample — `sample_layers` blocks + the consumed LM-head rows at the full shape — and extrapolated
    to the `seq_forwards` sequence-forwards (+ head calls) of one image.  The thread count is probed (the fastest of a few
    candidates on one block: every hyper-thread is NOT the fastest on a 2-socket box), and the unmodified reference's
    own number, measured where the reference tree exists (tools/cpu_reference_baseline.py), is attached."""
    from mmada_parallel_amd import synth
    from oracle import llada_oracle

    small = dict(cfg, n_layers=sample_layers)
    sd = synth.synthetic_state_dict(small, seed=0, device="cpu")
    L = ids.shape[1]
    ncpu = os.cpu_count() or 8
    probe = dict(cfg, n_layers=1)
    best_t, best_n = None, None
    for n in sorted({min(ncpu, c) for c in (8, 16, 32, 64, ncpu // 2, ncpu) if c >= 1}):
        torch.set_num_threads(n)
        llada_oracle.forward_hidden(sd, probe, ids)  # warm
        t0 = time.perf_counter()
        llada_oracle.forward_hidden(sd, probe, ids)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_n = dt, n
    torch.set_num_threads(best_n)
    t_layers, t_text, t_img = [], [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        x = llada_oracle.forward_hidden(sd, small, ids)
        t1 = time.perf_counter()
        llada_oracle.head(sd, small, x[:, text_rows[0]:text_rows[1]])
        t2 = time.perf_counter()
        llada_oracle.head(sd, small, x[:, img_rows[0]:img_rows[1]], synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK)
        t3 = time.perf_counter()
        t_layers.append((t1 - t0) / sample_layers)
        t_text.append(t2 - t1)
        t_img.append(t3 - t2)
    per_forward = min(t_layers) * cfg["n_layers"]
    per_image = seq_forwards * per_forward + text_heads * min(t_text) + img_heads * min(t_img)
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        flags = ""
    out = {"value": 1.0 / per_image, "unit": "images/sec", "cores": best_n, "kind": "port",
           "threads_probed": "fastest of {8,16,32,64,ncpu/2,ncpu} on one block", "logical_cpus": ncpu,
           "amx_bf16": "amx_bf16" in flags, "avx512_bf16": "avx512_bf16" in flags,
           "seconds_per_sequence_forward": per_forward,
           "sample": f"oracle/llada_oracle.py forward of {sample_layers} of {cfg['n_layers']} blocks + consumed LM-head "
                     f"rows at L={L} (best of {reps}), extrapolated to {seq_forwards} sequence-forwards/image: "
                     f"{per_forward:.2f} s/forward on {best_n} threads"}
    for key, fn in (("unmodified_reference_in_build_container", "r02_cpu_reference.json"),
                    ("unmodified_reference_end_to_end_configs0", "r03_cpu_reference_e2e.json")):
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as f:
                out[key] = json.load(f)
        except Exception:
            pass
    return out


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


def traffic_ratios(kernel):
    """Measured / algorithmic HBM-side bytes of the dominant kernel per launch shape (profiles/traffic.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        if t["kernel"] != kernel:
            return {}
        return {f"traffic_ratio_{k.lower()}": t["per_launch_bytes"][k] / t["algorithmic_bytes"][k] for k in t["per_launch_bytes"]}
    except Exception:
        return {}


def attainable_probe(lib, dev, smi_index):
    """~1 s of the library's MFMA-only kernel (mmada_mfma_probe: the production MFMA on register-resident RANDOM bf16
    operands, no memory / LDS / barrier traffic) right after the timed region, while the part is as warm as the bench
    left it: the rate a kernel could at best approach on THIS box at the clock it sustains, with the sclk / socket-power
    medians of that second."""
    from mmada_parallel_amd import abi

    nbytes = lib.mmada_mfma_probe_bytes()
    data = (torch.randn(nbytes // 2, device=dev) * 0.5).to(torch.bfloat16)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    tf, ms = C.c_double(), C.c_double()
    torch.cuda.synchronize()
    with SmiSampler(smi_index) as smi:
        abi.check(lib.mmada_mfma_probe(data.data_ptr(), sink.data_ptr(), 32768, 24, abi.stream_ptr(), C.byref(tf), C.byref(ms)),
                  "mmada_mfma_probe")
    return {"tflops": tf.value, "seconds": ms.value * 1e-3, "what": "v_mfma_f32_16x16x32_bf16 only, 8 waves/CU, random bf16 operands "
            "in registers (mmada_mfma_probe), run right after the timed region", "rocm_smi": smi.summary()}


def build_workload(args, cfgnum, model_cls_mod, dev, cfg, tp, world, rank):
    """Returns a dict describing one benchmark step for BASELINE configs[cfgnum]: run(), images per step, FLOPs per image,
    the strings of the JSON line, and what the CPU baseline has to extrapolate to."""
    from mmada_parallel_amd import synth
    from mmada_parallel_amd.generators.parallel_generator import image_step_indices

    full = synth.full_config(cfg)
    # tensor parallel: the library's exchange keeps every row of the last block (no consumed-row window)
    windowed = os.environ.get("MMADA_NO_WINDOW") != "1" and tp == 1
    V, CB = cfg["embedding_size"], synth.CODEBOOK

    def connect(model, max_batch, L):
        if tp > 1:
            model.init_tp_comm(max_batch=max_batch, max_len=L, transport=os.environ.get("MMADA_TP_TRANSPORT", "auto"))
    if cfgnum in (0, 1, 4):
        from mmada_parallel_amd import LLaDAForMultiModalGeneration, generate_ti2ti

        side = 256 if cfgnum == 0 else 512
        B = ((1 if args.scaling == "strong" else tp) if cfgnum in (0, 1) else 16) if args.batch is None else args.batch
        cfg_scale = 0.0 if cfgnum == 1 else 3.0
        sd = synth.synthetic_state_dict(cfg, seed=0, device=str(dev))
        model = LLaDAForMultiModalGeneration.from_state_dict(full, sd, device=dev, tp_rank=rank if tp > 1 else 0,
                                                             tp_size=tp, max_batch=2 * B)
        del sd
        torch.cuda.empty_cache()
        job = synth.synthetic_job(side, side, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
        ids = job["input_ids"].repeat(B, 1).to(dev)
        L = ids.shape[1]
        T, N = job["text_end"] - job["text_start"], job["seq_len"]
        grid = int(round(N ** 0.5))
        state = {}
        if args.temperature != 0:   # inference.py:164-167 creates a device generator when --seed != 0 (README: --seed 42)
            state["generator"] = torch.Generator(device=dev).manual_seed(42)
        connect(model, 2 * B, L)
        # pixels in -> pixels out, as inference.py runs a job (:94-96,127,218-225): the input image is tokenised by the VQ
        # model before the sampler and the sampled codes are decoded after it.  The tokenizer is diffusers' VQModel in the
        # reference (third-party, parity unpinned: DESIGN.md §6b); synthetic weights of the f16 / 8192-code geometry.
        from mmada_parallel_amd import VQModel

        vq_model = VQModel.from_state_dict(synth.VQMODEL_CFG_A, synth.synthetic_vqmodel_state_dict(synth.VQMODEL_CFG_A, 2), device=dev)
        pixels = ((synth.synthetic_image(B, 512, 512, seed=5) + 1.0) * 0.5).clamp(0, 1).to(dev)   # the INPUT image: 512x512 in every config
        row = job["input_ids"][0]
        where = torch.arange(L)
        in_pos = ((row >= synth.TEXT_VOCAB) & (row < synth.TEXT_VOCAB + CB) & (where < job["image_start"])).nonzero()[:, 0].to(dev)
        out_pos = torch.tensor([i for i in range(job["image_start"], job["image_start"] + N + N // job["newline_every"])
                                if int(row[i]) != synth.NEW_LINE])  # generate_ti2ti hands the final ids back on the host
        n_in = int(in_pos.numel())
        assert n_in == 1024 and out_pos.numel() == N

        # tensor parallel: every rank needs every job's ids, but tokenising / decoding all B images on every rank would be
        # replicated work that does not shrink with the rank count — each rank handles its own slice of the jobs and the
        # (tiny: N int64 per job) codes are all-gathered; the decoded pixels stay on the rank that decoded them
        share = tp > 1 and B % tp == 0
        mine = slice(rank * (B // tp), (rank + 1) * (B // tp)) if share else slice(0, B)

        def run():
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]   # read after the timed region (no sync here)
            ev[0].record()
            codes = vq_model.quantize(vq_model.encode(pixels[mine]).latents)[2][2].view(-1, n_in)   # encode_img_with_breaks
            ev[1].record()
            if share:
                import torch.distributed as dist

                if dist.get_backend() == "gloo":   # the one-GPU test rig: gloo gathers host tensors
                    parts = [torch.empty((B // tp, n_in), dtype=codes.dtype) for _ in range(tp)]
                    dist.all_gather(parts, codes.cpu())
                    codes = torch.cat(parts, 0).to(dev)
                else:
                    allc = torch.empty((B, n_in), dtype=codes.dtype, device=dev)
                    dist.all_gather_into_tensor(allc, codes.contiguous())
                    codes = allc
            job_ids = ids.clone()
            job_ids[:, in_pos] = codes + synth.TEXT_VOCAB
            vq, _, final = generate_ti2ti(model, job_ids, job["text_start"], job["text_end"], job["image_start"],
                                          job["seq_len"], job["newline_every"], text_steps=args.text_steps,
                                          timesteps=args.timesteps, temperature=args.temperature, text_temperature=0.0,
                                          cfg_scale=cfg_scale, cfg_img=4.0, uncon_text=job["uncon_text"],
                                          uncon_image=job["uncon_image"], return_state=True,
                                          generator=state.get("generator"))
            # decode_vq_to_image; the one position the schedule leaves masked is a random code in the reference (A.1)
            out_codes = (final[mine][:, out_pos] - synth.TEXT_VOCAB).clamp(0, CB - 1).view(-1, grid, grid).to(dev)
            ev[2].record()
            state["pixels"] = vq_model.decode(out_codes, force_not_quantize=True).sample.clip(0, 1)
            ev[3].record()
            state.setdefault("vq_events", []).append(ev)
            del state["vq_events"][:-64]   # bounded: only the last steps are ever read
            state["final"] = final
            return final

        def launch_probe(graph):
            """Host side of one image: time each step's ENQUEUE (next() of the step generator returns before the GPU has
            run the step — the loop has no host sync) and the whole image's wall time, eager or hipGraph-replayed."""
            from mmada_parallel_amd.generators.parallel_generator import _ti2ti_steps

            gen = _ti2ti_steps(model, ids.clone(), job["text_start"], job["text_end"], job["image_start"], job["seq_len"],
                               job["newline_every"], text_steps=args.text_steps, timesteps=args.timesteps, temperature=0.0,
                               text_temperature=0.0, cfg_scale=cfg_scale, cfg_img=4.0, uncon_text=job["uncon_text"],
                               uncon_image=job["uncon_image"], graph=graph)
            host = {False: [], True: []}
            torch.cuda.synchronize()
            w0 = time.perf_counter()
            while True:
                t0 = time.perf_counter()
                step, _, info = next(gen)
                if step >= args.text_steps:
                    break
                host[bool(info["image_step"])].append(time.perf_counter() - t0)
            torch.cuda.synchronize()
            wall = time.perf_counter() - w0
            med = lambda v: sorted(v)[len(v) // 2] * 1e3 if v else None
            return {"wall_ms_per_image": wall * 1e3, "host_enqueue_ms_text_step_median": med(host[False]),
                    "host_enqueue_ms_image_step_median": med(host[True]),
                    "host_enqueue_ms_per_image": (sum(host[False]) + sum(host[True])) * 1e3,
                    "graph_nodes_per_step_kind": {str(k): v for k, v in getattr(model, "graph_nodes", {}).items()} if graph else None}

        n_img = len(set(image_step_indices(args.text_steps, args.timesteps)))
        fl = job_flops(cfg, L, T, N, args.text_steps, n_img, V, CB, job if windowed else None)
        name = ("BASELINE configs[0]: MMaDA-Parallel-A 8B, 1 prompt, 256x256, timesteps=16, text_steps=32, cfg_img=4.0, "
                f"temperature=0, L={L}, 64 forwards/image (the reference's CPU-runnable case, here on the GPU), pixels in -> "
                "pixels out") if cfgnum == 0 else \
               ("BASELINE configs[1]: MMaDA-Parallel-A 8B, 512x512, timesteps=64, text_steps=128, cfg_img=4.0, "
                f"temperature={args.temperature:g}, L=2438, 256 forwards/image, pixels in -> pixels out (VQ encode + decode inside the step)"
                + (" [SECONDARY LINE: the reference README's sampling settings (temperature 1.0, text_temperature 0, seed 42), not "
                   "the parity configuration the headline is quoted on]" if args.temperature != 0 else "")
                + (f"; STRONG scaling: one job over {tp} tensor-parallel rank(s)" if args.scaling == "strong" else "")) \
            if cfgnum == 1 else \
               (f"BASELINE configs[4] (single-GPU form): MMaDA-Parallel-A 8B editing, batch={B}, 512x512, cfg_scale=3.0 + "
                "cfg_img=4.0 (triple-branch CFG), timesteps=64, text_steps=128, temperature=0, L=2438, 256 sequence-forwards/image")
        metric = "images/sec (256x256, 16 img + 32 text steps) MMaDA-Parallel-A 8B" if cfgnum == 0 else \
                 "images/sec (512x512, 64 img + 128 text steps) MMaDA-Parallel-A 8B" + \
                 ("" if cfgnum == 1 else ", editing mode, cfg_scale=3 + cfg_img=4")
        return dict(model=model, run=run, images_per_step=B, flops_per_image=fl, workload=name, metric=metric, L=L,
                    state=state, launch_probe=launch_probe, cpu=dict(ids=job["input_ids"], text_rows=(job["text_start"], job["text_end"]),
                                          img_rows=(job["image_start"], job["image_start"] + N),
                                          seq_forwards=args.text_steps + 2 * n_img, text_heads=args.text_steps, img_heads=3 * n_img))
    if cfgnum == 3:
        from types import SimpleNamespace

        from mmada_parallel_amd import MAGVITv2, MMadaModelLM
        from mmada_parallel_amd.vq import to_uint8_image

        jobs = 4 if args.batch is None else args.batch
        sd = synth.synthetic_state_dict(cfg, seed=0, device=str(dev))
        model = MMadaModelLM.from_state_dict(full, sd, device=dev, tp_rank=rank if tp > 1 else 0, tp_size=tp, max_batch=2)
        del sd
        torch.cuda.empty_cache()
        vsd = {"encoder." + k: v for k, v in synth.synthetic_vq_state_dict(synth.VQ_ENC_CFG_M, 1).items()}
        vsd.update({"decoder." + k: v for k, v in synth.synthetic_vq_state_dict(synth.VQ_CFG_M, 2).items()})
        vq = MAGVITv2(vsd, device=dev)
        text_vocab = synth.TEXT_VOCAB

        class Tok:
            bos_token_id = 126080

            def __len__(self):
                return text_vocab

        N, T = 1024, 256
        cfgobj = SimpleNamespace(model=SimpleNamespace(mmada=SimpleNamespace(num_vq_tokens=N, codebook_size=CB)),
                                 dataset=SimpleNamespace(preprocessing=SimpleNamespace(max_seq_length=T)))
        imgs = synth.synthetic_image(jobs, 512, 512, seed=5).to(dev)
        g = torch.Generator().manual_seed(3)
        texts = [torch.randint(0, 100000, (40,), generator=g).to(dev) for _ in range(jobs)]
        un_texts = [torch.randint(0, 100000, (40,), generator=g).to(dev) for _ in range(jobs)]
        head = torch.tensor([126340, 126084], device=dev)  # <|interleave|>, <|soi|> stand-ins (M/inference.py:1-14)
        eoi = torch.tensor([126085], device=dev)
        P = 2 + N + 1 + 40
        L = P + 1 + N + 1 + T
        connect(model, 2, L)
        image_steps = 30 if args.timesteps == 64 else args.timesteps
        state = {}

        def run():
            outs = []
            for j in range(jobs):  # the reference's interleave_generate takes ONE prompt (1-D ids): jobs run one after another
                tokens = vq.get_code(imgs[j:j + 1]) + text_vocab                              # M/inference.py:79
                inp = torch.cat([head, tokens[0], eoi, texts[j]])
                unc = torch.cat([head, torch.zeros_like(tokens[0]), eoi, un_texts[j]])
                out_img, out_text = model.interleave_generate(
                    inp, unc, text_cfg=2.5, image_cfg=4.0, text_steps=args.text_steps, image_steps=image_steps,
                    config=cfgobj, reserved_token_mapping={"<|soi|>": 126084, "<|eoi|>": 126085},
                    uni_prompting=SimpleNamespace(text_tokenizer=Tok()))
                outs.append(to_uint8_image(vq.decode_code(out_img)))                          # M/inference.py:127
            state["final"] = torch.cat([o.reshape(1, -1) for o in outs], 0)
            return state["final"]

        n_img = len(set(torch.linspace(args.text_steps // 4, args.text_steps - 1, image_steps).round().int().tolist()))
        fl = m_job_flops(cfg, L, T, N, args.text_steps, n_img, V, CB, P + 1, L - T, windowed)
        name = (f"BASELINE configs[3] (single-GPU form): MMaDA-Parallel-M 8B (MAGVITv2 tokenizer), batch={jobs} edit jobs run "
                f"one after another, 512x512 pixels in -> pixels out (get_code, interleave_generate text_steps=128 "
                f"image_steps=30 text_cfg=2.5 image_cfg=4.0: 128 batch-2 forwards at L={L}, decode_code)")
        ids1 = torch.zeros((1, L), dtype=torch.long)
        return dict(model=model, run=run, images_per_step=jobs, flops_per_image=fl, workload=name, L=L, state=state,
                    metric="images/sec (512x512 edit, 30 img + 128 text steps) MMaDA-Parallel-M 8B",
                    cpu=dict(ids=ids1, text_rows=(L - T, L), img_rows=(P + 1, P + 1 + N), seq_forwards=2 * args.text_steps,
                             text_heads=2 * args.text_steps, img_heads=2 * n_img))
    raise SystemExit(f"--config {cfgnum}: only 0, 1 (headline), 3 (M, batch 4) and 4 (A editing, batch 16) have a single-node form")


def self_launch(n, script=None, argv=None):
    """`python bench.py --gpus N ...` without a launcher: re-run this command under torch.distributed.run with N ranks on
    this node (one rank per GPU over RCCL; rendezvous on 127.0.0.1, a free port).  The ranks' stdout is relayed to stderr
    except rank 0's JSON line, which is printed as the LAST line of stdout; the exit code is the launcher's (non-zero if no
    line came back)."""
    import subprocess

    # the c10d rendezvous binds its own free port (endpoint port 0): no window between picking a port here and the launcher
    # re-binding it, so concurrent benches on one node cannot collide
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--rdzv-backend=c10d",
           "--rdzv-endpoint=127.0.0.1:0", "--local-addr", "127.0.0.1",
           script or os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv))
    env = dict(os.environ, MMADA_BENCH_SELF_LAUNCHED="1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line_out = None
    for line in proc.stdout:
        s = line.strip()
        if s.startswith('{"metric"') and s.endswith("}"):
            line_out = s
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if line_out is not None:
        print(line_out, flush=True)
    return rc if rc else (0 if line_out is not None else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=1, choices=[0, 1, 3, 4],
                    help="index into BASELINE.json configs (1 = the headline, the default the driver runs)")
    ap.add_argument("--batch", type=int, default=None, help="debug only: jobs per step (marks the line reduced)")
    ap.add_argument("--text-steps", type=int, default=None, help="debug only: any value other than the config's own "
                    "(128; config 0: 32) marks the line reduced")
    ap.add_argument("--timesteps", type=int, default=None, help="debug only (config's own: 64; config 0: 16)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="config 1 with N ranks: weak = N jobs per step (BASELINE configs[2]); strong = ONE job over all ranks")
    ap.add_argument("--launch-probe", action="store_true",
                    help="after the timed region: one image eager and one hipGraph-replayed with the host's per-step enqueue time "
                         "(always on with --scaling strong)")
    ap.add_argument("--temperature", type=float, default=0.0, help="image sampling temperature: 0 (default) is the parity "
                    "configuration the headline is quoted on; 1.0 is the reference README's setting (README.md:101-118: "
                    "--temperature 1.0 --text_temperature 0 --cfg_scale 0 --cfg_img 4.0 --seed 42): torch.multinomial + gather + "
                    "randn per image step, no step graph — a SECONDARY line, marked in config.sampling")
    ap.add_argument("--no-probe", action="store_true", help="skip the ~1 s attainable-MFMA probe after the timed region")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (marks the line reduced)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="hipGraph replay of the denoise step (see generators/graph_step.py); auto = MMADA_GRAPH env or off")
    ap.add_argument("--parallelism", choices=["tp", "dp"], default="tp",
                    help="tp (default, BASELINE configs[2]): the model is tensor-parallel over all ranks, batch = N jobs; "
                         "dp: every rank holds the full 16 GB model and runs its own job, no data-path collective")
    args = ap.parse_args()
    own_steps = (32, 16) if args.config == 0 else (128, 64)
    args.text_steps = own_steps[0] if args.text_steps is None else args.text_steps
    args.timesteps = own_steps[1] if args.timesteps is None else args.timesteps
    if args.graph != "auto":
        os.environ["MMADA_GRAPH"] = "1" if args.graph == "on" else "0"

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))   # plain `python bench.py --gpus N`: spawn the N ranks ourselves
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus N` (bench.py starts its own "
                 "ranks) or under torch.distributed.run --nproc-per-node N")
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

    from mmada_parallel_amd import abi, synth

    cfg = dict(synth.CFG_8B)
    if args.layers:
        cfg["n_layers"] = args.layers
    reduced = ((args.text_steps, args.timesteps) != own_steps or args.layers is not None or one_gpu
               or args.batch is not None)
    tp = world if args.parallelism == "tp" else 1
    wl = build_workload(args, args.config, None, dev, cfg, tp, world, rank)
    model, run_once, L = wl["model"], wl["run"], wl["L"]

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
        final_ids = run_once()
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
        dist.all_gather_object(allv, final_ids.tolist())  # all jobs, before the one random fill of the read-out
        ranks_agree = all(a == allv[0] for a in allv) if tp > 1 else None
    # the timed region's own verdict FIRST: a hand-off of the exchange that timed out during the measurement voids the images
    comm_error = None
    if getattr(model, "_comm_in_library", False):
        comm_error = model.comm_status()["error"]
        if use_dist:
            ce = torch.tensor([comm_error], device=dev, dtype=torch.int32)
            dist.all_reduce(ce, op=dist.ReduceOp.MAX)
            comm_error = int(ce.item())
        if comm_error:
            raise SystemExit(f"tensor-parallel exchange reported error {comm_error} (a peer never arrived): no benchmark line")
    # ... then the stand-alone probes (outside the timed region).  A hand-off that times out inside a probe is recorded in the
    # line (`probe_error`); it does not take the measurement away.  (An exception on ONE rank still ends the job: swallowing it
    # would leave the other ranks waiting in the next collective.)
    ar_probe, exposure, probe_error = None, None, None
    if use_dist and tp > 1:
        ar_probe = model.collective_probe(L) if hasattr(model, "collective_probe") else None
        if hasattr(model, "exchange_exposure_probe") and args.config in (0, 1, 4):
            # the conditional forward of one step (batch = the step's jobs) with and without its 2 x n_layers exchanges
            exposure = model.exchange_exposure_probe(wl["cpu"]["ids"].repeat(wl["images_per_step"], 1).to(dev))
        if getattr(model, "_comm_in_library", False) and model.comm_status()["error"]:
            probe_error = "a hand-off timed out inside a stand-alone probe (after the timed region)"
    lprobe = None
    if (args.launch_probe or args.scaling == "strong") and "launch_probe" in wl:
        barrier()
        lprobe = {"eager": wl["launch_probe"](False)}
        barrier()
        if model.graph_capturable():
            lprobe["graph"] = wl["launch_probe"](True)
        barrier()
    probe = None
    if rank == 0 and not args.no_probe:
        probe = attainable_probe(lib, dev, local if not one_gpu else 0)

    vq_ms = None
    if wl["state"].get("vq_events"):   # tokenizer share of the step (A: diffusers VQModel restatement, parity unpinned)
        torch.cuda.synchronize()
        evs = wl["state"]["vq_events"][-args.steps:]
        vq_ms = (sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs), sum(e[2].elapsed_time(e[3]) for e in evs) / len(evs))
    if rank == 0:
        fl_img = wl["flops_per_image"]
        # tp: ONE group holds all jobs of a step; dp: every rank is a replica running its own step
        images = args.steps * wl["images_per_step"] * (world if tp == 1 and world > 1 else 1)
        value = images / dt
        kinds = {}
        for i, nm in enumerate(KIND_NAMES):
            if cnt[i]:
                kinds[nm] = {"launches": cnt[i], "avg_ms": ms[i] / cnt[i], "tflops": fl[i] / (ms[i] * 1e-3) / 1e12}
        dom = max(range(5), key=lambda i: ms[i]) if any(cnt) else 3
        ach = fl[dom] / (ms[dom] * 1e-3) / 1e12 if cnt[dom] else 0.0
        out = {
            "metric": wl["metric"],
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if args.config == 1 and args.scaling == "weak" else "strong", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": wl["workload"] + (" [REDUCED DEBUG RUN]" if reduced else ""),
                       "sampling": {"temperature": args.temperature, "text_temperature": 0.0,
                                    "what": "parity configuration (temperature 0)" if args.temperature == 0 else
                                    "reference README settings: torch.multinomial + gather + randn per image step, no step graph"},
                       "baseline_config_index": args.config,
                       "global_batch": wl["images_per_step"] * (world if tp == 1 and world > 1 else 1), "seq_len": L,
                       "parallelism": f"{args.parallelism}{world}",
                       "text_steps": args.text_steps, "timesteps": args.timesteps, "n_layers": cfg["n_layers"],
                       "algorithmic_pflop_per_image": fl_img / 1e15,
                       "job_mfma_frac": value * fl_img / 1e12 / (world * MFMA_BF16_PEAK_TFLOPS),
                       "vq_encode_ms": vq_ms[0] if vq_ms else None, "vq_decode_ms": vq_ms[1] if vq_ms else None,
                       "vq_note": ("A tokenizer = restatement of diffusers.VQModel (third-party, absent offline): PARITY UNPINNED; "
                                   "its share of the step is vq_encode_ms + vq_decode_ms per step") if vq_ms else None,
                       "hipgraph_step": bool(getattr(model, "graph_replays", 0)),
                       "hipgraph_nodes_per_step_kind": {str(k): v for k, v in getattr(model, "graph_nodes", {}).items()} or None,
                       "tp_ranks_agree": ranks_agree, "tp_collective": getattr(model, "tp_collective", None),
                       # MMADA_TP_TRANSPORT=auto|pull|copy|rccl picks the exchange's data path, MMADA_TP_EXCHANGE_CUS=n its CU partition
                       "tp_exchange_cus": int(lib.mmada_comm_partition(h)) if getattr(model, "_comm_in_library", False) else 0,
                       # ranks of the RCCL communicators that actually exist in this run: torch.distributed's (control
                       # plane, codes all-gather) and the library's own (ncclCommCount; created beside the pull transport
                       # for the probe, or as the data path when the pull transport is not in use)
                       "rccl_nranks": (dist.get_world_size() if use_dist and dist.get_backend() == "nccl" else 0),
                       "library_rccl_nranks": model.rccl_nranks() if hasattr(model, "rccl_nranks") else 0,
                       "launched_by": ("bench.py self-launch -> torch.distributed.run" if os.environ.get("MMADA_BENCH_SELF_LAUNCHED")
                                       else "torch.distributed.run" if "RANK" in os.environ else "single process"),
                       "tp_comm_error": comm_error, "launch_probe": lprobe,
                       "allreduce_probe": ar_probe, "probe_error": probe_error,
                       "exposed_exchange_ms_per_forward": exposure["exposed_exchange_ms_per_forward"] if exposure else None,
                       "exchange_exposure_probe": exposure,
                       "rocm_smi_during_run": smi.summary() if smi else None,
                       "kernels": kinds},
            "roofline": {"bound": "mfma", "kernel": KIND_NAMES[dom], "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
                         "traffic": measured_traffic(KIND_NAMES[dom]) if args.config == 1 else None,
                         "traffic_source": "profiles/traffic.json: separate rocprofv3 --pmc passes of a shortened "
                                           "(--text-steps 8 --timesteps 4) run of this command, not this run"},
        }
        if args.config == 1:
            out["roofline"].update(traffic_ratios(KIND_NAMES[dom]))
        if probe:
            out["roofline"].update({"attainable_tflops": probe["tflops"],
                                    "frac_of_attainable": ach / probe["tflops"] if probe["tflops"] else None,
                                    "job_frac_of_attainable": value * fl_img / 1e12 / (world * probe["tflops"]) if probe["tflops"] else None,
                                    "attainable_probe": probe})
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, **wl["cpu"])
        # native libraries (RCCL's version banner, gloo's connection notes) write to the C stdio buffer of fd 1, which is
        # only flushed at exit when stdout is a pipe: push that out first so that the JSON line is the LAST line of stdout
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
