"""Shared test helpers (seeded inputs identical to oracle/gen_golden.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mmada_parallel_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
STUB_TEXT_VOCAB, STUB_CB = 2048, 512
SAMPLER_CASES = {
    "img4": dict(text_steps=8, timesteps=4, cfg_scale=0.0, cfg_img=4.0),
    "both": dict(text_steps=12, timesteps=6, cfg_scale=2.5, cfg_img=4.0),
    "odd": dict(text_steps=7, timesteps=5, cfg_scale=2.3, cfg_img=0.0),
    "nocfg": dict(text_steps=8, timesteps=8, cfg_scale=0.0, cfg_img=0.0),
}


def tiny_job():
    return synth.synthetic_job(height=64, width=64, text_gen_length=16, prompt_len=8, uncond_prompt_len=4,
                               in_height=64, in_width=64, seed=1)


def stub_logits(seed, call_idx, B, L, V):
    g = torch.Generator().manual_seed(seed * 100003 + call_idx)
    return (torch.randn(B, L, V, generator=g) * 2.0).to(torch.bfloat16)


def from_bits(a):
    return torch.from_numpy(np.ascontiguousarray(a)).view(torch.bfloat16)


def bits(t):
    return t.detach().to("cpu", torch.bfloat16).contiguous().view(torch.int16)


def golden_float(name, prefer=None):
    """Float-GEMM fixtures are recorded once per host ISA class (oracle/gen_golden.py): returns (npz, same_class) where
    same_class says that the file was recorded on a CPU of THIS host's class, i.e. bit-equality with a CPU recomputation
    may be demanded.  `prefer` picks a particular recording (the GPU tests' thresholds are calibrated on one)."""
    isa = synth.host_isa()
    for cand in ([prefer] if prefer else []) + [isa, "amx_bf16", "avx512", "avx512_bf16"]:
        path = os.path.join(GOLDEN, f"{name}.{cand}.npz")
        if os.path.exists(path):
            return np.load(path), cand == isa
    raise FileNotFoundError(f"no recording of {name} under {GOLDEN}")


_SD = {}


def tiny_sd():
    if "tiny" not in _SD:
        _SD["tiny"] = synth.synthetic_state_dict(synth.CFG_TINY, seed=0)
    return _SD["tiny"]

M_CASES = {
    "m_both": dict(text_cfg=1.5, image_cfg=3.5, text_steps=8, image_steps=4, image_temperature=1.0, text_temperature=0.0),
    "m_img": dict(text_cfg=0.0, image_cfg=2.0, text_steps=7, image_steps=7, image_temperature=0.5, text_temperature=0.0),
    "m_noisy": dict(text_cfg=0.7, image_cfg=3.5, text_steps=6, image_steps=3, image_temperature=1.0, text_temperature=0.8),
}
M_SHAPE = dict(prompt=6, N=16, T=16, text_vocab=2048, CB=512, soi=2040, eoi=2041, bos=2042, mask_id=126336)

STEPWISE_CASES = {"sw_img4": dict(text_steps=10, cfg_scale=0.0, cfg_img=4.0), "sw_both": dict(text_steps=14, cfg_scale=2.5, cfg_img=4.0)}


# generate_image (A, text-to-image) stub-logit cases: tests/golden/t2i_traj.npz
T2I_CASES = {
    "g_nocfg_t0": dict(timesteps=6, temperature=0.0, cfg_scale=0.0),
    "g_cfg_t0": dict(timesteps=5, temperature=0.0, cfg_scale=3.0),
    "g_cfg_t1": dict(timesteps=4, temperature=1.0, cfg_scale=2.0),
}


def t2i_job(seed=3, P=8, U=4, side=4):
    """prompt = [P text tokens][2 tokens][side rows of (side masks + newline)][2 tokens]; code_start = P + 2."""
    g = torch.Generator().manual_seed(seed)
    text = torch.randint(0, 1000, (P,), generator=g).tolist()
    body = []
    for _ in range(side):
        body += [synth.MASK] * side + [synth.NEW_LINE]
    prompt = torch.tensor([text + [synth.BOA, synth.BOI] + body + [synth.EOI, synth.EOA]], dtype=torch.long)
    uncon = torch.randint(0, 1000, (1, U), generator=g)
    return dict(prompt=prompt, uncon_ids=uncon, code_start=P + 2, seq_len=side * side, newline_every=side)


class ReplayRng:
    """Replays the reference's torch.rand(shape, dtype=bf16, generator=cpu_generator) draws on the CPU generator."""

    def rand(self, shape, dtype, device, generator):
        return torch.rand(tuple(shape), dtype=dtype, generator=generator).to(device)


# mmu_generate (M text sampler) stub-logit cases: tests/golden/mmu_traj.npz
MMU_CASES = {
    "mmu_plain": dict(max_new_tokens=16, steps=8, block_length=8, temperature=0.0, cfg_scale=0.0),
    "mmu_cfg": dict(max_new_tokens=12, steps=6, block_length=4, temperature=0.0, cfg_scale=1.5),
    "mmu_noisy": dict(max_new_tokens=8, steps=4, block_length=8, temperature=0.6, cfg_scale=0.0),
}
MMU_SHAPE = dict(B=2, P=7, V=2560, mask_id=126336)


# t2i_generate (M text-to-image sampler) stub-logit cases: tests/golden/m_t2i_traj.npz
M_T2I_CASES = {
    "t2i_cfg": dict(B=2, temperature=1.0, timesteps=5, guidance_scale=2.0, uncond=True, known=0),
    "t2i_plain": dict(B=1, temperature=0.7, timesteps=4, guidance_scale=0.0, uncond=False, known=3),
}
M_T2I_SHAPE = dict(P=9, N=16, resolution=6, text_vocab=2048, CB=512, mask_id=126336)


def m_t2i_job(seed, B, known):
    """[P prompt tokens][soi][N image slots][eoi]; `known` leading image slots already hold codebook tokens."""
    sh = M_T2I_SHAPE
    g = torch.Generator().manual_seed(seed)
    prompt = torch.randint(0, 2000, (B, sh["P"]), generator=g)
    unc = torch.randint(0, 2000, (B, sh["P"]), generator=g)
    img = torch.full((B, sh["N"]), sh["mask_id"], dtype=torch.long)
    if known:
        img[:, :known] = torch.randint(0, sh["CB"], (B, known), generator=g) + sh["text_vocab"]
    tail = torch.cat([torch.full((B, 1), 2040), img, torch.full((B, 1), 2041)], dim=1)
    return torch.cat([prompt, tail], dim=1), torch.cat([unc, tail], dim=1)


# generate_ti2ti at temperature > 0 (README default: temperature=1.0, text_temperature=0.7): tests/golden/sampler_noisy.npz
NOISY_CASES = {
    "noisy_img": dict(text_steps=8, timesteps=4, temperature=1.0, text_temperature=0.0, cfg_scale=0.0, cfg_img=4.0),
    "noisy_both": dict(text_steps=6, timesteps=6, temperature=1.0, text_temperature=0.7, cfg_scale=2.0, cfg_img=3.0),
}


class ReplayCpuRng:
    """Draws exactly what the reference draws when it runs on CPU with a seeded CPU generator (torch.rand / randn in the
    tensor's dtype, torch.multinomial), then moves the result to the device the HIP path works on."""

    def rand(self, shape, dtype, device, generator):
        return torch.rand(tuple(shape), dtype=dtype, generator=generator).to(device)

    def randn(self, shape, dtype, device, generator):
        return torch.randn(tuple(shape), dtype=dtype, generator=generator).to(device)

    def multinomial(self, probs2d, generator):
        return torch.multinomial(probs2d.cpu(), 1, generator=generator).to(probs2d.device)


# ---- measured parity numbers (tests/test_gpu_parity_depth.py, test_gpu_fullsize.py) -> gpurun_out/r06_parity.json (copied to profiles/ after a full suite run) ----
PARITY_REPORT = os.path.join(ROOT, "gpurun_out", "r06_parity.json")


def save_parity(section, payload):
    import json

    os.makedirs(os.path.dirname(PARITY_REPORT), exist_ok=True)
    data = {}
    if os.path.exists(PARITY_REPORT):
        try:
            with open(PARITY_REPORT) as f:
                data = json.load(f)
        except Exception:
            data = {}
    data[section] = payload
    with open(PARITY_REPORT, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


def host_threads():
    """Oracle matmuls on the physical cores, not on every hyper-thread (oversubscription made round 1's CPU numbers 3x slow)."""
    n = os.cpu_count() or 8
    torch.set_num_threads(max(1, n if n <= 64 else n // 2))


# painting mode (the output image span starts partly known): tests/golden/paint_traj.npz
PAINT_CASES = {
    "inpaint_img4": ("inpainting", dict(text_steps=8, timesteps=4, cfg_scale=0.0, cfg_img=4.0)),
    "outpaint_both": ("outpainting", dict(text_steps=10, timesteps=5, cfg_scale=2.5, cfg_img=4.0)),
}
TINY_JOB_KW = dict(height=64, width=64, text_gen_length=16, prompt_len=8, uncond_prompt_len=4, in_height=64, in_width=64, seed=1)


def paint_job(kind):
    return synth.paint_job(kind, codebook_size=STUB_CB, text_vocab=STUB_TEXT_VOCAB, **TINY_JOB_KW)


# remasking='random' (text positions ranked by a uniform draw from the GLOBAL RNG): tests/golden/random_traj.npz
RANDOM_CASES = {
    "rand_img4": dict(text_steps=8, timesteps=4, cfg_scale=0.0, cfg_img=4.0, temperature=0.0, text_temperature=0.0),
    "rand_both": dict(text_steps=9, timesteps=3, cfg_scale=2.0, cfg_img=4.0, temperature=0.0, text_temperature=0.0),
}
RANDOM_SEED = 4321


# edge cases of generate_ti2ti (tests/golden/edge_traj.npz): name -> (job tweak, uncon_text given, uncon_image given, kwargs)
EDGE_CASES = {
    # CFG scales > 0 but no unconditional prompts: the reference substitutes ZERO logits for both branches (:275-278)
    "cfg_without_uncond": ("plain", False, False, dict(text_steps=6, timesteps=3, cfg_scale=1.5, cfg_img=4.0)),
    # only the image branch has an unconditional prompt: both forwards still run (:243), the text one on the plain ids
    "only_uncon_image": ("plain", False, True, dict(text_steps=6, timesteps=3, cfg_scale=2.0, cfg_img=3.0)),
    # the text span is already complete: no text step ever runs (:183), image steps still do
    "text_done": ("text_done", True, True, dict(text_steps=5, timesteps=5, cfg_scale=0.0, cfg_img=4.0)),
    # more image steps requested than steps exist: linspace repeats indices, `step in list` runs each once
    "more_timesteps_than_steps": ("plain", True, True, dict(text_steps=4, timesteps=9, cfg_scale=0.0, cfg_img=4.0)),
    # a single step: the image step coincides with the only text step, ratio = 1 -> one cell left masked
    "single_step": ("plain", True, True, dict(text_steps=1, timesteps=1, cfg_scale=0.0, cfg_img=4.0)),
}


def edge_job(name):
    tweak, ut, ui, kw = EDGE_CASES[name]
    job = tiny_job()
    if tweak == "text_done":
        g = torch.Generator().manual_seed(77)
        ids = job["input_ids"].clone()
        ids[0, job["text_start"]:job["text_end"]] = torch.randint(0, 1000, (job["text_end"] - job["text_start"],), generator=g)
        job["input_ids"] = ids
    if not ut:
        job["uncon_text"] = None
    if not ui:
        job["uncon_image"] = None
    return job, kw


class FakeVq:
    """CPU stand-in with the surface utils/image_utils.py touches: latents = 2x2 mean of the red channel, index = its
    value quantised to 0..63 (so token arithmetic and mask geometry can be checked without a GPU)."""

    def __init__(self):
        from types import SimpleNamespace

        self.config = SimpleNamespace(block_out_channels=[1, 1], latent_channels=1)
        self.device = torch.device("cpu")

    def encode(self, x):
        from types import SimpleNamespace

        return SimpleNamespace(latents=torch.nn.functional.avg_pool2d(x[:, :1], 2))

    def quantize(self, latents):
        return None, None, (None, None, (latents.reshape(-1) * 63).round().long())

    def decode(self, codes, force_not_quantize=False, shape=None):
        from types import SimpleNamespace

        assert force_not_quantize and tuple(shape) == (codes.shape[0], codes.shape[1], codes.shape[2], 1)
        g = codes.float() / 63.0 * 1.2 - 0.1                               # leaves [0, 1] on both sides: the clip matters
        x = torch.stack([g, 1.0 - g, g * g], 1)
        return SimpleNamespace(sample=torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest"))


# utils/image_utils.py against the reference's own functions (tests/golden/image_utils_tokens.npz)
PAINT_UTIL_CASES = {
    "inpaint": dict(mask_h_ratio=0.5, mask_w_ratio=0.25, mask_mode="inpainting"),
    "outpaint": dict(mask_h_ratio=0.5, mask_w_ratio=0.25, mask_mode="outpainting"),
    "inpaint_dilated": dict(mask_h_ratio=0.4, mask_w_ratio=0.3, dilate_latent_k=1, mask_mode="inpainting"),
    "inpaint_nearest": dict(mask_h_ratio=0.37, mask_w_ratio=0.21, downsample_mode="nearest", mask_mode="inpainting"),
    "outpaint_bilinear": dict(mask_h_ratio=0.55, mask_w_ratio=0.45, downsample_mode="bilinear", gray_value=90, mask_mode="outpainting"),
}


def paint_util_image():
    rng = np.random.default_rng(5)
    return rng.integers(0, 256, size=(36, 70, 3), dtype=np.uint8)       # resized to 36 x 70 -> multiples of 2: unchanged size


# ---- in-process tensor-parallel rank groups (tests/test_gpu_tp.py, tests/test_gpu_parity_depth.py) ----
_TP_STREAMS = []


def tp_group(cfg_base, sd, tp, max_rows, dev="cuda:0", transport="pull", exchange_cus=0):
    """The ranks of a tensor-parallel group as separate handles of ONE process, each on its own stream, connected with
    mmada_comm_connect_local (tests/test_gpu_tp.py explains why this is the multi-device code path unchanged)."""
    import ctypes as C

    from mmada_parallel_amd import LLaDAForMultiModalGeneration, abi

    cfg = synth.full_config(cfg_base)
    ranks = [LLaDAForMultiModalGeneration.from_state_dict(cfg, sd, device=dev, tp_rank=r, tp_size=tp) for r in range(tp)]
    lib = ranks[0]._lib
    for m in ranks:
        abi.check(lib.mmada_comm_create(m._handle, max_rows, None), "comm_create")
        m._comm_rows = max_rows
    arr = (C.c_void_p * tp)(*[m._handle.value for m in ranks])
    for m in ranks:
        abi.check(lib.mmada_comm_connect_local(m._handle, arr), "connect_local")
        m._comm_in_library, m.tp_collective = True, transport
        if transport == "copy":   # the same mapped buffers, bytes moved by the copy engines (csrc/tp_comm.hip mode 4)
            abi.check(lib.mmada_comm_set_mode(m._handle, 4), "set_mode")
        if exchange_cus:
            abi.check(lib.mmada_comm_set_partition(m._handle, exchange_cus), "set_partition")
    # One pool of compute streams for every group this process ever builds: a rank's wait kernel spins until its peers'
    # launches run, so no two live streams may share a hardware queue (GPU_MAX_HW_QUEUES, tests/conftest.py) — fresh
    # streams per group would walk through the queues and end up doubling up (seen as hand-off timeouts, status.error).
    while len(_TP_STREAMS) < tp:
        _TP_STREAMS.append(torch.cuda.Stream(device=dev))
    return ranks, _TP_STREAMS[:tp]


def tp_each(ranks, streams, fn):
    """fn(rank_model) enqueued for every rank on its own stream; NO host sync until all are enqueued."""
    out = []
    cur = torch.cuda.current_stream()
    for m, s in zip(ranks, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            out.append(fn(m))
    for s in streams:
        cur.wait_stream(s)
    torch.cuda.synchronize()
    return out
