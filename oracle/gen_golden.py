"""Generate tests/golden/*.npz by importing and RUNNING the reference itself (CPU, this container only).

    python oracle/gen_golden.py            # needs /root/reference (read-only); never run on the GPU box

The reference has no tests / golden vectors of its own, so these fixtures — outputs of the
unmodified reference code on seeded synthetic inputs — are what pins the oracle (tests/test_oracle_golden.py)
and, through it, the HIP path.  Nothing here is copied from the reference; it is imported via sys.path.

Fixtures  (forward_tiny / e2e_tiny contain the outputs of CPU bf16 GEMMs, which are bit-reproducible only on CPUs of the
same ISA class — AMX, AVX512-BF16 and plain AVX-512 hosts take different oneDNN kernels.  They are stored
once per class, <name>.<synth.host_isa()>.npz; this script writes the file of the host it runs on and leaves the others.)
  forward_tiny.*.npz LLaDAForMultiModalGeneration(tiny cfg, synthetic weights, bf16)(ids): hidden states after each
                     block, logits slices (image rows x codebook columns, text rows x first 4096 columns, per-row
                     top-8) — model/modeling_xllmx_dimoo.py:41-72 + model/modeling_llada.py:1201-1415
  sampler_traj.npz   generate_ti2ti driven by a STUB model that returns seeded random bf16 logits: the ids the
                     sampler passes to every model call + final outputs — generators/parallel_generator.py:102-368
  paint_traj.npz     the same in painting mode: the output image span starts partly known (in- / out-painting rectangle)
  random_traj.npz    the same with remasking='random' (uniform draws from the global CPU generator rank the text positions)
  edge_traj.npz      edge cases: CFG scales without unconditional prompts, one prompt only, complete text span, more image
                     steps than steps, a single step
  image_utils_tokens.npz  the reference's encode_img_with_breaks / encode_img_with_paint (utils/image_utils.py:159-284) on a
                     fake tokenizer, imported with a stub `diffusers`
  sampler_noisy.npz  the same at temperature 1.0 / text_temperature 0.7 with every draw taken from a seeded CPU generator
  e2e_tiny.*.npz     generate_ti2ti with the real tiny model at temperature 0: ids at every model call + outputs
  peaked_traj.*.npz  generate_ti2ti free-running on the PEAKED synthetic checkpoint (synth.synthetic_state_dict_peaked, 4 blocks,
                     d = 1024) at BASELINE configs[0] geometry (L = 1654, 32 text + 16 image steps, 64 model calls): ids of every
                     call + outputs — the trajectory a re-ordered GEMM can be held to (decision margins >> bf16 noise)
  m_peaked_traj.*.npz  MMaDA-Parallel-M: MMadaModelLM.interleave_generate free-running on the M tree's own LLaDAModelLM with the
                     peaked checkpoint at configs[3] geometry (L = 2349, batch-2 forwards, text_cfg 2.5, image_cfg 4): ids of every
                     call + outputs — models/modeling_mmada.py:117-248, models/modeling_llada.py
  dllm_cache.*.npz   LLaDAModelLM.forward(use_cache=True, to_compute_mask=..., cat=...) on the tiny model: a prime call and
                     compute-mask steps on changed ids, two cache keys, with and without caching(True): logit slices +
                     arg-max of the returned logit cache — model/modeling_llada.py:593-600,929-940,1244-1245,1406-1426
  logconf_table.npy  torch.log(p + 1e-10) in bf16 for every non-negative bf16 p (parallel_generator.py:36)
  t2i_traj.npz       generate_image (A text-to-image MaskGIT sampler) driven by the same kind of stub model, with and
                     without CFG, temperature 0 and 1 (seeded CPU generator) — generators/image_generation_generator.py
  mmu_traj.npz       MMaDA-Parallel-M MMadaModelLM.mmu_generate (block-wise text sampler) on stub logits, B = 2, with
                     and without CFG, temperature 0 and 0.6 — models/modeling_mmada.py:618-692
  m_t2i_traj.npz     MMaDA-Parallel-M MMadaModelLM.t2i_generate (MaskGIT text-to-image) on stub logits with replayed
                     multinomial / uniform draws, with and without guidance — models/modeling_mmada.py:264-359
  vq_decode.npz      MMaDA-Parallel-M MAGVITv2.decode_code (LFQuantizer.get_codebook_entry + VQGANDecoder, fp32) on
                     seeded synthetic decoder weights: a 2-level decoder (full output) and the default 5-level
                     decoder at 32x32 codes -> 512x512 (every 4th pixel) — models/modeling_magvitv2.py:208-221,277-433
  vq_encode.npz      MAGVITv2.get_code (VQGANEncoder + LFQuantizer sign quantisation / get_indices) on seeded weights and
                     a seeded synthetic image: indices + pre-quantisation z — models/modeling_magvitv2.py:62-171,201-206,422-427
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/MMaDA-Parallel-A"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from mmada_parallel_amd import synth  # noqa: E402  (pure-python helpers: shapes + seeded inputs)

OUT = os.path.join(REPO, "tests", "golden")


def bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.bfloat16).contiguous().view(torch.int16).numpy()


def tiny_job():
    return synth.synthetic_job(height=64, width=64, text_gen_length=16, prompt_len=8, uncond_prompt_len=4,
                               in_height=64, in_width=64, seed=1)


def build_reference_model(cfg: dict, sd: dict):
    from model import LLaDAForMultiModalGeneration
    from model.configuration_llada import LLaDAConfig

    with contextlib.redirect_stdout(io.StringIO()):
        m = LLaDAForMultiModalGeneration(LLaDAConfig(**synth.full_config(cfg)))
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    return m.to(torch.bfloat16).eval()


class Recorder:
    """Wraps a model: records the ids of every call (the sampler's observable state)."""

    def __init__(self, fn):
        self.fn, self.calls = fn, []

    def __call__(self, ids, infer=True, use_cache=False):
        self.calls.append(ids.clone())
        return self.fn(ids, len(self.calls))


def stub_logits(seed: int, call_idx: int, B: int, L: int, V: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed * 100003 + call_idx)
    return (torch.randn(B, L, V, generator=g) * 2.0).to(torch.bfloat16)


def compute_forward() -> dict:
    """The unmodified reference forward on the tiny model, ON THIS HOST (also called live by tests/test_oracle_golden.py
    when /root/reference is mounted: the oracle must reproduce it bit for bit whatever CPU this is)."""
    from model.modeling_llada import LLaDAModel  # noqa: F401  (import check)

    cfg = synth.CFG_TINY
    sd = synthetic_sd()
    model = build_reference_model(cfg, sd)
    job = tiny_job()
    ids = job["input_ids"]
    taps = []
    hooks = [blk.register_forward_hook(lambda _m, _i, o: taps.append(o[0].detach().clone()))
             for blk in model.model.transformer.blocks]
    with torch.no_grad():
        logits = model(ids, infer=True, use_cache=False).logits
    for h in hooks:
        h.remove()
    ts, te = job["text_start"], job["text_end"]
    pos = [i for i in range(job["image_start"], job["image_start"] + job["seq_len"] + job["seq_len"] // job["newline_every"])
           if int(ids[0, i]) != synth.NEW_LINE]
    top = torch.topk(logits[0].float(), 8, dim=-1)
    return dict(
        ids=ids.numpy(), hidden=np.stack([bits(t[0]) for t in taps]),
        img_logits=bits(logits[0, pos, synth.TEXT_VOCAB:synth.TEXT_VOCAB + synth.CODEBOOK]),
        text_logits_head=bits(logits[0, ts:te, :4096]),
        top8_val=bits(top.values.to(torch.bfloat16)), top8_idx=top.indices.numpy().astype(np.int32),
        argmax=logits[0].argmax(-1).numpy().astype(np.int32), pos=np.array(pos, np.int32),
    )


def gen_forward():
    # float GEMM outputs: one file per host ISA class (synth.host_isa), see the module docstring
    d = compute_forward()
    np.savez_compressed(os.path.join(OUT, f"forward_tiny.{synth.host_isa()}.npz"), **d)
    print(f"forward_tiny.{synth.host_isa()}: L =", d["ids"].shape[1], "hidden", d["hidden"].shape)


_SD = None


def synthetic_sd():
    global _SD
    if _SD is None:
        _SD = synth.synthetic_state_dict(synth.CFG_TINY, seed=0)
    return _SD


def run_reference_sampler(model, job, generator=None, **kw):
    from generators.parallel_generator import generate_ti2ti

    rec = Recorder(model)
    torch.manual_seed(1234)  # pins the one torch.randint fill and the randn draws (A.2)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        vq, text = generate_ti2ti(rec, job["input_ids"], job["text_start"], job["text_end"], job["image_start"],
                                  job["seq_len"], job["newline_every"], uncon_text=job["uncon_text"],
                                  uncon_image=job["uncon_image"], tokenizer=None, generator=generator, **kw)
    return rec.calls, vq, text


SAMPLER_CASES = {
    # name: (V_text, CB, kwargs)
    "img4": dict(text_steps=8, timesteps=4, temperature=0.0, text_temperature=0.0, cfg_scale=0.0, cfg_img=4.0),
    "both": dict(text_steps=12, timesteps=6, temperature=0.0, text_temperature=0.0, cfg_scale=2.5, cfg_img=4.0),
    "odd": dict(text_steps=7, timesteps=5, temperature=0.0, text_temperature=0.0, cfg_scale=2.3, cfg_img=0.0),
    "nocfg": dict(text_steps=8, timesteps=8, temperature=0.0, text_temperature=0.0, cfg_scale=0.0, cfg_img=0.0),
}
STUB_TEXT_VOCAB, STUB_CB = 2048, 512


def gen_sampler_traj():
    job = tiny_job()
    # remap the job onto a small synthetic vocabulary: only MASK / NEW_LINE identities matter to the sampler
    L = job["input_ids"].shape[1]
    V = STUB_TEXT_VOCAB + STUB_CB
    out = {}
    for ci, (name, kw) in enumerate(SAMPLER_CASES.items()):
        seed = 7 + ci

        def fn(ids, call_idx, seed=seed):
            return SimpleNamespace(logits=stub_logits(seed, call_idx, ids.shape[0], ids.shape[1], V))

        calls, vq, text = run_reference_sampler(fn, job, text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB, **kw)
        out[name + "_calls"] = torch.cat(calls, 0).numpy()
        out[name + "_vq"] = np.array(vq, np.int64)
        out[name + "_text"] = np.array(text, np.int64)
        out[name + "_seed"] = np.array(seed)
        print(f"sampler_traj[{name}]: {len(calls)} model calls, {len(text)} text tokens")
    np.savez_compressed(os.path.join(OUT, "sampler_traj.npz"), **out)


def gen_paint_traj():
    """generate_ti2ti in painting mode: the output image span starts partly known (inference.py:141-146), so the image
    branch's unknown count starts below N; stub logits, temperature 0."""
    from tests.helpers import PAINT_CASES, paint_job

    V = STUB_TEXT_VOCAB + STUB_CB
    out = {}
    for ci, (name, (kind, kw)) in enumerate(PAINT_CASES.items()):
        seed = 61 + ci
        job = paint_job(kind)

        def fn(ids, call_idx, seed=seed):
            return SimpleNamespace(logits=stub_logits(seed, call_idx, ids.shape[0], ids.shape[1], V))

        calls, vq, text = run_reference_sampler(fn, job, temperature=0.0, text_temperature=0.0,
                                                text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB, **kw)
        out[name + "_calls"] = torch.cat(calls, 0).numpy()
        out[name + "_vq"] = np.array(vq, np.int64)
        out[name + "_text"] = np.array(text, np.int64)
        out[name + "_seed"] = np.array(seed)
        span = job["input_ids"][0, job["image_start"]:job["image_start"] + job["seq_len"] + job["seq_len"] // job["newline_every"]]
        n_known = int(((span != synth.MASK) & (span != synth.NEW_LINE)).sum())
        print(f"paint_traj[{name}]: {len(calls)} model calls, {n_known} of {job['seq_len']} output cells known at the start")
    np.savez_compressed(os.path.join(OUT, "paint_traj.npz"), **out)


def gen_random_traj():
    """generate_ti2ti with remasking='random' and generator=None (the only working form): the text positions
    to unmask are ranked by torch.rand draws from the GLOBAL CPU generator, which mask_by_random_topk's randn also advances."""
    from tests.helpers import RANDOM_CASES, RANDOM_SEED
    from generators.parallel_generator import generate_ti2ti

    job = tiny_job()
    V = STUB_TEXT_VOCAB + STUB_CB
    out = {}
    for ci, (name, kw) in enumerate(RANDOM_CASES.items()):
        seed = 91 + ci
        rec = Recorder(lambda ids, call_idx, seed=seed: SimpleNamespace(
            logits=stub_logits(seed, call_idx, ids.shape[0], ids.shape[1], V)))
        torch.manual_seed(RANDOM_SEED)
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            vq, text = generate_ti2ti(rec, job["input_ids"], job["text_start"], job["text_end"], job["image_start"],
                                      job["seq_len"], job["newline_every"], uncon_text=job["uncon_text"],
                                      uncon_image=job["uncon_image"], tokenizer=None, generator=None, remasking="random",
                                      text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB, **kw)
        out[name + "_calls"] = torch.cat(rec.calls, 0).numpy()
        out[name + "_text"] = np.array(text, np.int64)
        out[name + "_seed"] = np.array(seed)
        print(f"random_traj[{name}]: {len(rec.calls)} model calls, {len(text)} text tokens")
    np.savez_compressed(os.path.join(OUT, "random_traj.npz"), **out)


def gen_edge_traj():
    """Edge cases of generate_ti2ti (tests/helpers.py EDGE_CASES): CFG scales without unconditional prompts (zero logits),
    one prompt only, a text span that is already complete, more image steps than steps, a single step."""
    from tests.helpers import EDGE_CASES, edge_job

    V = STUB_TEXT_VOCAB + STUB_CB
    out = {}
    for ci, name in enumerate(EDGE_CASES):
        seed = 211 + ci
        job, kw = edge_job(name)

        def fn(ids, call_idx, seed=seed):
            return SimpleNamespace(logits=stub_logits(seed, call_idx, ids.shape[0], ids.shape[1], V))

        calls, vq, text = run_reference_sampler(fn, job, temperature=0.0, text_temperature=0.0,
                                                text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB, **kw)
        out[name + "_calls"] = torch.cat(calls, 0).numpy()
        out[name + "_vq"] = np.array(vq, np.int64)
        out[name + "_text"] = np.array(text, np.int64)
        out[name + "_seed"] = np.array(seed)
        print(f"edge_traj[{name}]: {len(calls)} model calls, {len(text)} text tokens")
    np.savez_compressed(os.path.join(OUT, "edge_traj.npz"), **out)


def gen_image_utils():
    """The reference's own utils/image_utils.py functions around the tokenizer.  The module imports `diffusers` at the top
    (VQModel, VaeImageProcessor), which is not installed: a stub module supplies them — VQModel is only a type annotation
    there, and VaeImageProcessor(vae_scale_factor, do_normalize=False).preprocess is the PIL + numpy restatement of
    mmada_parallel_amd/utils/image_utils.py (the one thing this fixture cannot pin).  Everything else — token layout, special
    ids, the in- / out-painting mask geometry, dilation, the three mask down-samplers — is the reference's code."""
    import types

    from PIL import Image

    from mmada_parallel_amd.utils import image_utils as mine
    from tests.helpers import PAINT_UTIL_CASES, FakeVq, paint_util_image

    class VaeImageProcessor:
        def __init__(self, vae_scale_factor=8, do_normalize=True, **_):
            assert not do_normalize
            self.f = vae_scale_factor

        def preprocess(self, img):
            return mine.pil_to_unit_tensor(img, self.f)

        def postprocess(self, x, output_type="pil"):
            return mine.unit_tensor_to_pil(x)

    stub = types.ModuleType("diffusers")
    stub.VQModel = object
    sub = types.ModuleType("diffusers.image_processor")
    sub.VaeImageProcessor = VaeImageProcessor
    stub.image_processor = sub
    saved = {k: sys.modules.get(k) for k in ("diffusers", "diffusers.image_processor", "utils", "utils.image_utils")}
    sys.modules["diffusers"], sys.modules["diffusers.image_processor"] = stub, sub
    for k in ("utils", "utils.image_utils"):
        sys.modules.pop(k, None)
    try:
        import importlib

        ref = importlib.import_module("utils.image_utils")
        img, vq = Image.fromarray(paint_util_image()), FakeVq()
        out = {"breaks": np.array(ref.encode_img_with_breaks(img, vq, vae_scale_factor=2), np.int64)}
        for name, kw in PAINT_UTIL_CASES.items():
            toks, vis = ref.encode_img_with_paint(img, vq, **kw)
            out[name + "_tokens"] = np.array(toks, np.int64)
            out[name + "_vis"] = np.asarray(vis)
        codes = torch.arange(18 * 35).view(1, -1) % 64
        out["decoded"] = np.asarray(ref.decode_vq_to_image(codes, None, None, 36, 70, vq))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    np.savez_compressed(os.path.join(OUT, "image_utils_tokens.npz"), **out)
    print("image_utils_tokens:", {k: v.shape for k, v in out.items() if k.endswith("_tokens") or k == "breaks"})


def gen_sampler_noisy():
    """generate_ti2ti at temperature > 0 (the README's defaults are temperature 1.0 / text_temperature 0.7): stub logits, all
    random draws from a seeded CPU generator handed to the reference (`generator=`)."""
    from tests.helpers import NOISY_CASES

    job = tiny_job()
    V = STUB_TEXT_VOCAB + STUB_CB
    out = {}
    for ci, (name, kw) in enumerate(NOISY_CASES.items()):
        seed = 131 + ci

        def fn(ids, call_idx, seed=seed):
            return SimpleNamespace(logits=stub_logits(seed, call_idx, ids.shape[0], ids.shape[1], V))

        gen = torch.Generator().manual_seed(500 + ci)
        calls, vq, text = run_reference_sampler(fn, job, generator=gen, text_vocab_size=STUB_TEXT_VOCAB,
                                                codebook_size=STUB_CB, **kw)
        out[name + "_calls"] = torch.cat(calls, 0).numpy()
        out[name + "_vq"] = np.array(vq, np.int64)
        out[name + "_text"] = np.array(text, np.int64)
        out[name + "_seed"] = np.array(seed)
        out[name + "_gen_seed"] = np.array(500 + ci)
        print(f"sampler_noisy[{name}]: {len(calls)} model calls, {len(text)} text tokens")
    np.savez_compressed(os.path.join(OUT, "sampler_noisy.npz"), **out)


def compute_e2e() -> dict:
    cfg = synth.CFG_TINY
    model = build_reference_model(cfg, synthetic_sd())
    job = tiny_job()

    def fn(ids, _idx):
        with torch.no_grad():
            return model(ids, infer=True, use_cache=False)

    kw = dict(text_steps=8, timesteps=4, temperature=0.0, text_temperature=0.0, cfg_scale=0.0, cfg_img=4.0)
    calls, vq, text = run_reference_sampler(fn, job, **kw)
    return dict(calls=torch.cat(calls, 0).numpy(), vq=np.array(vq, np.int64), text=np.array(text, np.int64))


PEAKED_KW = dict(text_steps=32, timesteps=16, temperature=0.0, text_temperature=0.0, cfg_scale=0.0, cfg_img=4.0)


def peaked_job():
    """BASELINE configs[0] geometry: 256x256 output, 512x512 conditioning image, 256 text tokens: L = 1654."""
    return synth.synthetic_job(256, 256, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)


def compute_peaked() -> dict:
    """The unmodified reference — generate_ti2ti + LLaDAForMultiModalGeneration (4 blocks, d = 1024, bf16, CPU) — free-running
    on the PEAKED synthetic checkpoint (synth.synthetic_state_dict_peaked: decisions far above bf16 noise) at configs[0]
    geometry and schedule: 32 text + 16 image steps, 64 model calls.  Records the ids of every model call, the outputs, and
    per model call the smallest top-1 / top-2 margin (in logit sigmas) among the still-masked text rows (for the report)."""
    cfg = synth.CFG_PEAKED
    job = peaked_job()
    model = build_reference_model(cfg, synth.synthetic_state_dict_peaked(cfg, synth.peaked_delta(job)))
    margins = []

    def fn(ids, _idx):
        with torch.no_grad():
            out = model(ids, infer=True, use_cache=False)
        masked = ids[0, job["text_start"]:job["text_end"]] == synth.MASK   # the rows a text step can still commit
        if bool(masked.any()):
            lg = out.logits[0, job["text_start"]:job["text_end"]][masked].float()
            top = lg.topk(2, -1).values
            margins.append(((top[:, 0] - top[:, 1]) / lg.std(-1)).min().item())
        else:
            margins.append(float("inf"))
        return out

    # the confidences every re-mask cut is taken on (mask_by_random_topk's `probs` argument, bf16): torch.sort leaves the
    # order of exact ties unspecified (oracle/generate_oracle.py: tie_order), so a comparison has to know where the ties are
    import generators.parallel_generator as pgen

    commits, real = [], pgen.mask_by_random_topk

    def spy(mask_len, probs, temperature=1.0, generator=None):
        m = real(mask_len, probs, temperature, generator)
        commits.append((bits(probs[0]), m[0].numpy().copy(), int(mask_len.reshape(-1)[0])))
        return m

    pgen.mask_by_random_topk = spy
    try:
        calls, vq, text = run_reference_sampler(fn, job, **PEAKED_KW)
    finally:
        pgen.mask_by_random_topk = real
    return dict(calls=torch.cat(calls, 0).numpy().astype(np.int32), vq=np.array(vq, np.int64), text=np.array(text, np.int64),
                min_text_margin_sigma=np.array(margins, np.float32),
                commit_conf=np.stack([c[0] for c in commits]), commit_masking=np.stack([c[1] for c in commits]),
                commit_mask_len=np.array([c[2] for c in commits], np.int32))


def gen_peaked():
    d = compute_peaked()
    np.savez_compressed(os.path.join(OUT, f"peaked_traj.{synth.host_isa()}.npz"), **d)
    print(f"peaked_traj.{synth.host_isa()}: {d['calls'].shape[0]} model calls of L = {d['calls'].shape[1]}; "
          f"{len(set(d['vq'].tolist()))} distinct VQ ids, {len(set(d['text'].tolist()))} distinct text ids; "
          f"smallest top-1/top-2 text margin over all calls {d['min_text_margin_sigma'].min():.2f} sigma")


def gen_e2e():
    d = compute_e2e()
    np.savez_compressed(os.path.join(OUT, f"e2e_tiny.{synth.host_isa()}.npz"), **d)
    print(f"e2e_tiny.{synth.host_isa()}: {d['calls'].shape[0]} model calls; vq[:8]={d['vq'][:8].tolist()}")


# ---- dLLM cache: LLaDAModelLM.forward(use_cache=True, to_compute_mask=..., cat=...) on the tiny model -----------------------
def compute_dllm_cache() -> dict:
    """Runs the unmodified reference: caching(True), then the script; also the same first two calls with the blocks'
    use_cache flag left off (queries then take the LAST Tc rotary positions, modeling_llada.py:421-425)."""
    from model.modeling_llada import LLaDAModelLM

    model = build_reference_model(synth.CFG_TINY, synthetic_sd())
    out = {}

    def run(tag, script):
        for n, (cat, ids, m) in enumerate(script):
            with torch.no_grad():
                lg = LLaDAModelLM.forward(model, input_ids=ids, use_cache=True, to_compute_mask=m, cat=cat).logits
            out[f"{tag}{n}_argmax"] = lg[0].argmax(-1).numpy().astype(np.int32)
            out[f"{tag}{n}_img"] = bits(lg[0, :, synth.TEXT_VOCAB:synth.TEXT_VOCAB + 256])
            out[f"{tag}{n}_txt"] = bits(lg[0, :, :256])

    model.caching(True)
    run("on", synth.dllm_cache_script())
    model.caching(False)          # clears the caches too (:598-600)
    run("off", synth.dllm_cache_script()[:2])
    model.empty_cache()
    return out


def gen_dllm_cache():
    d = compute_dllm_cache()
    np.savez_compressed(os.path.join(OUT, f"dllm_cache.{synth.host_isa()}.npz"), **d)
    print(f"dllm_cache.{synth.host_isa()}: {len(d) // 3} calls")


# ---- M variant end to end: the reference's interleave_generate on the reference's own LLaDAModelLM (peaked checkpoint) ----
def _import_m_models():
    """MMaDA-Parallel-M/models as a package WITHOUT its __init__ (which imports files that do not exist)."""
    import importlib
    import types

    pkg = types.ModuleType("models")
    pkg.__path__ = [M_REF + "/models"]
    saved = sys.modules.get("models")
    sys.modules["models"] = pkg
    return importlib.import_module("models.modeling_mmada"), importlib.import_module("models.modeling_llada"), \
        importlib.import_module("models.configuration_llada"), saved


def compute_m_peaked() -> dict:
    """The UNMODIFIED MMaDA-Parallel-M sampler (MMadaModelLM.interleave_generate, models/modeling_mmada.py:117-248) free-running
    on the UNMODIFIED M denoiser (models/modeling_llada.py LLaDAModelLM, 4 blocks, d = 1024, bf16, CPU) with the peaked
    synthetic checkpoint at BASELINE configs[3] geometry: L = 2349, batch 2 (cond || uncond) every step, text_cfg 2.5,
    image_cfg 4, 24 text steps of which 8 are image steps.  torch.multinomial / Tensor.uniform_ are served by per-call seeded
    generators (SeededRng) that the oracle and the GPU test replay.  Records the ids of every model call and the outputs, and
    per call the largest fp64 text confidence among the still-masked rows (1 - p must stay > 0: no exact ties for topk)."""
    from unittest import mock

    from oracle.interleave_oracle import SeededRng

    mm, ml, mc, saved = _import_m_models()
    try:
        cfg, job, kw = synth.CFG_PEAKED, synth.m_peaked_job(), dict(synth.M_PEAKED_KW)
        sd = synth.synthetic_state_dict_peaked(cfg, job["delta"], beta=synth.M_PEAKED_BETA)
        full = dict(synth.full_config(cfg), mask_token_id=synth.MASK)
        with contextlib.redirect_stdout(io.StringIO()):
            model = ml.LLaDAModelLM(mc.LLaDAConfig(**full))
        # environment shim, not a change of the reference: LLaDAConfig hands use_cache=False to PretrainedConfig.__init__, which
        # the pinned transformers 4.46.2 stores as config.use_cache (forward reads it, models/modeling_llada.py:1407-1408) and
        # the transformers 5.x of this container drops
        model.config.use_cache = False
        model.load_state_dict(sd, strict=True)
        model = model.to(torch.bfloat16).eval()
        calls, top_conf = [], []
        ts = job["text_start"]

        class Self:   # what interleave_generate touches of `self`: __call__ and config.mask_token_id
            config = SimpleNamespace(mask_token_id=synth.MASK)

            def __call__(self, ids):
                calls.append(ids.clone())
                with torch.no_grad():
                    out = model(ids)
                lg = out.logits
                c, u = lg[0:1, ts:], lg[1:2, ts:]
                comb = c + kw["text_cfg"] * (u - c)
                masked = ids[0, ts:] == synth.MASK
                if bool(masked.any()):
                    p = torch.softmax(comb[0][masked].to(torch.float64), -1).max(-1).values
                    top_conf.append(float((1.0 - p).min()))
                else:
                    top_conf.append(1.0)
                return out

        class Tok:
            bos_token_id = job["bos"]

            def __len__(self):
                return job["text_vocab"]

        cfgobj = SimpleNamespace(model=SimpleNamespace(mmada=SimpleNamespace(num_vq_tokens=job["N"], codebook_size=job["codebook"])),
                                 dataset=SimpleNamespace(preprocessing=SimpleNamespace(max_seq_length=job["T"])))
        rng = SeededRng(M_PEAKED_SEED)
        real_uniform = torch.Tensor.uniform_

        def fake_multinomial(inp_, num, replacement=False, *, generator=None):
            return rng.multinomial(inp_)[:, None]

        def fake_uniform(self, a=0, b=1, *, generator=None):
            return real_uniform(self, a, b, generator=rng._g())

        with mock.patch.object(torch, "multinomial", fake_multinomial), \
                mock.patch.object(torch.Tensor, "uniform_", fake_uniform), \
                contextlib.redirect_stdout(io.StringIO()):
            img, text = mm.MMadaModelLM.interleave_generate(
                Self(), job["input_ids"], job["uncond_input_ids"], reserved_token_mapping={"<|soi|>": job["soi"], "<|eoi|>": job["eoi"]},
                config=cfgobj, uni_prompting=SimpleNamespace(text_tokenizer=Tok()), generator=None, **kw)
        return dict(calls=torch.stack(calls, 0).numpy().astype(np.int32), img=img.numpy().astype(np.int64),
                    text=text.numpy().astype(np.int64), one_minus_top_text_conf=np.array(top_conf, np.float64))
    finally:
        if saved is None:
            sys.modules.pop("models", None)
        else:
            sys.modules["models"] = saved
        for k in [k for k in sys.modules if k.startswith("models.")]:
            sys.modules.pop(k, None)


M_PEAKED_SEED = 53


def gen_m_peaked():
    d = compute_m_peaked()
    np.savez_compressed(os.path.join(OUT, f"m_peaked_traj.{synth.host_isa()}.npz"), **d)
    print(f"m_peaked_traj.{synth.host_isa()}: {d['calls'].shape[0]} batch-2 forwards of L = {d['calls'].shape[2]}; "
          f"{len(set(d['img'].reshape(-1).tolist()))} distinct image ids, {len(set(d['text'].reshape(-1).tolist()))} distinct text ids; "
          f"smallest 1 - p of a masked text row over all calls {d['one_minus_top_text_conf'].min():.3e}")


# ---- M variant: MMadaModelLM.interleave_generate driven by stub logits and per-call seeded RNG draws ------------------
M_REF = "/root/reference/MMaDA-Parallel-M"
M_CASES = {
    "m_both": dict(text_cfg=1.5, image_cfg=3.5, text_steps=8, image_steps=4, image_temperature=1.0, text_temperature=0.0),
    "m_img": dict(text_cfg=0.0, image_cfg=2.0, text_steps=7, image_steps=7, image_temperature=0.5, text_temperature=0.0),
    "m_noisy": dict(text_cfg=0.7, image_cfg=3.5, text_steps=6, image_steps=3, image_temperature=1.0, text_temperature=0.8),
}
M_SHAPE = dict(prompt=6, N=16, T=16, text_vocab=2048, CB=512, soi=2040, eoi=2041, bos=2042, mask_id=126336)


def gen_m_traj():
    import importlib
    import types
    from unittest import mock

    from oracle.interleave_oracle import SeededRng

    pkg = types.ModuleType("models")
    pkg.__path__ = [M_REF + "/models"]
    saved = {k: sys.modules.get(k) for k in ("models",)}
    sys.modules["models"] = pkg  # the package __init__ imports files that do not exist: bypass it
    mm = importlib.import_module("models.modeling_mmada")
    sh = M_SHAPE
    V = sh["text_vocab"] + sh["CB"]
    out = {}
    for ci, (name, kw) in enumerate(M_CASES.items()):
        seed = 31 + ci
        g = torch.Generator().manual_seed(seed)
        inp = torch.randint(0, 2000, (sh["prompt"],), generator=g)
        unc = torch.randint(0, 2000, (sh["prompt"],), generator=g)
        calls = []
        n = [0]

        class FakeSelf:
            config = SimpleNamespace(mask_token_id=sh["mask_id"])

            def __call__(self, ids):
                n[0] += 1
                calls.append(ids.clone())
                return SimpleNamespace(logits=stub_logits(seed, n[0], ids.shape[0], ids.shape[1], V))

        class Tok:
            bos_token_id = sh["bos"]

            def __len__(self):
                return sh["text_vocab"]

        cfgobj = SimpleNamespace(model=SimpleNamespace(mmada=SimpleNamespace(num_vq_tokens=sh["N"], codebook_size=sh["CB"])),
                                 dataset=SimpleNamespace(preprocessing=SimpleNamespace(max_seq_length=sh["T"])))
        rng = SeededRng(seed)
        real_uniform = torch.Tensor.uniform_

        def fake_multinomial(inp_, num, replacement=False, *, generator=None):
            return rng.multinomial(inp_)[:, None]

        def fake_uniform(self, a=0, b=1, *, generator=None):
            return real_uniform(self, a, b, generator=rng._g())

        def fake_rand_like(t, dtype=None, **_):
            return rng.rand_f64(t.shape)

        with mock.patch.object(torch, "multinomial", fake_multinomial), \
                mock.patch.object(torch.Tensor, "uniform_", fake_uniform), \
                mock.patch.object(torch, "rand_like", fake_rand_like), \
                contextlib.redirect_stdout(io.StringIO()):
            img, text = mm.MMadaModelLM.interleave_generate(
                FakeSelf(), inp, unc, reserved_token_mapping={"<|soi|>": sh["soi"], "<|eoi|>": sh["eoi"]}, config=cfgobj,
                uni_prompting=SimpleNamespace(text_tokenizer=Tok()), generator=None, **kw)
        out[name + "_calls"] = torch.stack(calls, 0).numpy()   # [steps, 2, L]
        out[name + "_img"] = img.numpy()
        out[name + "_text"] = text.numpy()
        out[name + "_seed"] = np.array(seed)
        out[name + "_inp"] = inp.numpy()
        out[name + "_unc"] = unc.numpy()
        print(f"m_traj[{name}]: {len(calls)} forwards, L={calls[0].shape[1]}")
    np.savez_compressed(os.path.join(OUT, "m_traj.npz"), **out)
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def gen_stepwise_traj():
    """Reference Gradio sampler generate_ti2ti_stepwise (app.py:143-398) on stub logits; gradio / diffusers are
    absent here and only used for UI / VQ decode, so they are stubbed (the decode call sits in a try/except)."""
    import importlib
    import types

    from unittest import mock

    for name in ("gradio", "diffusers", "diffusers.image_processor"):
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock()  # app.py builds its gr.Blocks UI at import time
    app = importlib.import_module("app")

    class Tok:
        def decode(self, ids, **_):
            return "x"

    job = tiny_job()
    V = STUB_TEXT_VOCAB + STUB_CB
    out = {}
    for ci, (name, kw) in enumerate({"sw_img4": dict(text_steps=10, cfg_scale=0.0, cfg_img=4.0),
                                     "sw_both": dict(text_steps=14, cfg_scale=2.5, cfg_img=4.0)}.items()):
        seed = 51 + ci

        def fn(ids, call_idx, seed=seed):
            return SimpleNamespace(logits=stub_logits(seed, call_idx, ids.shape[0], ids.shape[1], V))

        rec = Recorder(fn)
        torch.manual_seed(4321)
        steps = []
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            try:
                for s_ in app.generate_ti2ti_stepwise(
                        rec, job["input_ids"], job["text_start"], job["text_end"], job["image_start"], job["seq_len"],
                        job["newline_every"], temperature=0.0, text_temperature=0.0, uncon_text=job["uncon_text"],
                        uncon_image=job["uncon_image"], tokenizer=Tok(), text_vocab_size=STUB_TEXT_VOCAB,
                        codebook_size=STUB_CB, vqvae=None, image_height=64, image_width=64, **kw):
                    steps.append(s_[0])
            except Exception:
                pass  # the final VQ-VAE decode (mocked diffusers) fails after the sampling loop has finished
        out[name + "_calls"] = torch.cat(rec.calls, 0).numpy()
        out[name + "_yields"] = np.array(steps)
        out[name + "_seed"] = np.array(seed)
        print(f"stepwise_traj[{name}]: {len(rec.calls)} model calls, yields at {steps}")
    np.savez_compressed(os.path.join(OUT, "stepwise_traj.npz"), **out)


def gen_vq_decode():
    """Runs the reference's own MAGVITv2.decode_code on its own LFQuantizer / VQGANDecoder modules.  Only
    models/modeling_utils.py (the HF ModelMixin clone: checkpoint IO, no arithmetic; needs diffusers/omegaconf) is
    replaced by plain nn.Module mixins, and the unused encoder is not instantiated."""
    import importlib
    import types
    from unittest import mock

    pkg = types.ModuleType("models")
    pkg.__path__ = [M_REF + "/models"]
    saved = {k: sys.modules.get(k) for k in ("models", "models.modeling_utils")}
    sys.modules["models"] = pkg
    for name in ("omegaconf", "jaxtyping", "typeguard"):
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock()
    mu = types.ModuleType("models.modeling_utils")
    mu.ConfigMixin = type("ConfigMixin", (), {})
    mu.ModelMixin = type("ModelMixin", (torch.nn.Module,), {})
    mu.register_to_config = lambda f: f
    sys.modules["models.modeling_utils"] = mu
    mv = importlib.import_module("models.modeling_magvitv2")
    out = {}
    for name, cfg, B, hz in (("tiny", synth.VQ_CFG_TINY, 2, 8), ("full", synth.VQ_CFG_M, 1, 32)):
        seed = 7 if name == "tiny" else 8
        sd = synth.synthetic_vq_state_dict(cfg, seed)
        with contextlib.redirect_stdout(io.StringIO()):
            dec = mv.VQGANDecoder(ch=cfg["ch"], ch_mult=cfg["ch_mult"], num_res_blocks=cfg["num_res_blocks"],
                                  z_channels=cfg["z_channels"], out_ch=cfg["out_ch"])
            quant = mv.LFQuantizer(codebook_dim=cfg["z_channels"])
        dec.load_state_dict(sd, strict=True)
        dec.eval()
        if name == "full":  # the checkpoint contract: key names and shapes of the reference module with default arguments
            out["full_keys"] = np.array([f"{k}:{tuple(v.shape)}" for k, v in dec.state_dict().items()])
        g = torch.Generator().manual_seed(100 + seed)
        idx = torch.randint(0, 2 ** cfg["z_channels"], (B, hz * hz), generator=g)
        with torch.no_grad():
            img = mv.MAGVITv2.decode_code(SimpleNamespace(quantize=quant, decoder=dec), idx)
        out[name + "_idx"] = idx.numpy()
        out[name + "_seed"] = np.array(seed)
        out[name + "_out"] = (img if name == "tiny" else img[:, :, ::4, ::4]).contiguous().numpy()
        out[name + "_stats"] = np.array([img.mean().item(), img.std().item(), img.abs().max().item()])
        print(f"vq_decode[{name}]: {tuple(idx.shape)} -> {tuple(img.shape)}, std {img.std().item():.4f}")
    np.savez_compressed(os.path.join(OUT, "vq_decode.npz"), **out)
    # encoder direction: the reference's own MAGVITv2.get_code on its VQGANEncoder + LFQuantizer.forward / get_indices
    out = {}
    for name, cfg, B, res in (("tiny", synth.VQ_ENC_CFG_TINY, 2, 16), ("full", synth.VQ_ENC_CFG_M, 1, 512)):
        seed = 17 if name == "tiny" else 18
        sd = synth.synthetic_vq_state_dict(cfg, seed)
        with contextlib.redirect_stdout(io.StringIO()):
            enc = mv.VQGANEncoder(ch=cfg["ch"], ch_mult=cfg["ch_mult"], num_res_blocks=cfg["num_res_blocks"],
                                  z_channels=cfg["z_channels"], in_ch=cfg["in_ch"])
            quant = mv.LFQuantizer(codebook_dim=cfg["z_channels"])
        enc.load_state_dict(sd, strict=True)
        enc.eval()
        if name == "full":
            out["full_keys"] = np.array([f"{k}:{tuple(v.shape)}" for k, v in enc.state_dict().items()])
        img = synth.synthetic_image(B, res, res, seed=200 + seed)
        with torch.no_grad():
            idx = mv.MAGVITv2.get_code(SimpleNamespace(encoder=enc, quantize=quant), img)
            z = enc(img)
        out[name + "_seed"] = np.array(seed)
        out[name + "_idx"] = idx.numpy()
        out[name + "_z"] = z.numpy()
        print(f"vq_encode[{name}]: {tuple(img.shape)} -> {tuple(idx.shape)}, z std {z.std().item():.3f}, "
              f"|z| < 1e-3: {(z.abs() < 1e-3).sum().item()}")
    np.savez_compressed(os.path.join(OUT, "vq_encode.npz"), **out)
    for k, v in saved.items():
        sys.modules.pop(k, None)
        if v is not None:
            sys.modules[k] = v
    for k in [k for k in sys.modules if k.startswith("models.")]:
        sys.modules.pop(k, None)


def gen_mmu_traj():
    """Reference MMadaModelLM.mmu_generate (models/modeling_mmada.py:618-692) on stub logits (B = 2); temperature > 0
    draws (torch.rand_like, float64) come from per-call seeded generators (SeededRng), replayed by the tests."""
    import importlib
    import types
    from unittest import mock

    from oracle.interleave_oracle import SeededRng
    from tests.helpers import MMU_CASES, MMU_SHAPE

    pkg = types.ModuleType("models")
    pkg.__path__ = [M_REF + "/models"]
    saved = sys.modules.get("models")
    sys.modules["models"] = pkg
    mm = importlib.import_module("models.modeling_mmada")
    sh = MMU_SHAPE
    out = {}
    for ci, (name, kw) in enumerate(MMU_CASES.items()):
        seed = 91 + ci
        g = torch.Generator().manual_seed(seed)
        idx = torch.randint(0, 2000, (sh["B"], sh["P"]), generator=g)
        calls, n = [], [0]

        class FakeSelf:
            device = torch.device("cpu")

            def __call__(self, ids, attention_bias=None):
                n[0] += 1
                calls.append(ids.clone())
                return SimpleNamespace(logits=stub_logits(seed, n[0], ids.shape[0], ids.shape[1], sh["V"]))

        rng = SeededRng(seed)

        def fake_rand_like(t, dtype=None, **_):
            return rng.rand_f64(t.shape)

        with mock.patch.object(torch, "rand_like", fake_rand_like):
            x = mm.MMadaModelLM.mmu_generate(FakeSelf(), idx=idx, mask_id=sh["mask_id"], **kw)
        out[name + "_idx"] = idx.numpy()
        out[name + "_calls"] = torch.stack(calls, 0).numpy()
        out[name + "_x"] = x.numpy()
        out[name + "_seed"] = np.array(seed)
        print(f"mmu_traj[{name}]: {len(calls)} forwards of shape {tuple(calls[0].shape)}")
    np.savez_compressed(os.path.join(OUT, "mmu_traj.npz"), **out)
    sys.modules.pop("models", None)
    if saved is not None:
        sys.modules["models"] = saved


def gen_m_t2i_traj():
    """Reference MMadaModelLM.t2i_generate (models/modeling_mmada.py:264-359) on stub logits; torch.multinomial and
    Tensor.uniform_ are served by per-call seeded generators (SeededRng) that the tests replay."""
    import importlib
    import types
    from unittest import mock

    from oracle.interleave_oracle import SeededRng
    from tests.helpers import M_T2I_CASES, M_T2I_SHAPE, m_t2i_job

    pkg = types.ModuleType("models")
    pkg.__path__ = [M_REF + "/models"]
    saved = sys.modules.get("models")
    sys.modules["models"] = pkg
    mm = importlib.import_module("models.modeling_mmada")
    sh = M_T2I_SHAPE
    V = sh["text_vocab"] + sh["CB"]
    out = {}
    for ci, (name, kw) in enumerate(M_T2I_CASES.items()):
        seed = 111 + ci
        inp, unc = m_t2i_job(seed, kw["B"], kw["known"])
        calls, n = [], [0]

        class FakeSelf:
            def __call__(self, ids, attention_bias=None):
                n[0] += 1
                calls.append(ids.clone())
                return SimpleNamespace(logits=stub_logits(seed, n[0], ids.shape[0], ids.shape[1], V))

        class Tok:
            def __len__(self):
                return sh["text_vocab"]

        rng = SeededRng(seed)
        real_uniform = torch.Tensor.uniform_

        def fake_multinomial(inp_, num, replacement=False, *, generator=None):
            return rng.multinomial(inp_)[:, None]

        def fake_uniform(self, a=0, b=1, *, generator=None):
            return real_uniform(self, a, b, generator=rng._g())

        ones = torch.ones_like(inp)
        work = inp.clone()
        with mock.patch.object(torch, "multinomial", fake_multinomial), \
                mock.patch.object(torch.Tensor, "uniform_", fake_uniform):
            ids = mm.MMadaModelLM.t2i_generate(
                FakeSelf(), input_ids=work, uncond_input_ids=unc.clone() if kw["uncond"] else None, attention_mask=ones,
                uncond_attention_mask=ones, temperature=kw["temperature"], timesteps=kw["timesteps"],
                guidance_scale=kw["guidance_scale"], seq_len=sh["N"], mask_token_id=sh["mask_id"],
                resolution=sh["resolution"], codebook_size=sh["CB"], uni_prompting=SimpleNamespace(text_tokenizer=Tok()))
        out[name + "_calls"] = torch.stack(calls, 0).numpy()
        out[name + "_ids"] = ids.numpy()
        out[name + "_final_input"] = work.numpy()
        out[name + "_seed"] = np.array(seed)
        print(f"m_t2i_traj[{name}]: {len(calls)} forwards of shape {tuple(calls[0].shape)}")
    np.savez_compressed(os.path.join(OUT, "m_t2i_traj.npz"), **out)
    sys.modules.pop("models", None)
    if saved is not None:
        sys.modules["models"] = saved


def gen_t2i_traj():
    """Reference generate_image (generators/image_generation_generator.py) driven by a stub model that returns seeded
    random bf16 logits: ids of every model call + returned vq ids.  temperature > 0 draws from a seeded CPU generator."""
    from generators.image_generation_generator import generate_image
    from tests.helpers import T2I_CASES, t2i_job

    class Stub(torch.nn.Module):
        def __init__(self, seed, V):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.seed, self.V, self.n, self.calls = seed, V, 0, []
            self.module = self  # the reference calls model.module.caching() on anything that is not its own class

        def caching(self, enable=True):
            pass

        def forward(self, ids, infer=True, use_cache=False):
            self.n += 1
            self.calls.append(ids.clone())
            return SimpleNamespace(logits=stub_logits(self.seed, self.n, 1, ids.shape[1], self.V))

    job = t2i_job()
    V = STUB_TEXT_VOCAB + STUB_CB
    out = {}
    for ci, (name, kw) in enumerate(T2I_CASES.items()):
        seed = 71 + ci
        stub = Stub(seed, V)
        gen = torch.Generator().manual_seed(900 + ci) if kw["temperature"] > 0 else None
        with contextlib.redirect_stdout(io.StringIO()):
            vq = generate_image(stub, job["prompt"], seq_len=job["seq_len"], newline_every=job["newline_every"],
                                uncon_ids=job["uncon_ids"], code_start=job["code_start"], codebook_size=STUB_CB,
                                text_vocab_size=STUB_TEXT_VOCAB, generator=gen, debug=False, **kw)
        L = job["prompt"].shape[1]
        out[name + "_calls_len"] = np.array([c.shape[1] for c in stub.calls])
        out[name + "_calls"] = torch.cat([torch.nn.functional.pad(c, (0, L + 8 - c.shape[1]), value=-1) for c in stub.calls], 0).numpy()
        out[name + "_vq"] = vq.numpy()
        out[name + "_seed"] = np.array(seed)
        out[name + "_gen_seed"] = np.array(900 + ci)
        print(f"t2i_traj[{name}]: {len(stub.calls)} model calls, vq {tuple(vq.shape)}")
    np.savez_compressed(os.path.join(OUT, "t2i_traj.npz"), **out)


def gen_tables():
    b = torch.arange(0, 0x7f80, dtype=torch.int32).to(torch.int16)
    np.save(os.path.join(OUT, "logconf_table.npy"), torch.log(b.view(torch.bfloat16) + 1e-10).view(torch.int16).numpy())


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not mounted at " + REF)
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    if only:  # e.g. `python oracle/gen_golden.py gen_vq_decode` regenerates one fixture
        for fn in only:
            globals()[fn]()
        sys.exit(0)
    gen_tables()
    gen_t2i_traj()
    gen_mmu_traj()
    gen_m_t2i_traj()
    gen_vq_decode()
    gen_stepwise_traj()
    gen_m_traj()
    gen_sampler_traj()
    gen_paint_traj()
    gen_random_traj()
    gen_edge_traj()
    gen_image_utils()
    gen_sampler_noisy()
    gen_forward()
    gen_e2e()
    gen_peaked()
    gen_m_peaked()
    gen_dllm_cache()
