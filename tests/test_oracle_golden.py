"""Pin the CPU oracle against fixtures produced by RUNNING the reference (oracle/gen_golden.py -> tests/golden/)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import (GOLDEN, SAMPLER_CASES, STUB_CB, STUB_TEXT_VOCAB, bits, from_bits, golden_float, stub_logits, tiny_job,
                     tiny_sd)
from mmada_parallel_amd import synth
from oracle import generate_oracle, llada_oracle
from oracle import sampler_oracle as so


def test_log_conf_table_matches_reference_exhaustively():
    # torch.log(p + 1e-10) in bf16 for every non-negative bf16 p (parallel_generator.py:36)
    ref = torch.from_numpy(np.load(os.path.join(GOLDEN, "logconf_table.npy")))
    assert torch.equal(so.log_conf_table(), ref)


@pytest.mark.parametrize("name", list(SAMPLER_CASES))
def test_sampler_trajectory_matches_reference(name):
    """The whole sampler (text select, CFG, softmax/argmax, re-mask, write-back, schedules) driven by stub logits:
    the ids handed to EVERY model call must equal what the reference's generate_ti2ti produced."""
    z = np.load(os.path.join(GOLDEN, "sampler_traj.npz"))
    calls_ref = torch.from_numpy(z[name + "_calls"])
    seed = int(z[name + "_seed"])
    job, kw = tiny_job(), SAMPLER_CASES[name]
    V = STUB_TEXT_VOCAB + STUB_CB
    n = [0]

    def model_fn(ids):
        n[0] += 1
        return stub_logits(seed, n[0], ids.shape[0], ids.shape[1], V)

    trace = []
    final = generate_oracle.generate(model_fn, job["input_ids"], job["text_start"], job["text_end"], job["image_start"],
                                     job["seq_len"], job["newline_every"], uncon_text=job["uncon_text"],
                                     uncon_image=job["uncon_image"], text_vocab_size=STUB_TEXT_VOCAB,
                                     codebook_size=STUB_CB, trace=trace, **kw)
    got = torch.cat(trace, 0)
    assert got.shape == calls_ref.shape
    assert torch.equal(got, calls_ref), f"first differing call: {(got != calls_ref).any(1).nonzero()[0].item()}"
    # final outputs: all non-MASK image tokens and the text must agree (the single MASK left is a random fill, A.1)
    pos = [i for i in range(job["image_start"], job["image_start"] + job["seq_len"] + job["seq_len"] // job["newline_every"])
           if int(job["input_ids"][0, i]) != synth.NEW_LINE]
    vq_ref = z[name + "_vq"]
    still = 0
    for j, p in enumerate(pos):
        tok = int(final[0, p])
        if tok == synth.MASK:
            still += 1
        else:
            assert tok - STUB_TEXT_VOCAB == vq_ref[j]
    assert still == 1  # exactly one token is left for torch.randint
    text = [t for t in final[0, job["text_start"]:job["text_end"]].tolist() if t != synth.MASK]
    assert text == z[name + "_text"].tolist()


@pytest.mark.parametrize("name", ["inpaint_img4", "outpaint_both"])
def test_painting_mode_trajectory_matches_reference(name):
    """Painting mode (inference.py:141-146): the output image span starts partly known, so the image branch's unknown count
    starts below N and the known cells must survive every re-mask."""
    from helpers import PAINT_CASES, paint_job

    z = np.load(os.path.join(GOLDEN, "paint_traj.npz"))
    kind, kw = PAINT_CASES[name]
    job = paint_job(kind)
    seed = int(z[name + "_seed"])
    V = STUB_TEXT_VOCAB + STUB_CB
    n = [0]

    def model_fn(ids):
        n[0] += 1
        return stub_logits(seed, n[0], ids.shape[0], ids.shape[1], V)

    trace = []
    final = generate_oracle.generate(model_fn, job["input_ids"], job["text_start"], job["text_end"], job["image_start"],
                                     job["seq_len"], job["newline_every"], uncon_text=job["uncon_text"],
                                     uncon_image=job["uncon_image"], text_vocab_size=STUB_TEXT_VOCAB,
                                     codebook_size=STUB_CB, trace=trace, **kw)
    assert torch.equal(torch.cat(trace, 0), torch.from_numpy(z[name + "_calls"]))
    pos = [i for i in range(job["image_start"], job["image_start"] + job["seq_len"] + job["seq_len"] // job["newline_every"])
           if int(job["input_ids"][0, i]) != synth.NEW_LINE]
    known0 = [int(job["input_ids"][0, p]) for p in pos]
    for j, p in enumerate(pos):
        tok = int(final[0, p])
        if known0[j] != synth.MASK:
            assert tok == known0[j]                       # a known cell is never re-masked or overwritten
        if tok != synth.MASK:
            assert tok - STUB_TEXT_VOCAB == z[name + "_vq"][j]


@pytest.mark.parametrize("name", ["rand_img4", "rand_both"])
def test_random_remasking_trajectory_matches_reference(name):
    """remasking='random' (inference.py --remasking random; generators/parallel_generator.py:194-198): text positions ranked
    by uniform draws from the global CPU generator, which the re-mask jitter's randn advances too."""
    from helpers import RANDOM_CASES, RANDOM_SEED

    z = np.load(os.path.join(GOLDEN, "random_traj.npz"))
    job, kw = tiny_job(), RANDOM_CASES[name]
    seed = int(z[name + "_seed"])
    V = STUB_TEXT_VOCAB + STUB_CB
    n = [0]

    def model_fn(ids):
        n[0] += 1
        return stub_logits(seed, n[0], ids.shape[0], ids.shape[1], V)

    trace = []
    torch.manual_seed(RANDOM_SEED)
    final = generate_oracle.generate(model_fn, job["input_ids"], job["text_start"], job["text_end"], job["image_start"],
                                     job["seq_len"], job["newline_every"], uncon_text=job["uncon_text"],
                                     uncon_image=job["uncon_image"], text_vocab_size=STUB_TEXT_VOCAB,
                                     codebook_size=STUB_CB, trace=trace, remasking="random", **kw)
    assert torch.equal(torch.cat(trace, 0), torch.from_numpy(z[name + "_calls"]))
    text = [t for t in final[0, job["text_start"]:job["text_end"]].tolist() if t != synth.MASK]
    assert text == z[name + "_text"].tolist()


@pytest.mark.parametrize("name", ["cfg_without_uncond", "only_uncon_image", "text_done", "more_timesteps_than_steps", "single_step"])
def test_edge_case_trajectory_matches_reference(name):
    from helpers import edge_job

    z = np.load(os.path.join(GOLDEN, "edge_traj.npz"))
    job, kw = edge_job(name)
    seed = int(z[name + "_seed"])
    V = STUB_TEXT_VOCAB + STUB_CB
    n = [0]

    def model_fn(ids):
        n[0] += 1
        return stub_logits(seed, n[0], ids.shape[0], ids.shape[1], V)

    trace = []
    final = generate_oracle.generate(model_fn, job["input_ids"], job["text_start"], job["text_end"], job["image_start"],
                                     job["seq_len"], job["newline_every"], uncon_text=job["uncon_text"],
                                     uncon_image=job["uncon_image"], text_vocab_size=STUB_TEXT_VOCAB,
                                     codebook_size=STUB_CB, trace=trace, **kw)
    got = torch.cat(trace, 0)
    ref = torch.from_numpy(z[name + "_calls"])
    assert got.shape == ref.shape and torch.equal(got, ref)
    text = [t for t in final[0, job["text_start"]:job["text_end"]].tolist() if t != synth.MASK]
    assert text == z[name + "_text"].tolist()


def _live_reference():
    """oracle/gen_golden.py's compute_* functions run the UNMODIFIED reference on this host; None where the reference tree
    is not mounted (GPU box)."""
    if not os.path.isdir("/root/reference/MMaDA-Parallel-A"):
        return None
    from oracle import gen_golden

    return gen_golden


def _check_forward(z, exact):
    ids = torch.from_numpy(z["ids"])
    sd, cfg = tiny_sd(), synth.CFG_TINY
    taps = []
    x = llada_oracle.forward_hidden(sd, cfg, ids, taps)
    hidden_ref = from_bits(z["hidden"])
    pos = z["pos"].tolist()
    img = llada_oracle.head(sd, cfg, x[:, pos], synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK)
    job = tiny_job()
    full = llada_oracle.head(sd, cfg, x)
    if exact:
        for i, t in enumerate(taps):
            assert torch.equal(bits(t[0]), bits(hidden_ref[i])), f"block {i}"
        # the reference multiplies ALL rows by the whole head matrix (modeling_llada.py:1399-1404); the row / column subset
        # form `head(x[:, pos], lo, hi)` is the same dot products in a GEMM of another shape, whose blocking (and last bit)
        # depends on the CPU class: bit-equal on AMX hosts, 1 ulp apart on some entries elsewhere
        assert torch.equal(bits(full[0, pos, synth.TEXT_VOCAB:synth.TEXT_VOCAB + synth.CODEBOOK]),
                           torch.from_numpy(z["img_logits"]))
        ref_img = from_bits(z["img_logits"]).float()
        assert (img[0].float() - ref_img).abs().max() <= 2.0 ** -7 * ref_img.abs().max()
        assert torch.equal(full[0].argmax(-1).int(), torch.from_numpy(z["argmax"]))
        assert torch.equal(bits(full[0, job["text_start"]:job["text_end"], :4096]), torch.from_numpy(z["text_logits_head"]))
        return
    # a recording from another CPU class: same arithmetic, another fp32 accumulation order inside the bf16 GEMMs
    for i, t in enumerate(taps):
        r = hidden_ref[i].float()
        err = (t[0].float() - r).abs()
        assert err.max() <= 2.0 ** -6 * r.abs().max() and err.mean() <= 2.0 ** -9 * r.abs().max(), f"block {i}"
    ref_img = from_bits(z["img_logits"]).float()
    assert (img[0].float() - ref_img).abs().mean() <= 2.0 ** -7 * ref_img.abs().max()
    assert (full[0].argmax(-1).int() == torch.from_numpy(z["argmax"])).float().mean() >= 0.9


def test_forward_matches_reference():
    """(i) /root/reference mounted: the unmodified reference is run HERE and the oracle must equal it bit for bit — the pin,
    independent of which CPU this is; (ii) the committed recording of this host's ISA class: bit for bit; (iii) recordings
    of other classes: bf16 re-association tolerance."""
    live = _live_reference()
    if live is not None:
        _check_forward(live.compute_forward(), exact=True)
    z, same = golden_float("forward_tiny")
    _check_forward(z, exact=same)
    if same:  # the other classes' recordings, by tolerance
        for other in ("amx_bf16", "avx512"):
            p = os.path.join(GOLDEN, f"forward_tiny.{other}.npz")
            if other != synth.host_isa() and os.path.exists(p):
                _check_forward(np.load(p), exact=False)


def _oracle_e2e_calls():
    sd, cfg, job = tiny_sd(), synth.CFG_TINY, tiny_job()
    trace = []
    generate_oracle.generate(lambda ids: llada_oracle.forward_logits(sd, cfg, ids), job["input_ids"], job["text_start"],
                             job["text_end"], job["image_start"], job["seq_len"], job["newline_every"], text_steps=8,
                             timesteps=4, cfg_scale=0.0, cfg_img=4.0, uncon_text=job["uncon_text"],
                             uncon_image=job["uncon_image"], trace=trace)
    return torch.cat(trace, 0)


def test_e2e_tiny_matches_reference():
    got = _oracle_e2e_calls()
    live = _live_reference()
    if live is not None:
        assert torch.equal(got, torch.from_numpy(live.compute_e2e()["calls"]))
    z, same = golden_float("e2e_tiny")
    ref = torch.from_numpy(z["calls"])
    assert got.shape == ref.shape
    if same:
        assert torch.equal(got, ref)
    else:  # free-running on random weights across CPU classes: near-tie flips compound; the first call is the
        # shared input, and most ids still agree
        assert torch.equal(got[0], ref[0]) and (got == ref).float().mean() >= 0.6


def test_peaked_trajectory_matches_reference():
    """The PEAKED synthetic checkpoint (synth.synthetic_state_dict_peaked: one planted copy circuit on top of the random
    weights, decision margins far above bf16 noise), free-running at BASELINE configs[0] geometry and schedule (L = 1654, 32 text
    + 16 image steps, 64 model calls): the oracle must reproduce the ids the UNMODIFIED reference passed to every model call
    (tests/golden/peaked_traj.*.npz) — bit for bit on a CPU of the recording's class, and >= 99.5 % of all ids on any other
    (that is the point of the checkpoint: no near-ties for a re-ordered GEMM to flip).  The re-mask cut of an image step is
    taken with torch.sort as the reference takes it (tie_order="torch"): a peaked model's bf16 confidences tie massively (many
    are exactly 1.0) and torch.sort's order of ties is a property of the PyTorch build, not of the algorithm."""
    cfg = synth.CFG_PEAKED
    job = synth.synthetic_job(256, 256, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    sd = synth.synthetic_state_dict_peaked(cfg, synth.peaked_delta(job))
    trace = []
    torch.manual_seed(1234)
    generate_oracle.generate(lambda ids: llada_oracle.forward_logits(sd, cfg, ids), job["input_ids"], job["text_start"],
                             job["text_end"], job["image_start"], job["seq_len"], job["newline_every"], text_steps=32,
                             timesteps=16, cfg_scale=0.0, cfg_img=4.0, uncon_text=job["uncon_text"],
                             uncon_image=job["uncon_image"], trace=trace, tie_order="torch")
    got = torch.cat(trace, 0)
    z, same = golden_float("peaked_traj")
    ref = torch.from_numpy(z["calls"].astype(np.int64))
    assert got.shape == ref.shape == (64, 1654)
    assert len(set(z["vq"].tolist())) > 100 and len(set(z["text"].tolist())) > 100   # position-dependent predictions
    if same:
        assert torch.equal(got, ref)
    else:
        assert (got == ref).float().mean() >= 0.995


def test_m_peaked_trajectory_matches_reference():
    """MMaDA-Parallel-M end to end: oracle/interleave_oracle.py driven by oracle/llada_oracle.py on the peaked checkpoint at
    BASELINE configs[3] geometry (L = 2349, batch-2 forwards, text_cfg 2.5, image_cfg 4, 24 steps of which 8 image steps) must
    reproduce the ids the UNMODIFIED reference — MMadaModelLM.interleave_generate on the M tree's own LLaDAModelLM,
    oracle/gen_golden.py gen_m_peaked — passed to every forward, and its outputs, with the reference's multinomial / uniform
    draws replayed: bit for bit on a CPU of the recording's class, >= 99.5 % on any other.  The oracle's head runs on the rows
    the sampler reads (image rows x codebook slab, text rows x vocabulary)."""
    from oracle import interleave_oracle as io_
    from oracle.interleave_oracle import SeededRng

    z, same = golden_float("m_peaked_traj")
    ref = torch.from_numpy(z["calls"].astype(np.int64))
    cfg, job, kw = synth.CFG_PEAKED, synth.m_peaked_job(), dict(synth.M_PEAKED_KW)
    assert float(z["one_minus_top_text_conf"].min()) > 0.0, "an fp64 text confidence of exactly 1.0: torch.topk would order ties"
    sd = synth.synthetic_state_dict_peaked(cfg, job["delta"], beta=synth.M_PEAKED_BETA)
    i0, ts, N, L = job["img_start"], job["text_start"], job["N"], job["L"]
    lo, V = job["text_vocab"], cfg["vocab_size"]
    buf = torch.zeros((2, L, V), dtype=torch.bfloat16)

    def model_fn(both):
        h = llada_oracle.forward_hidden(sd, cfg, both)
        buf[:, i0:i0 + N, lo:lo + job["codebook"]] = llada_oracle.head(sd, cfg, h[:, i0:i0 + N], lo, lo + job["codebook"])
        buf[:, ts:] = llada_oracle.head(sd, cfg, h[:, ts:])
        return buf

    trace = []
    img, text = io_.generate(model_fn, job["input_ids"], job["uncond_input_ids"], kw["text_cfg"], kw["image_cfg"], kw["text_steps"],
                             kw["image_steps"], job["soi"], job["eoi"], job["bos"], synth.MASK, lo, N, job["codebook"], job["T"],
                             kw["image_temperature"], SeededRng(53), text_temperature=kw["text_temperature"], trace=trace)
    got = torch.stack(trace, 0)
    assert got.shape == ref.shape == (kw["text_steps"], 2, 2349)
    assert len(set(z["img"].reshape(-1).tolist())) > 300
    if same:
        assert torch.equal(got, ref), f"first differing call: {int((got != ref).any(2).any(1).nonzero()[0])}"
        assert torch.equal(img, torch.from_numpy(z["img"])) and torch.equal(text, torch.from_numpy(z["text"]))
    else:
        assert (got == ref).float().mean() >= 0.995


def _check_dllm_cache(z, exact):
    """oracle/llada_oracle.py forward_logits_cached replayed over the fixture's script (oracle/gen_golden.py
    dllm_cache_script) must return the reference's logit cache after every call."""
    sd, cfg = tiny_sd(), synth.CFG_TINY
    for tag, enable, script in (("on", True, synth.dllm_cache_script()), ("off", False, synth.dllm_cache_script()[:2])):
        cache = llada_oracle.DllmCache(cfg["n_layers"])
        cache.caching(enable)
        for n, (cat, ids, m) in enumerate(script):
            lg = llada_oracle.forward_logits_cached(sd, cfg, ids, cache, to_compute_mask=m, cat=cat)
            img = lg[0, :, synth.TEXT_VOCAB:synth.TEXT_VOCAB + 256]
            txt = lg[0, :, :256]
            if exact:
                assert torch.equal(bits(img), torch.from_numpy(z[f"{tag}{n}_img"])), f"{tag}{n} img"
                assert torch.equal(bits(txt), torch.from_numpy(z[f"{tag}{n}_txt"])), f"{tag}{n} txt"
                assert torch.equal(lg[0].argmax(-1).int(), torch.from_numpy(z[f"{tag}{n}_argmax"])), f"{tag}{n} argmax"
            else:
                for got, key in ((img, "img"), (txt, "txt")):
                    ref = from_bits(z[f"{tag}{n}_{key}"]).float()
                    assert (got.float() - ref).abs().mean() <= 2.0 ** -7 * ref.abs().max(), f"{tag}{n} {key}"
                assert (lg[0].argmax(-1).int() == torch.from_numpy(z[f"{tag}{n}_argmax"])).float().mean() >= 0.9


def test_dllm_cache_matches_reference():
    """LLaDAModelLM.forward(use_cache=True, to_compute_mask=..., cat=...): live reference (bit for bit) where it is
    mounted, the committed recording of this host class bit for bit, other classes' recordings by tolerance."""
    live = _live_reference()
    if live is not None:
        _check_dllm_cache(live.compute_dllm_cache(), exact=True)
    z, same = golden_float("dllm_cache")
    _check_dllm_cache(z, exact=same)


def test_dllm_cache_oracle_properties():
    """Size-independent properties of the restatement: a step whose mask covers every token equals a plain forward; a step
    on unchanged ids leaves the cached logits of the untouched rows exactly as they were."""
    sd, cfg = tiny_sd(), synth.CFG_TINY
    (_, ids0, _), (_, ids1, m1) = synth.dllm_cache_script()[:2]
    cache = llada_oracle.DllmCache(cfg["n_layers"])
    cache.caching(True)
    full0 = llada_oracle.forward_logits_cached(sd, cfg, ids0, cache, cat="c").clone()
    assert torch.equal(full0, llada_oracle.forward_logits(sd, cfg, ids0))
    lg = llada_oracle.forward_logits_cached(sd, cfg, ids1, cache, to_compute_mask=m1, cat="c")
    assert torch.equal(lg[~m1], full0[~m1])
    allm = torch.ones_like(m1)
    lg_all = llada_oracle.forward_logits_cached(sd, cfg, ids1, cache, to_compute_mask=allm, cat="c")
    assert torch.equal(lg_all, llada_oracle.forward_logits(sd, cfg, ids1))


def test_image_probs_against_torch_ops():
    # parallel_generator.py:282-295 evaluated with the reference's own torch ops on seeded random logits
    torch.manual_seed(3)
    B, N, CB = 1, 64, 8192
    c = (torch.randn(B, N, CB) * 1.5).to(torch.bfloat16)
    ut = (c.float() + torch.randn(B, N, CB) * 0.5).to(torch.bfloat16)
    ui = (c.float() + torch.randn(B, N, CB) * 0.5).to(torch.bfloat16)
    for cs, ci in [(0.0, 4.0), (2.3, 4.0), (3.0, 0.0), (0.0, 0.0)]:
        il = c
        if cs != 0.0:
            il = il + cs * (c - ut)
        if ci != 0.0:
            il = il + ci * (c - ui)
        probs = F.softmax(il, dim=-1)
        am_ref = probs.argmax(-1)
        pm_ref = torch.gather(probs, -1, am_ref[..., None]).squeeze(-1)
        am, pm, pr = so.image_probs(c, ut, ui, cs, ci, want_probs=True)
        assert torch.equal(am.long(), am_ref)
        assert torch.equal(bits(pm), bits(pm_ref))
        # individual bf16 probabilities may differ in the last bit at rounding boundaries (libm exp / sum order)
        assert (bits(pr) != bits(probs)).float().mean() < 1e-4


def test_text_select_against_torch_ops():
    # parallel_generator.py:185-217 with the reference's torch ops, B = 2
    torch.manual_seed(5)
    B, T, V, L, ts = 2, 24, 4096, 40, 10
    logits = (torch.randn(B, T, V) * 2).to(torch.bfloat16)
    ids = torch.randint(0, 1000, (B, L))
    ids[:, ts:ts + T] = synth.MASK
    ids[0, ts + 3] = 5
    ids[1, ts + 7] = 9
    k = [5, 9]
    masked = ids[:, ts:ts + T] == synth.MASK
    x0 = torch.argmax(logits, dim=-1)
    p = F.softmax(logits.to(torch.float64), dim=-1)
    x0_p = torch.gather(p, -1, x0[..., None]).squeeze(-1)
    x0 = torch.where(masked, x0, ids[:, ts:ts + T])
    conf = torch.where(masked, x0_p, -np.inf)
    ref = ids.clone()
    tr = torch.zeros_like(x0, dtype=torch.bool)
    for j in range(B):
        _, sel = torch.topk(conf[j], k=k[j])
        tr[j, sel] = True
    ref[:, ts:ts + T][tr] = x0[tr]
    got, conf_o, _ = so.text_select(logits, None, ids, ts, k)
    assert torch.equal(got, ref)
    m = masked
    assert torch.allclose(conf_o[m], x0_p[m], rtol=1e-12, atol=0)


def test_lfq_gather_against_reference_formula():
    # MMaDA-Parallel-M/models/modeling_magvitv2.py:186-194,208-221
    nbits = 13
    idx = torch.randint(0, 2 ** nbits, (2, 50))
    mask = 2 ** torch.arange(nbits - 1, -1, -1)
    ref = ((idx[..., None] & mask) != 0).float() * 2 - 1  # [B,N,nbits]
    assert torch.equal(so.lfq_gather(idx, nbits), ref.permute(0, 2, 1).contiguous())


# ---- M variant: MMadaModelLM.interleave_generate (MMaDA-Parallel-M/models/modeling_mmada.py:117-248) ---------------
from helpers import M_CASES, M_SHAPE  # noqa: E402


@pytest.mark.parametrize("name", list(M_CASES))
def test_m_interleave_trajectory_matches_reference(name):
    """Oracle restatement of the M sampler vs the ids the reference passed to every forward (stub logits, per-call
    seeded draws for multinomial / gumbel / float64 text noise)."""
    from oracle import interleave_oracle as io_

    z = np.load(os.path.join(GOLDEN, "m_traj.npz"))
    sh, kw = M_SHAPE, dict(M_CASES[name])
    seed = int(z[name + "_seed"])
    V = sh["text_vocab"] + sh["CB"]
    n = [0]

    def model_fn(ids):
        n[0] += 1
        return stub_logits(seed, n[0], ids.shape[0], ids.shape[1], V)

    trace = []
    img, text = io_.generate(model_fn, torch.from_numpy(z[name + "_inp"]), torch.from_numpy(z[name + "_unc"]),
                             text_cfg=kw["text_cfg"], image_cfg=kw["image_cfg"], text_steps=kw["text_steps"],
                             image_steps=kw["image_steps"], soi=sh["soi"], eoi=sh["eoi"], bos=sh["bos"],
                             mask_id=sh["mask_id"], text_vocab=sh["text_vocab"], num_vq_tokens=sh["N"],
                             codebook_size=sh["CB"], max_seq_length=sh["T"], image_temperature=kw["image_temperature"],
                             rng=io_.SeededRng(seed), text_temperature=kw["text_temperature"], trace=trace)
    got = torch.stack(trace, 0)
    ref = torch.from_numpy(z[name + "_calls"])
    assert got.shape == ref.shape
    assert torch.equal(got, ref), f"first differing forward: {(got != ref).flatten(1).any(1).nonzero()[0].item()}"
    assert torch.equal(img, torch.from_numpy(z[name + "_img"]))
    assert torch.equal(text, torch.from_numpy(z[name + "_text"]))


# ---- Gradio sampler generate_ti2ti_stepwise (app.py:143-398): same loop, image steps linspace(0, S-1, int(0.3 S)) ------
from helpers import STEPWISE_CASES  # noqa: E402


@pytest.mark.parametrize("name", list(STEPWISE_CASES))
def test_stepwise_trajectory_matches_reference(name):
    z = np.load(os.path.join(GOLDEN, "stepwise_traj.npz"))
    seed, job, kw = int(z[name + "_seed"]), tiny_job(), STEPWISE_CASES[name]
    V = STUB_TEXT_VOCAB + STUB_CB
    n = [0]

    def model_fn(ids):
        n[0] += 1
        return stub_logits(seed, n[0], ids.shape[0], ids.shape[1], V)

    S = kw["text_steps"]
    sched = torch.linspace(0, S - 1, int(S * 0.3)).round().int().tolist()
    trace = []
    generate_oracle.generate(model_fn, job["input_ids"], job["text_start"], job["text_end"], job["image_start"],
                             job["seq_len"], job["newline_every"], text_steps=S, timesteps=0, cfg_scale=kw["cfg_scale"],
                             cfg_img=kw["cfg_img"], uncon_text=job["uncon_text"], uncon_image=job["uncon_image"],
                             text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB, trace=trace, image_step_list=sched)
    assert torch.equal(torch.cat(trace, 0), torch.from_numpy(z[name + "_calls"]))


# ---- MAGVITv2 decode (MMaDA-Parallel-M models/modeling_magvitv2.py:208-221,277-433), fp32 -----------------------------
@pytest.mark.parametrize("name", ["tiny", "full"])
def test_vq_decode_oracle_matches_reference(name):
    """oracle/vq_oracle.py restates LFQuantizer.get_codebook_entry + VQGANDecoder.forward; the fixture is the output of
    the reference's own modules on the same seeded weights.  Same torch CPU ops in the same order: tolerance covers
    only thread-count-dependent conv blocking (<= 2e-5 of the output range)."""
    from oracle import vq_oracle

    z = np.load(os.path.join(GOLDEN, "vq_decode.npz"))
    cfg = synth.VQ_CFG_TINY if name == "tiny" else synth.VQ_CFG_M
    sd = synth.synthetic_vq_state_dict(cfg, int(z[name + "_seed"]))
    img = vq_oracle.decode_code(sd, cfg, torch.from_numpy(z[name + "_idx"]))
    ref = torch.from_numpy(z[name + "_out"])
    got = img if name == "tiny" else img[:, :, ::4, ::4]
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    stats = z[name + "_stats"]
    assert abs(img.std().item() - stats[1]) < 1e-5 and abs(img.abs().max().item() - stats[2]) < 1e-4


def test_lfq_codebook_entry_is_the_bit_pattern():
    from oracle import vq_oracle

    idx = torch.tensor([[0, 1, 4096, 8191]])
    z = vq_oracle.lfq_codebook_entry(idx, 13, shape=(2, 2))  # [1, 13, 2, 2]; channel 0 = MSB
    assert z[0, :, 0, 0].tolist() == [-1.0] * 13
    assert z[0, :, 0, 1].tolist() == [-1.0] * 12 + [1.0]
    assert z[0, :, 1, 0].tolist() == [1.0] + [-1.0] * 12
    assert z[0, :, 1, 1].tolist() == [1.0] * 13


@pytest.mark.parametrize("name", ["tiny", "full"])
def test_vq_get_code_oracle_matches_reference(name):
    """Encoder direction: oracle get_code vs the reference's MAGVITv2.get_code (VQGANEncoder + LFQuantizer).  The
    indices are signs of z: they must agree wherever |z| is not within fp32 noise of zero."""
    from oracle import vq_oracle

    z = np.load(os.path.join(GOLDEN, "vq_encode.npz"))
    cfg = synth.VQ_ENC_CFG_TINY if name == "tiny" else synth.VQ_ENC_CFG_M
    seed = int(z[name + "_seed"])
    sd = synth.synthetic_vq_state_dict(cfg, seed)
    B, res = (2, 16) if name == "tiny" else (1, 512)
    img = synth.synthetic_image(B, res, res, seed=200 + seed)
    idx, zz = vq_oracle.get_code(sd, cfg, img, return_z=True)
    zref = torch.from_numpy(z[name + "_z"])
    assert (zz - zref).abs().max().item() <= 2e-5 * zref.abs().max().item()
    iref = torch.from_numpy(z[name + "_idx"])
    sure = (zref.abs() > 1e-4).all(1).reshape(B, -1)  # positions whose 13 signs are all unambiguous
    assert sure.float().mean().item() > 0.95
    assert torch.equal(idx[sure], iref[sure])


# ---- generate_image (A text-to-image MaskGIT sampler, generators/image_generation_generator.py:14-251) ----------------
from helpers import T2I_CASES, t2i_job  # noqa: E402


def _t2i_calls(z, name):
    lens = z[name + "_calls_len"]
    return [torch.from_numpy(z[name + "_calls"][i, :n]).view(1, -1) for i, n in enumerate(lens)]


@pytest.mark.parametrize("name", list(T2I_CASES))
def test_t2i_trajectory_matches_reference(name):
    from oracle import generate_image_oracle

    z = np.load(os.path.join(GOLDEN, "t2i_traj.npz"))
    seed, job, kw = int(z[name + "_seed"]), t2i_job(), T2I_CASES[name]
    V = STUB_TEXT_VOCAB + STUB_CB
    n = [0]

    def model_fn(ids):
        n[0] += 1
        return stub_logits(seed, n[0], 1, ids.shape[1], V)

    gen = torch.Generator().manual_seed(int(z[name + "_gen_seed"])) if kw["temperature"] > 0 else None
    trace = []
    vq = generate_image_oracle.generate(model_fn, job["prompt"], job["seq_len"], kw["timesteps"], kw["temperature"],
                                        kw["cfg_scale"], job["uncon_ids"], job["code_start"], STUB_CB, STUB_TEXT_VOCAB,
                                        generator=gen, trace=trace)
    ref = _t2i_calls(z, name)
    assert len(trace) == len(ref)
    for i, (a, b) in enumerate(zip(trace, ref)):
        assert torch.equal(a, b), f"model call {i} differs"
    assert torch.equal(vq, torch.from_numpy(z[name + "_vq"]))


# ---- mmu_generate (M block-wise text sampler, MMaDA-Parallel-M/models/modeling_mmada.py:618-692) ----------------------
from helpers import MMU_CASES, MMU_SHAPE  # noqa: E402


@pytest.mark.parametrize("name", list(MMU_CASES))
def test_mmu_trajectory_matches_reference(name):
    from oracle import interleave_oracle as io_

    z = np.load(os.path.join(GOLDEN, "mmu_traj.npz"))
    seed, kw, sh = int(z[name + "_seed"]), MMU_CASES[name], MMU_SHAPE
    n = [0]

    def model_fn(ids):
        n[0] += 1
        return stub_logits(seed, n[0], ids.shape[0], ids.shape[1], sh["V"])

    trace = []
    x = io_.mmu_generate(model_fn, torch.from_numpy(z[name + "_idx"]), kw["max_new_tokens"], kw["steps"], kw["block_length"],
                         kw["temperature"], kw["cfg_scale"], sh["mask_id"], rng=io_.SeededRng(seed), trace=trace)
    assert torch.equal(torch.stack(trace, 0), torch.from_numpy(z[name + "_calls"]))
    assert torch.equal(x, torch.from_numpy(z[name + "_x"]))


# ---- t2i_generate (M MaskGIT text-to-image sampler, MMaDA-Parallel-M/models/modeling_mmada.py:264-359) ----------------
from helpers import M_T2I_CASES, M_T2I_SHAPE, m_t2i_job  # noqa: E402


@pytest.mark.parametrize("name", list(M_T2I_CASES))
def test_m_t2i_trajectory_matches_reference(name):
    from oracle import interleave_oracle as io_

    z = np.load(os.path.join(GOLDEN, "m_t2i_traj.npz"))
    seed, kw, sh = int(z[name + "_seed"]), M_T2I_CASES[name], M_T2I_SHAPE
    V = sh["text_vocab"] + sh["CB"]
    inp, unc = m_t2i_job(seed, kw["B"], kw["known"])
    n = [0]

    def model_fn(ids):
        n[0] += 1
        return stub_logits(seed, n[0], ids.shape[0], ids.shape[1], V)

    trace = []
    ids = io_.t2i_generate(model_fn, inp, unc if kw["uncond"] else None, kw["temperature"], kw["timesteps"],
                           kw["guidance_scale"], sh["N"], sh["mask_id"], sh["resolution"], sh["CB"], sh["text_vocab"],
                           io_.SeededRng(seed), trace=trace)
    assert torch.equal(torch.stack(trace, 0), torch.from_numpy(z[name + "_calls"]))
    assert torch.equal(ids, torch.from_numpy(z[name + "_ids"]))
    assert torch.equal(inp, torch.from_numpy(z[name + "_final_input"]))


# ---- generate_ti2ti at temperature > 0 (README defaults), every draw from a seeded CPU generator ---------------------
from helpers import NOISY_CASES  # noqa: E402


@pytest.mark.parametrize("name", list(NOISY_CASES))
def test_noisy_sampler_trajectory_matches_reference(name):
    z = np.load(os.path.join(GOLDEN, "sampler_noisy.npz"))
    calls_ref = torch.from_numpy(z[name + "_calls"])
    seed, job, kw = int(z[name + "_seed"]), tiny_job(), NOISY_CASES[name]
    V = STUB_TEXT_VOCAB + STUB_CB
    n = [0]

    def model_fn(ids):
        n[0] += 1
        return stub_logits(seed, n[0], ids.shape[0], ids.shape[1], V)

    trace = []
    gen = torch.Generator().manual_seed(int(z[name + "_gen_seed"]))
    generate_oracle.generate(model_fn, job["input_ids"], job["text_start"], job["text_end"], job["image_start"],
                             job["seq_len"], job["newline_every"], text_steps=kw["text_steps"], timesteps=kw["timesteps"],
                             cfg_scale=kw["cfg_scale"], cfg_img=kw["cfg_img"], uncon_text=job["uncon_text"],
                             uncon_image=job["uncon_image"], text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB,
                             trace=trace, temperature=kw["temperature"], text_temperature=kw["text_temperature"],
                             generator=gen)
    got = torch.cat(trace, 0)
    assert got.shape == calls_ref.shape
    assert torch.equal(got, calls_ref), f"first differing model call: {(got != calls_ref).any(1).nonzero()[0].item()}"
