"""Decision-level parity that CAN fail: the peaked synthetic checkpoint (synth.synthetic_state_dict_peaked).

With the flat random weights of the other parity tests every post-CFG arg-max is a near-tie among thousands of classes: the
reference does not reproduce its own free-running trajectory across two x86 hosts, and the flat-weights numbers
(tests/test_gpu_parity_depth.py) can only be held to an envelope.  The peaked checkpoint plants one circuit (block 0 copies the
token `delta` positions back, the LM head reads it out) on top of the same seeded random weights, so that every decision the
sampler takes has a margin far above bf16 rounding noise — and then the reference's decisions are the test:

  * free-running generate_ti2ti (BASELINE configs[0] geometry and schedule: L = 1654, 32 text + 16 image steps, 64 model calls,
    4 blocks, d = 1024) against the ids the UNMODIFIED reference recorded at every model call (tests/golden/peaked_traj.*.npz,
    oracle/gen_golden.py gen_peaked): text span >= 99 % equal at every call, image tokens >= 99 % equal wherever both runs have
    unmasked a slot, final tokens >= 99 % equal; WHICH of several exactly tied bf16 confidences stay masked at a re-mask cut is
    torch.sort's unspecified tie order in the reference (not stable for N >= 64 on this PyTorch) — the first step whose
    re-mask pattern differs must be exactly such a tie at the cut;
  * one image step at 8B depth and width (32 blocks, d = 4096, L = 2438) through the dual-CFG combine: post-CFG arg-max >= 98 %
    equal to the oracle's, and HIP-vs-fp32 agreement within one sigma of oracle-vs-fp32.
Reference lines: generators/parallel_generator.py:102-368 (loop), :282-295,311 (CFG combine, arg-max), :36-39 (confidence).
"""
import numpy as np
import pytest
import torch

from helpers import golden_float, save_parity
from mmada_parallel_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KW = dict(text_steps=32, timesteps=16, temperature=0.0, text_temperature=0.0, cfg_scale=0.0, cfg_img=4.0)


def _job():
    return synth.synthetic_job(256, 256, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)


def test_peaked_free_running_trajectory_equals_the_reference_recording():
    from mmada_parallel_amd import LLaDAForMultiModalGeneration, generate_ti2ti

    z, _ = golden_float("peaked_traj")
    calls_ref = torch.from_numpy(z["calls"].astype(np.int64))
    cfg, job = synth.CFG_PEAKED, _job()
    assert job["input_ids"].shape[1] == calls_ref.shape[1] == 1654
    sd = synth.synthetic_state_dict_peaked(cfg, synth.peaked_delta(job))
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=DEV, max_batch=2)
    calls, real_fb = [], model.forward_body

    def rec(ids, consumed=None):
        calls.extend(ids[b:b + 1].cpu().clone() for b in range(ids.shape[0]))
        return real_fb(ids, consumed=consumed)

    model.forward_body = rec
    torch.manual_seed(1234)   # the one torch.randint fill of the read-out, as in the recording
    vq, text, final = generate_ti2ti(model, job["input_ids"].to(DEV), job["text_start"], job["text_end"], job["image_start"],
                                     job["seq_len"], job["newline_every"], uncon_text=job["uncon_text"],
                                     uncon_image=job["uncon_image"], return_state=True, **KW)
    model.forward_body = real_fb
    got = torch.cat(calls, 0)
    assert got.shape == calls_ref.shape, (got.shape, calls_ref.shape)
    same = (got == calls_ref).all(1)
    ids_equal = (got == calls_ref).float().mean().item()
    pos = [i for i in range(job["image_start"], job["image_start"] + job["seq_len"] + job["seq_len"] // job["newline_every"])
           if int(job["input_ids"][0, i]) != synth.NEW_LINE]
    ts, te = job["text_start"], job["text_end"]
    text_equal = (got[:, ts:te] == calls_ref[:, ts:te]).float().mean().item()
    gi, ri = got[:, pos], calls_ref[:, pos]
    both = (gi != synth.MASK) & (ri != synth.MASK)
    img_token_equal = ((gi == ri) & both).sum().item() / max(1, int(both.sum()))
    img_mask_pattern_equal = ((gi == synth.MASK) == (ri == synth.MASK)).float().mean().item()
    vq_ref, text_ref = z["vq"].tolist(), z["text"].tolist()
    settled = [j for j, p in enumerate(pos) if int(final[0, p]) != synth.MASK]   # all but the one slot the schedule leaves masked
    vq_agree = sum(vq[j] == vq_ref[j] for j in settled) / len(settled)
    text_agree = sum(a == b for a, b in zip(text, text_ref)) / len(text_ref)
    # The first image step whose re-mask pattern differs from the reference's: every slot in the symmetric difference must
    # carry EXACTLY the reference's cut confidence — a tie at the cut, whose order the reference's torch.sort leaves to the
    # PyTorch build (oracle/generate_oracle.py tie_order; our kernel keeps the lowest indices masked).
    from mmada_parallel_amd.generators.parallel_generator import image_step_indices

    img_steps = sorted(set(image_step_indices(KW["text_steps"], KW["timesteps"])))
    conf = torch.from_numpy(z["commit_conf"].astype(np.int16)).view(torch.bfloat16).float()     # [image steps, N]
    ref_masking = torch.from_numpy(z["commit_masking"])
    call, first_tie_step, explained = 0, None, None
    for step in range(KW["text_steps"]):
        nxt = call + (3 if step in img_steps else 1)
        if step in img_steps and nxt < got.shape[0] and first_tie_step is None:
            k = img_steps.index(step)
            hip_masked = got[nxt, pos] == synth.MASK
            assert torch.equal(calls_ref[nxt, pos] == synth.MASK, ref_masking[k]), "fixture self-consistency"
            if not torch.equal(hip_masked, ref_masking[k]):
                first_tie_step = step
                cut = conf[k][ref_masking[k]].max()            # the largest confidence the reference kept masked
                diff = hip_masked ^ ref_masking[k]
                explained = bool((conf[k][diff] == cut).all()) and bool(torch.equal(got[call], calls_ref[call]))
        call = nxt
    per_call = (got == calls_ref).float().mean(1)
    rep = {"model_calls": int(got.shape[0]), "calls_identical": int(same.sum()),
           "first_diverging_call": int((~same).nonzero()[0]) if not bool(same.all()) else -1,
           "ids_equal_fraction_over_all_calls": ids_equal, "worst_call_ids_equal": per_call.min().item(),
           "text_span_ids_equal_over_all_calls": text_equal, "image_tokens_equal_where_both_unmasked": img_token_equal,
           "image_mask_pattern_equal": img_mask_pattern_equal,
           "first_image_step_with_another_remask_pattern": first_tie_step, "that_difference_is_a_tie_at_the_cut": explained,
           "final_vq_agreement": vq_agree, "final_text_agreement": text_agree,
           "distinct_vq": len(set(vq)), "distinct_text": len(set(text)),
           "reference_min_text_margin_sigma": float(z["min_text_margin_sigma"].min())}
    print("peaked checkpoint, free-running vs the reference's recording:", rep)
    save_parity("peaked_free_running_vs_reference", rep)
    assert text_equal >= 0.99 and img_token_equal >= 0.99, rep
    assert vq_agree >= 0.99 and text_agree >= 0.99, rep
    if first_tie_step is not None:
        assert explained, rep
    assert rep["distinct_vq"] > 100 and rep["distinct_text"] > 100, "the planted circuit must give position-dependent predictions"


@pytest.fixture(scope="module")
def peaked8b():
    """ONE peaked 32-block, d = 4096 checkpoint (the A job's copy distance) shared by the three full-depth decision tests, with
    the three evaluations of every branch cached: HIP, the CPU oracle (the reference's bf16 arithmetic) and the same weights in
    fp32 (exact arithmetic on bf16-representable weights)."""
    from mmada_parallel_amd import LLaDAForMultiModalGeneration
    from oracle import llada_oracle

    cfg = dict(synth.CFG_8B)
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    sd_dev = synth.synthetic_state_dict_peaked(cfg, synth.peaked_delta(job), device=DEV)
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd_dev, device=DEV, max_batch=2)
    sd = {k: v.cpu() for k, v in sd_dev.items()}
    del sd_dev
    torch.cuda.empty_cache()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    state = {"sd32": None}

    def oracle_rows(x, rows_by_seq, lo, hi, fp32=False):
        """[len(x)] lists of logits [rows, hi - lo] of the oracle forward (bf16 weights, or their fp32 copies)."""
        if fp32 and state["sd32"] is None:
            state["sd32"] = {k: v.float() for k, v in sd.items()}
        sdd = state["sd32"] if fp32 else sd
        h = llada_oracle.forward_hidden(sdd, cfg, x)
        return [llada_oracle.head(sdd, cfg, h[b:b + 1, rows_by_seq], lo, hi)[0].float() for b in range(x.shape[0])]

    return dict(cfg=cfg, job=job, model=model, oracle_rows=oracle_rows, cache={})


def _a_branches(p8):
    """Consumed image logits of the conditional, text-unconditional and image-unconditional forwards of ONE image step of
    configs[1] (L = 2438, N = 1024) — HIP as the sampler launches them (conditional alone, the unconditional pair as one
    batch-2 launch), oracle bf16, fp32."""
    if "a" in p8["cache"]:
        return p8["cache"]["a"]
    job, model = p8["job"], p8["model"]
    ids = job["input_ids"]
    L = ids.shape[1]
    N, nl = job["seq_len"], job["newline_every"]
    pos = [i for i in range(job["image_start"], job["image_start"] + N + N // nl) if int(ids[0, i]) != synth.NEW_LINE]
    unc2 = ids.repeat(2, 1)
    unc2[0, :job["uncon_text"].shape[1]] = job["uncon_text"][0]
    unc2[1, :job["uncon_image"].shape[1]] = job["uncon_image"][0]
    lo, hi = synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK
    rows = torch.tensor(pos, dtype=torch.int32, device=DEV)
    model.forward_body(ids.to(DEV))
    hc = model.head_rows(rows, lo, hi).float().cpu()
    model.forward_body(unc2.to(DEV))
    hu = model.head_rows(torch.cat([rows, rows + L]), lo, hi).float().cpu()
    allx = torch.cat([ids, unc2], 0)
    ob = p8["oracle_rows"](allx, pos, lo, hi)
    of = p8["oracle_rows"](allx, pos, lo, hi, fp32=True)
    out = dict(pos=pos, hip=(hc, hu[:len(pos)], hu[len(pos):]), oracle=tuple(ob), fp32=tuple(of))
    p8["cache"]["a"] = out
    return out


@pytest.mark.parametrize("cfg_scale,cfg_img", [(0.0, 4.0), (3.0, 4.0)])
def test_peaked_post_cfg_decisions_at_8b_depth(peaked8b, cfg_scale, cfg_img):
    """One image step of configs[1] (L = 2438, N = 1024) on a PEAKED 32-block, d = 4096 checkpoint through the CFG combine of
    parallel_generator.py:282-295 in the reference's bf16 order (the C oracle of it, pinned by tests/golden), arg-max:
    HIP vs the CPU oracle vs the same weights in fp32.  (0, 4): BASELINE configs[1]'s dual branch; (3, 4): configs[4]'s
    TRIPLE branch — c + 3 (c - u_text) + 4 (c - u_img), all three forwards contribute."""
    from oracle import sampler_oracle as so

    br = _a_branches(peaked8b)
    job, pos = peaked8b["job"], br["pos"]
    ids = job["input_ids"]
    lo = synth.TEXT_VOCAB

    def decide(c, ut, ui):   # the reference's combine + soft-max + arg-max on bf16 logits
        cb, tb, ib = (t.to(torch.bfloat16)[None].contiguous() for t in (c, ut, ui))
        return so.image_probs(cb, tb, ib, float(cfg_scale), float(cfg_img))[0][0].to(torch.int64)

    a_hip, a_ora = decide(*br["hip"]), decide(*br["oracle"])
    c32, t32, i32 = br["fp32"]
    exact = c32 + (cfg_scale * (c32 - t32) if cfg_scale else 0.0) + (cfg_img * (c32 - i32) if cfg_img else 0.0)
    a_ex = exact.argmax(-1)
    want = torch.tensor([int(ids[0, p - synth.peaked_delta(job)]) - lo for p in pos])
    n = len(pos)
    rep = {"cfg_scale": cfg_scale, "cfg_img": cfg_img, "slots": n, "hip_vs_oracle": (a_hip == a_ora).float().mean().item(),
           "hip_vs_fp32": (a_hip == a_ex).float().mean().item(), "oracle_vs_fp32": (a_ora == a_ex).float().mean().item(),
           "fp32_copies_planted_token": (a_ex == want).float().mean().item(), "distinct_codes": len(set(a_ex.tolist())),
           "text_branch_differs_from_cond_argmax": (br["fp32"][1].argmax(-1) != c32.argmax(-1)).float().mean().item()}
    sigma = (rep["oracle_vs_fp32"] * (1 - rep["oracle_vs_fp32"]) / n) ** 0.5
    rep["one_sigma_of_oracle_vs_fp32"] = sigma
    print("peaked checkpoint, one image step at 8B depth through the CFG combine:", rep)
    save_parity("peaked_post_cfg_full_depth_8b" + ("_triple_branch" if cfg_scale else ""), rep)
    assert rep["distinct_codes"] > 300
    assert rep["hip_vs_oracle"] >= 0.98, rep
    assert rep["hip_vs_fp32"] >= rep["oracle_vs_fp32"] - max(sigma, 1.0 / n), rep


def test_m_teacher_forced_step_at_8b_depth(peaked8b):
    """MMaDA-Parallel-M, ONE step of interleave_generate (models/modeling_mmada.py:163-246) at 8B depth and width in the M
    layout: the batch-2 forward (cond || uncond) at L = 2349 (Lp = 2352 — not the 2440-row panels of every A test), then the text
    step (cond + 2.5 (uncond - cond), fp64 soft-max confidence, top-k commit: mmada_text_select_cfg on the real head rows) and
    the image step's (1 + 4) cond - 4 uncond soft-max (mmada_image_probs_m), HIP vs the C oracle of the same lines fed with
    the CPU oracle's logits, vs fp32.  The checkpoint is the A job's (copy distance 1084): the 18 image slots whose source lies
    before the input image codes have no planted prediction and are left out."""
    from mmada_parallel_amd import abi
    from oracle import sampler_oracle as so

    model = peaked8b["model"]
    lib, h = model._lib, model._handle
    mj = synth.m_peaked_job()
    N, T, P, L = mj["N"], mj["T"], mj["P"], mj["L"]
    assert L == 2349
    tail = torch.cat([torch.tensor([mj["soi"]]), torch.full((N,), synth.MASK), torch.tensor([mj["eoi"], mj["bos"]]),
                      torch.full((T - 1,), synth.MASK)])
    both = torch.stack([torch.cat([mj["input_ids"], tail]), torch.cat([mj["uncond_input_ids"], tail])], 0)
    i0, ts = mj["img_start"], mj["text_start"]
    lo, CB, V = synth.TEXT_VOCAB, synth.CODEBOOK, model.vocab
    delta = synth.peaked_delta(peaked8b["job"])
    live = torch.tensor([2 <= (i0 + j) - delta < 2 + N for j in range(N)])
    K = 48
    ids_dev = both.to(DEV)
    model.forward_body(ids_dev, consumed=(i0, L))
    ar = torch.arange(ts, L, dtype=torch.int32, device=DEV)
    pm = torch.arange(i0, i0 + N, dtype=torch.int32, device=DEV)
    tl = model.head_rows(torch.cat([ar, ar + L]), 0, V)
    il = model.head_rows(torch.cat([pm, pm + L]), lo, lo + CB)
    st = abi.stream_ptr()
    k_dev = torch.tensor([K], dtype=torch.int32, device=DEV)
    scratch = torch.empty(T * 16, dtype=torch.uint8, device=DEV)
    hip_ids = ids_dev[:1].clone()
    abi.check(lib.mmada_text_select_cfg(h, tl[:T].data_ptr(), tl[T:].data_ptr(), 2.5, None, 1, T, V, V, hip_ids.data_ptr(), L, ts,
                                        k_dev.data_ptr(), scratch.data_ptr(), st), "mmada_text_select_cfg")
    probs = torch.empty((N, CB), dtype=torch.bfloat16, device=DEV)
    am = torch.empty((1, N), dtype=torch.int32, device=DEV)
    pmax = torch.empty((1, N), dtype=torch.bfloat16, device=DEV)
    abi.check(lib.mmada_image_probs_m(h, il[:N].data_ptr(), il[N:].data_ptr(), 1, N, CB, 4.0, probs.data_ptr(), am.data_ptr(),
                                      pmax.data_ptr(), st), "mmada_image_probs_m")
    torch.cuda.synchronize()
    hip_x0 = scratch[T * 8: T * 12].view(torch.int32).cpu().to(torch.int64)
    hip_text = hip_ids[0, ts:].cpu()
    hip_am = am[0].cpu().to(torch.int64)

    def oracle_side(fp32):
        rows = list(range(i0, i0 + N)) + list(range(ts, L))
        full = peaked8b["oracle_rows"](both, rows, 0, V, fp32=fp32)     # [2] x [N + T, V]
        return [f[:N, lo:lo + CB] for f in full], [f[N:] for f in full]

    (ic, iu), (tc, tu) = oracle_side(False)
    o_ids, _, o_x0 = so.text_select_cfg(tc.to(torch.bfloat16)[None].contiguous(), tu.to(torch.bfloat16)[None].contiguous(), 2.5,
                                        both[:1], ts, [K], mask_id=synth.MASK)
    o_am = so.image_probs_m(ic.to(torch.bfloat16)[None].contiguous(), iu.to(torch.bfloat16)[None].contiguous(), 4.0)[0][0].to(torch.int64)
    (ic32, iu32), (tc32, tu32) = oracle_side(True)
    e_am = ((1 + 4.0) * ic32 - 4.0 * iu32).argmax(-1)
    e_x0 = (tc32 + 2.5 * (tu32 - tc32)).argmax(-1)
    masked = both[0, ts:] == synth.MASK
    o_x0 = o_x0[0].to(torch.int64)
    nl = int(live.sum())
    rep = {"L": L, "image_slots_with_a_planted_source": nl,
           "image_argmax_hip_vs_oracle": (hip_am[live] == o_am[live]).float().mean().item(),
           "image_argmax_hip_vs_fp32": (hip_am[live] == e_am[live]).float().mean().item(),
           "image_argmax_oracle_vs_fp32": (o_am[live] == e_am[live]).float().mean().item(),
           "image_distinct_codes": len(set(e_am[live].tolist())),
           "text_x0_hip_vs_oracle": (hip_x0[masked] == o_x0[masked]).float().mean().item(),
           "text_x0_hip_vs_fp32": (hip_x0[masked] == e_x0[masked]).float().mean().item(),
           "text_x0_oracle_vs_fp32": (o_x0[masked] == e_x0[masked]).float().mean().item(),
           "text_committed": int((hip_text != synth.MASK).sum() - (both[0, ts:] != synth.MASK).sum()),
           "text_committed_positions_equal": float(((hip_text != synth.MASK) == (o_ids[0, ts:] != synth.MASK)).float().mean()),
           "text_committed_tokens_equal_where_both": float((hip_text == o_ids[0, ts:])[(hip_text != synth.MASK) & (o_ids[0, ts:] != synth.MASK)].float().mean())}
    print("M layout, one teacher-forced step at 8B depth:", rep)
    save_parity("m_teacher_forced_step_full_depth_8b", rep)
    assert rep["text_committed"] == K
    assert rep["image_distinct_codes"] > 300
    s_img = (rep["image_argmax_oracle_vs_fp32"] * (1 - rep["image_argmax_oracle_vs_fp32"]) / nl) ** 0.5
    assert rep["image_argmax_hip_vs_oracle"] >= 0.98 and rep["image_argmax_hip_vs_fp32"] >= rep["image_argmax_oracle_vs_fp32"] - max(s_img, 1.0 / nl), rep
    assert rep["text_x0_hip_vs_oracle"] >= 0.98 and rep["text_committed_tokens_equal_where_both"] >= 0.98, rep
    assert rep["text_committed_positions_equal"] >= 0.9, rep


def test_flat_weights_post_cfg_envelope_over_many_jobs():
    """Round-3 review: on ONE image step of the FLAT synthetic 8B checkpoint the HIP path agreed with exact fp32 arithmetic on
    63.0 % of the post-CFG arg-maxima against 68.4 % for the reference's own bf16 evaluation (1024 slots: 2.5 sigma).  Settle
    it with more samples: tests/golden/postcfg_flat_8b.npz (oracle/gen_postcfg_flat.py, build container) holds, for several
    jobs, the oracle-bf16 and the fp32 post-CFG arg-max of every slot.  Here: the HIP arg-maxima of the same jobs, with the
    unconditional branch taken (a) from the batch-2 launch the sampler issues (M = 4876 GEMM panels) and (b) from a batch-1
    launch (M = 2438, the conditional branch's blocking) — if c - u decorrelated because the two branches run different GEMM
    blockings, (b) would agree with fp32 visibly more often than (a)."""
    import os

    from helpers import GOLDEN
    from mmada_parallel_amd import LLaDAForMultiModalGeneration
    from oracle import sampler_oracle as so

    path = os.path.join(GOLDEN, "postcfg_flat_8b.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/postcfg_flat_8b.npz not generated")
    z = np.load(path)
    seeds = [int(s) for s in z["seeds"] if f"am_fp32_{int(s)}" in z.files]
    assert seeds
    cfg = dict(synth.CFG_8B)
    sd = synth.synthetic_state_dict(cfg, seed=3, device="cpu")   # the fixture's weights: the CPU generator's draws (a device
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=DEV, max_batch=2)   # generator differs)
    del sd
    torch.cuda.empty_cache()
    lo, hi = synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK

    def combine(c, u):   # the reference's bf16 combine at cfg_scale 0, cfg_img 4 (C oracle of parallel_generator.py:282-295)
        cb, ub = c.to(torch.bfloat16)[None].contiguous(), u.to(torch.bfloat16)[None].contiguous()
        return so.image_probs(cb, ub, ub, 0.0, 4.0)[0][0].to(torch.int32)

    tot = {"slots": 0, "oracle": 0, "hip_batch2": 0, "hip_batch1": 0, "hip_b2_only": 0, "oracle_only": 0, "hip_b2_eq_oracle": 0}
    for seed in seeds:
        job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=seed)
        ids = job["input_ids"]
        L = ids.shape[1]
        N, nl = job["seq_len"], job["newline_every"]
        pos = [i for i in range(job["image_start"], job["image_start"] + N + N // nl) if int(ids[0, i]) != synth.NEW_LINE]
        rows = torch.tensor(pos, dtype=torch.int32, device=DEV)
        unc2 = ids.repeat(2, 1)
        unc2[0, :job["uncon_text"].shape[1]] = job["uncon_text"][0]
        unc2[1, :job["uncon_image"].shape[1]] = job["uncon_image"][0]
        model.forward_body(ids.to(DEV))
        c = model.head_rows(rows, lo, hi).cpu()
        model.forward_body(unc2.to(DEV))                       # what the sampler launches: the unconditional pair as one batch
        u_b2 = model.head_rows(rows + L, lo, hi).cpu()
        model.forward_body(unc2[1:2].to(DEV))                  # the same sequence alone
        u_b1 = model.head_rows(rows, lo, hi).cpu()
        am32 = torch.from_numpy(z[f"am_fp32_{seed}"])
        amo = torch.from_numpy(z[f"am_oracle_{seed}"])
        a2, a1 = combine(c, u_b2), combine(c, u_b1)
        tot["slots"] += len(pos)
        tot["oracle"] += int((amo == am32).sum())
        tot["hip_batch2"] += int((a2 == am32).sum())
        tot["hip_batch1"] += int((a1 == am32).sum())
        tot["hip_b2_only"] += int(((a2 == am32) & (amo != am32)).sum())
        tot["oracle_only"] += int(((a2 != am32) & (amo == am32)).sum())
        tot["hip_b2_eq_oracle"] += int((a2 == amo).sum())
    n = tot["slots"]
    rep = {"jobs": len(seeds), "slots": n}
    for k in ("oracle", "hip_batch2", "hip_batch1"):
        p = tot[k] / n
        rep[f"{k}_vs_fp32"] = p
        rep[f"{k}_vs_fp32_ci95"] = 1.96 * (p * (1 - p) / n) ** 0.5
    rep["hip_batch2_vs_oracle"] = tot["hip_b2_eq_oracle"] / n
    # paired: slots where exactly one of the two bf16 evaluations matches exact arithmetic (sign test)
    b, c_ = tot["hip_b2_only"], tot["oracle_only"]
    rep["paired_hip_only_right"], rep["paired_oracle_only_right"] = b, c_
    rep["paired_z"] = (b - c_) / max(1.0, (b + c_) ** 0.5)
    print("flat weights, post-CFG arg-max vs exact fp32 arithmetic over", len(seeds), "jobs:", rep)
    save_parity("post_cfg_flat_envelope_many_jobs", rep)
    sigma = (rep["oracle_vs_fp32"] * (1 - rep["oracle_vs_fp32"]) / n) ** 0.5
    assert rep["hip_batch2_vs_fp32"] >= rep["oracle_vs_fp32"] - 4 * sigma - 0.01, rep


def test_m_peaked_free_running_interleave_generate_equals_the_reference_recording():
    """MMaDA-Parallel-M END TO END with a real forward (round-4 review: the M sampler had only run on stub logits on the GPU):
    MMadaModelLM.interleave_generate free-running on the peaked checkpoint at BASELINE configs[3] geometry — L = 2349, every step
    ONE batch-2 (cond || uncond) HIP forward, mmada_text_select_cfg / mmada_image_probs_m / mmada_image_commit_m on the real head
    rows, text_cfg 2.5, image_cfg 4 — against the ids the UNMODIFIED reference (its own sampler on its own LLaDAModelLM,
    oracle/gen_golden.py gen_m_peaked -> tests/golden/m_peaked_traj.*.npz) passed to every forward, with the reference's
    multinomial / uniform draws replayed (SeededRng).  models/modeling_mmada.py:117-248, MMaDA-Parallel-M/inference.py:113-127."""
    from types import SimpleNamespace

    from mmada_parallel_amd import MMadaModelLM
    from oracle.interleave_oracle import SeededRng

    z, _ = golden_float("m_peaked_traj")
    ref = torch.from_numpy(z["calls"].astype(np.int64))          # [steps, 2, L]
    job, kw = synth.m_peaked_job(), dict(synth.M_PEAKED_KW)
    cfg = synth.CFG_PEAKED
    assert ref.shape == (kw["text_steps"], 2, job["L"]) and job["L"] == 2349
    sd = synth.synthetic_state_dict_peaked(cfg, job["delta"], beta=synth.M_PEAKED_BETA)
    model = MMadaModelLM.from_state_dict(synth.full_config(cfg), sd, device=DEV, max_batch=2)

    class Tok:
        bos_token_id = job["bos"]

        def __len__(self):
            return job["text_vocab"]

    cfgobj = SimpleNamespace(model=SimpleNamespace(mmada=SimpleNamespace(num_vq_tokens=job["N"], codebook_size=job["codebook"])),
                             dataset=SimpleNamespace(preprocessing=SimpleNamespace(max_seq_length=job["T"])))
    trace = []
    img, text = model.interleave_generate(job["input_ids"], job["uncond_input_ids"],
                                          reserved_token_mapping={"<|soi|>": job["soi"], "<|eoi|>": job["eoi"]}, config=cfgobj,
                                          uni_prompting=SimpleNamespace(text_tokenizer=Tok()), rng=SeededRng(53), trace=trace, **kw)
    got = torch.stack(trace, 0)
    assert got.shape == ref.shape
    i0, ts, N = job["img_start"], job["text_start"], job["N"]
    gi, ri = got[:, 0, i0:i0 + N], ref[:, 0, i0:i0 + N]
    both = (gi != synth.MASK) & (ri != synth.MASK)
    per_call = (got == ref).float().mean((1, 2))
    img_ref, text_ref = torch.from_numpy(z["img"]), torch.from_numpy(z["text"])
    rep = {"forwards": int(got.shape[0]), "L": int(got.shape[2]), "calls_identical": int((got == ref).all(2).all(1).sum()),
           "ids_equal_fraction_over_all_calls": (got == ref).float().mean().item(), "worst_call_ids_equal": per_call.min().item(),
           "text_span_ids_equal_over_all_calls": (got[:, 0, ts:] == ref[:, 0, ts:]).float().mean().item(),
           "image_tokens_equal_where_both_unmasked": ((gi == ri) & both).sum().item() / max(1, int(both.sum())),
           "image_mask_pattern_equal": ((gi == synth.MASK) == (ri == synth.MASK)).float().mean().item(),
           "final_image_ids_agreement": (img.cpu() == img_ref).float().mean().item(),
           "final_text_ids_agreement": (text.cpu() == text_ref).float().mean().item(),
           "distinct_image_ids": len(set(img.cpu().reshape(-1).tolist())), "distinct_text_ids": len(set(text.cpu().reshape(-1).tolist())),
           "reference_min_one_minus_text_conf": float(z["one_minus_top_text_conf"].min())}
    print("M variant, peaked checkpoint, free-running interleave_generate vs the reference's recording:", rep)
    save_parity("m_peaked_free_running_vs_reference", rep)
    assert rep["ids_equal_fraction_over_all_calls"] >= 0.99 and rep["worst_call_ids_equal"] >= 0.98, rep
    assert rep["text_span_ids_equal_over_all_calls"] >= 0.99 and rep["image_tokens_equal_where_both_unmasked"] >= 0.99, rep
    assert rep["final_image_ids_agreement"] >= 0.99 and rep["final_text_ids_agreement"] >= 0.99, rep
    assert rep["distinct_image_ids"] > 300, "the planted circuit must give position-dependent predictions"
