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


@pytest.mark.parametrize("seed", [1])
def test_peaked_post_cfg_decisions_at_8b_depth(seed):
    """One image step of configs[1] (L = 2438, N = 1024) on a PEAKED 32-block, d = 4096 checkpoint: conditional and
    unconditional forwards, c + 4 (c - u_img) in the reference's bf16 order, arg-max.  HIP vs the CPU oracle (the
    reference's arithmetic) vs the same weights evaluated in fp32."""
    from mmada_parallel_amd import LLaDAForMultiModalGeneration
    from oracle import llada_oracle

    cfg = dict(synth.CFG_8B)
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=seed)
    ids = job["input_ids"]
    L = ids.shape[1]
    sd = synth.synthetic_state_dict_peaked(cfg, synth.peaked_delta(job), device=DEV)
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=DEV, max_batch=2)
    sd = {k: v.cpu() for k, v in sd.items()}
    N, nl = job["seq_len"], job["newline_every"]
    pos = [i for i in range(job["image_start"], job["image_start"] + N + N // nl) if int(ids[0, i]) != synth.NEW_LINE]
    unc = ids.clone()
    unc[0, :job["uncon_image"].shape[1]] = job["uncon_image"][0]
    lo, hi = synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK
    rows = torch.tensor(pos, dtype=torch.int32, device=DEV)

    def hip(x):
        model.forward_body(x.to(DEV))
        return model.head_rows(rows, lo, hi).float().cpu()

    def cfg_combine(c, u):   # parallel_generator.py:282-295 at cfg_scale = 0: bf16 tensors, one rounding per operation
        c, u = c.to(torch.bfloat16), u.to(torch.bfloat16)
        return (c + 4.0 * (c - u)).float()

    hip_f = cfg_combine(hip(ids), hip(unc))
    torch.set_num_threads(min(32, torch.get_num_threads()))

    def oracle(x, sdd):
        h = llada_oracle.forward_hidden(sdd, cfg, x)
        return llada_oracle.head(sdd, cfg, h[:, pos], lo, hi)[0].float()

    ora_f = cfg_combine(oracle(ids, sd), oracle(unc, sd))
    sd32 = {k: v.float() for k, v in sd.items()}   # exact arithmetic on the same bf16-representable weights
    c32, u32 = oracle(ids, sd32), oracle(unc, sd32)
    del sd32
    exact = c32 + 4.0 * (c32 - u32)
    a_hip, a_ora, a_ex = hip_f.argmax(-1), ora_f.argmax(-1), exact.argmax(-1)
    want = torch.tensor([int(ids[0, p - synth.peaked_delta(job)]) - lo for p in pos])
    n = len(pos)
    top2 = exact.topk(2, -1).values
    rep = {"slots": n, "hip_vs_oracle": (a_hip == a_ora).float().mean().item(), "hip_vs_fp32": (a_hip == a_ex).float().mean().item(),
           "oracle_vs_fp32": (a_ora == a_ex).float().mean().item(), "fp32_copies_planted_token": (a_ex == want).float().mean().item(),
           "distinct_codes": len(set(a_ex.tolist())),
           "min_margin_over_max_err_hip": ((top2[:, 0] - top2[:, 1]) / (hip_f - exact).abs().max(-1).values.clamp_min(1e-9)).min().item()}
    sigma = (rep["oracle_vs_fp32"] * (1 - rep["oracle_vs_fp32"]) / n) ** 0.5
    rep["one_sigma_of_oracle_vs_fp32"] = sigma
    print("peaked checkpoint, one image step at 8B depth through the CFG combine:", rep)
    save_parity("peaked_post_cfg_full_depth_8b", rep)
    assert rep["distinct_codes"] > 300
    assert rep["hip_vs_oracle"] >= 0.98, rep
    assert rep["hip_vs_fp32"] >= rep["oracle_vs_fp32"] - max(sigma, 1.0 / n), rep


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
