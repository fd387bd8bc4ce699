"""Full-size (BASELINE config 2 shapes: d=4096, F=12288, 32 heads, L=2438, N=1024, CB=8192, T=256, V=134656) checks.

The oracle cannot run a whole 8B forward in test time, so at full size the tests use (a) ONE 8B-shape block against the
CPU oracle, and (b) size-independent properties of the path: batch invariance, run-to-run determinism, and the
counting invariants of the sampler (exactly k text tokens unmasked per step, known image tokens never re-masked,
exactly mask_len image tokens left masked, every written id inside the codebook range)."""
import pytest
import torch

from mmada_parallel_amd import abi, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def block8b():
    from mmada_parallel_amd import LLaDAForMultiModalGeneration

    cfg = dict(synth.CFG_8B, n_layers=1)
    sd = synth.synthetic_state_dict(cfg, seed=3, device="cpu")
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=DEV, max_batch=2)
    return cfg, sd, model


def test_one_8b_block_vs_oracle_at_full_length(block8b):
    from oracle import llada_oracle

    cfg, sd, model = block8b
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids = job["input_ids"]
    assert ids.shape[1] == 2438
    model.forward_body(ids.to(DEV))
    got = model.hidden_state().float().cpu()
    ref = llada_oracle.forward_hidden(sd, cfg, ids).float()
    scale = ref.abs().max().item()
    err = (got - ref).abs()
    print(f"8B block, L=2438: max|err|={err.max():.4g} mean|err|={err.mean():.4g} scale={scale:.4g}")
    # measured (round 2, profiles/r02_parity.json): max 1.02e-2, mean 7.0e-4 of the stream's magnitude after one 8B block
    assert err.max().item() < 1.5e-2 * scale and err.mean().item() < 1.1e-3 * scale
    # consumed LM-head rows: text span x full vocabulary, image positions x codebook slab
    ts, te = job["text_start"], job["text_end"]
    rows = torch.arange(ts, ts + 8, dtype=torch.int32, device=DEV)
    lg = model.head_rows(rows, 0, cfg["embedding_size"]).float().cpu()
    lr = llada_oracle.head(sd, cfg, llada_oracle.forward_hidden(sd, cfg, ids)[:, ts:ts + 8])[0].float()
    lerr, lscale = (lg - lr).abs(), lr.abs().max().item()
    print(f"8B block, 8 text rows x V logits: max|err|={lerr.max():.4g} mean|err|={lerr.mean():.4g} scale={lscale:.4g}")
    from helpers import save_parity

    save_parity("one_8b_block_L2438", {"stream_max_rel": err.max().item() / scale, "stream_mean_rel": err.mean().item() / scale,
                                       "logits_max_rel": lerr.max().item() / lscale, "logits_mean_rel": lerr.mean().item() / lscale})
    assert lerr.max().item() < 1.5e-2 * lscale and lerr.mean().item() < 1.5e-3 * lscale


def test_batch_invariance_and_determinism_full_length(block8b):
    _, _, model = block8b
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids = job["input_ids"].to(DEV)
    other = ids.clone()
    other[0, :64] = torch.arange(1000, 1064, device=DEV)
    model.forward_body(ids)
    a = model.hidden_state().clone()
    model.forward_body(ids)
    assert torch.equal(model.hidden_state(), a), "run-to-run determinism"
    model.forward_body(torch.cat([other, ids], 0))
    c = model.hidden_state()
    assert torch.equal(c[1], a[0]), "a sequence's result must not depend on what shares the batch"


def test_block_is_bit_identical_under_every_gemm_configuration(block8b):
    """One 8B block (QKV + rotary, attn_out + residual, gate/up + SiLU*mul, down + residual) with every GEMM of it pinned to
    each tile configuration in turn: the fused epilogues of the 8-phase kernel (transposed accumulator, 16-byte accesses after
    the half-row lane exchange, both read schedules) and of the 16-wave kernel (one element per access) must produce the
    same bits — the planner's choice, which depends on the batch, may never show in a result."""
    _, _, model = block8b
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids1 = job["input_ids"].to(DEV)
    ids2 = torch.cat([ids1, ids1.flip(1)], 0)
    lib = abi.lib()
    try:
        for ids in (ids1, ids2, ids1[:, :333]):
            ref = None
            for code in (-1, 0, 1, 2, 3, 1160, 1256, 1320):
                abi.check(lib.mmada_set_option(b"gemm_config", code), "set_option")
                model.forward_body(ids)
                got = model.hidden_state().clone()
                assert torch.isfinite(got.float()).all()
                if ref is None:
                    ref = got
                assert torch.equal(got, ref), f"gemm_config {code}, ids {tuple(ids.shape)}: {int((got != ref).sum())} elements differ"
    finally:
        lib.mmada_set_option(b"gemm_config", -1)


def test_sampler_counting_invariants_full_size(block8b):
    """generate_ti2ti at the full sampler sizes on stub logits: integer invariants that hold for any logits."""
    from mmada_parallel_amd import LLaDAForMultiModalGeneration
    from mmada_parallel_amd.generators.parallel_generator import (generate_ti2ti, get_num_transfer_tokens,
                                                                    image_step_indices, mask_len_schedule)

    _, _, real = block8b
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    L, V = job["input_ids"].shape[1], 134656
    ts, te = job["text_start"], job["text_end"]
    N, steps, tsteps = job["seq_len"], 16, 8
    pos = [i for i in range(job["image_start"], job["image_start"] + N + N // job["newline_every"])
           if int(job["input_ids"][0, i]) != synth.NEW_LINE]
    snaps = []

    class Stub(LLaDAForMultiModalGeneration):
        def __init__(self):
            self.__dict__.update(real.__dict__)
            self.n = 0

        def __del__(self):
            pass

        def forward_body(self, ids, consumed=None):
            self.n += 1
            if ids.shape[0] == 1:
                snaps.append(ids.cpu().clone())
            g = torch.Generator(device=DEV).manual_seed(self.n)
            self._h = torch.randn(ids.shape[0] * ids.shape[1], 64, device=DEV, generator=g)
            self._w = torch.randn(V, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(99))

        def head_rows(self, rows, c0, c1, out=None):   # low-rank random logits: cheap at [256, 134656]
            r = (self._h[rows.long()] @ self._w[c0:c1].t()).to(torch.bfloat16).contiguous()
            return r if out is None else out.copy_(r)

    stub = Stub()
    vq, text, final = generate_ti2ti(stub, job["input_ids"].to(DEV), ts, te, job["image_start"], N, job["newline_every"],
                                     text_steps=steps, timesteps=tsteps, temperature=0.0, text_temperature=0.0,
                                     cfg_scale=0.0, cfg_img=4.0, uncon_text=job["uncon_text"],
                                     uncon_image=job["uncon_image"], return_state=True)
    snaps.append(final)
    k_sched = get_num_transfer_tokens(job["input_ids"][:, ts:te] == synth.MASK, steps)[0].tolist()
    img_steps = set(image_step_indices(steps, tsteps))
    mlen = mask_len_schedule(N, steps)
    unknown = N
    for s in range(steps):
        before, after = snaps[s][0], snaps[s + 1][0]
        # text: exactly k[s] positions leave MASK, none re-enters it, nothing outside the spans changes
        was, now = before[ts:te] == synth.MASK, after[ts:te] == synth.MASK
        assert int(was.sum() - now.sum()) == k_sched[s] and not (now & ~was).any()
        keep = torch.ones(L, dtype=torch.bool)
        keep[ts:te] = False
        keep[pos] = False
        assert torch.equal(before[keep], after[keep])
        bi, ai = before[pos], after[pos]
        if s in img_steps:
            known = bi != synth.MASK
            assert torch.equal(ai[known], bi[known]), "a known image token must never change or be re-masked"
            expect = max(1, min(unknown - 1, mlen[s]))
            assert int((ai == synth.MASK).sum()) == expect
            unknown = expect
            vals = ai[ai != synth.MASK] - synth.TEXT_VOCAB
            assert int(vals.min()) >= 0 and int(vals.max()) < synth.CODEBOOK
        else:
            assert torch.equal(bi, ai)
    assert unknown == 1 and len(vq) == N and len(text) == 256 and all(0 <= v < synth.CODEBOOK for v in vq)


@pytest.mark.parametrize("which", ["cond_text_only", "cond_image_step", "uncond_pair"])
def test_consumed_row_window_bit_identical_at_8b_shapes(block8b, which):
    """The three row windows generate_ti2ti uses at config 2 (L = 2438), on an 8B-shape block that is also the LAST block:
    the logits of the consumed rows must equal a full forward bit for bit although every GEMM of the windowed pass runs
    with a different row count (and hence a different row-tile height) and the attention writes compact rows."""
    _, _, model = block8b
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids = job["input_ids"].to(DEV)
    L = ids.shape[1]
    ts, te, i0 = job["text_start"], job["text_end"], job["image_start"]
    img_end = ts - 1
    if which == "cond_text_only":
        win, B = (ts, te), 1
    elif which == "cond_image_step":
        win, B = (i0, te), 1
    else:
        win, B = (i0, img_end), 2
        other = ids.clone()
        other[0, :64] = torch.arange(1000, 1064, device=DEV)
        ids = torch.cat([ids, other], 0)
    rows = (torch.arange(B, device=DEV)[:, None] * L + torch.arange(win[0], win[1], device=DEV)[None, :]).reshape(-1).to(torch.int32)
    c0 = synth.TEXT_VOCAB
    model.forward_body(ids)
    full = model.head_rows(rows, c0, c0 + 1024).clone()
    model.forward_body(ids, consumed=win)
    assert torch.equal(model.head_rows(rows, c0, c0 + 1024), full)
    model.forward_body(ids)


def test_config0_shape_end_to_end_vs_oracle_teacher_forced():
    """BASELINE configs[0] geometry (256x256 output: N = 256, newline every 16, L = 1654, text_steps 32, timesteps 16,
    cfg_img 4) on a 2-block 8B-head-width model, ALL 32 steps, text AND image decisions, teacher-forced: at every step
    the ids the GPU trajectory fed to the model (conditional and both unconditional sequences) are given to the CPU oracle
    (forward + C sampler), and the oracle's decisions for that step are compared with what the GPU committed:
      * text: token committed at every position the GPU unmasked; positions unmasked (set equality);
      * image: the sampled token of every still-masked slot (before re-masking) and the set of slots left masked.
    A differing decision must be a near-tie of the ORACLE's own numbers; counts and worst margins are recorded in
    gpurun_out/r02_parity.json."""
    from mmada_parallel_amd import LLaDAForMultiModalGeneration
    from mmada_parallel_amd.generators.parallel_generator import _ti2ti_steps, get_num_transfer_tokens, mask_len_schedule
    from oracle import llada_oracle
    from oracle import sampler_oracle as so
    from helpers import host_threads as _host_threads, save_parity as _save

    _host_threads()
    cfg = dict(synth.CFG_8B, n_layers=2, d_model=1024, n_heads=8, n_kv_heads=8, mlp_hidden_size=2048)
    sd = synth.synthetic_state_dict(cfg, seed=5, device="cpu")
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=DEV, max_batch=2)
    job = synth.synthetic_job(256, 256, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids0 = job["input_ids"]
    assert ids0.shape[1] == 1654 and job["seq_len"] == 256 and job["newline_every"] == 16
    ts, te, N = job["text_start"], job["text_end"], job["seq_len"]
    pos = [i for i in range(job["image_start"], job["image_start"] + N + N // 16) if int(ids0[0, i]) != synth.NEW_LINE]
    steps, tsteps = 32, 16
    cond_in, unc_in = [], {}
    real_fb = model.forward_body
    cur = {"step": 0}

    def recording_fb(ids, consumed=None):
        if ids.shape[0] == 1:
            cond_in.append(ids.cpu().clone())
        else:
            unc_in[cur["step"]] = ids.cpu().clone()
        return real_fb(ids, consumed=consumed)

    model.forward_body = recording_fb
    after, sampled = [], {}
    with torch.no_grad():
        for step, ids, info in _ti2ti_steps(model, ids0.to(DEV), ts, te, job["image_start"], N, 16, text_steps=steps,
                                            timesteps=tsteps, temperature=0.0, text_temperature=0.0, cfg_scale=0.0,
                                            cfg_img=4.0, uncon_text=job["uncon_text"], uncon_image=job["uncon_image"]):
            if step >= steps:
                break
            after.append(ids.cpu().clone())
            if info["sampled"] is not None:
                sampled[step] = info["sampled"].cpu().clone()
            cur["step"] = step + 1
    model.forward_body = real_fb
    assert len(cond_in) == steps and len(after) == steps and len(sampled) == len(unc_in) > 0
    assert not bool((after[-1][0, ts:te] == synth.MASK).any())

    k_sched = get_num_transfer_tokens(ids0[:, ts:te] == synth.MASK, steps)[0].tolist()
    mlen = mask_len_schedule(N, steps)
    lo, hi = synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK
    st = dict(text_committed=0, text_token_diff=0, text_position_diff=0, img_slots=0, img_token_diff=0, img_mask_diff=0,
              img_steps=0, worst_text_margin_sigma=0.0, worst_img_prob_ratio=1.0, worst_text_conf_gap=0.0)
    for s in range(steps):
        before = cond_in[s]
        x = llada_oracle.forward_hidden(sd, cfg, before)
        tl = llada_oracle.head(sd, cfg, x[:, ts:te]).contiguous()                      # [1, T, V] bf16
        ids_o, conf_o, x0_o = so.text_select(tl, None, before, ts, [k_sched[s]])
        was = before[0, ts:te] == synth.MASK
        got_t, ora_t = after[s][0, ts:te], ids_o[0, ts:te]
        g_new = was & (got_t != synth.MASK)
        o_new = was & (ora_t != synth.MASK)
        assert int(g_new.sum()) == k_sched[s] == int(o_new.sum())
        st["text_committed"] += int(g_new.sum())
        tlf = tl[0].float()
        sig = tlf.std().item()
        for t in g_new.nonzero().flatten().tolist():
            if int(got_t[t]) != int(x0_o[0, t]):                                       # token differs from the oracle arg-max
                st["text_token_diff"] += 1
                top = tlf[t].max().item()
                gap = (top - tlf[t, int(got_t[t])].item()) / sig
                st["worst_text_margin_sigma"] = max(st["worst_text_margin_sigma"], gap)
        for t in (g_new ^ o_new).nonzero().flatten().tolist():                         # a different position was unmasked
            st["text_position_diff"] += 1
            # confidence of this position vs the oracle's k-th best: how close to the cut was it
            c = conf_o[0][torch.isfinite(conf_o[0])]
            kth = c.sort(descending=True).values[k_sched[s] - 1].item() if k_sched[s] > 0 else 0.0
            st["worst_text_conf_gap"] = max(st["worst_text_conf_gap"], abs(conf_o[0, t].item() - kth) / max(kth, 1e-30))
        if s in sampled:
            st["img_steps"] += 1
            unc = unc_in[s]
            assert torch.equal(unc[0, ts:te], after[s][0, ts:te]), "uncond sequences are built after the text update"
            cond_vq = llada_oracle.head(sd, cfg, x[:, pos], lo, hi).contiguous()
            xu = llada_oracle.forward_hidden(sd, cfg, unc)
            uu = llada_oracle.head(sd, cfg, xu[:, pos], lo, hi).contiguous()
            am, pm, probs = so.image_probs(cond_vq, uu[0:1].contiguous(), uu[1:2].contiguous(), 0.0, 4.0, want_probs=True)
            slots_masked = torch.tensor([int(before[0, p]) == synth.MASK for p in pos])
            gs = sampled[s][0]
            for n in slots_masked.nonzero().flatten().tolist():
                st["img_slots"] += 1
                if int(gs[n]) != int(am[0, n]):
                    st["img_token_diff"] += 1
                    ratio = probs[0, n, int(gs[n])].float().item() / max(pm[0, n].float().item(), 1e-30)
                    st["worst_img_prob_ratio"] = min(st["worst_img_prob_ratio"], ratio)
            # re-mask decision on the oracle's own numbers vs the slots the GPU left masked
            ids_txt = after[s].clone()
            for p in pos:
                ids_txt[0, p] = before[0, p]
            ids_img_o = so.image_commit(ids_txt, pos, am, pm, torch.zeros((1, N), dtype=torch.bfloat16), 0.0, mlen[s])
            gm = torch.tensor([int(after[s][0, p]) == synth.MASK for p in pos])
            om = torch.tensor([int(ids_img_o[0, p]) == synth.MASK for p in pos])
            assert int(gm.sum()) == int(om.sum())
            st["img_mask_diff"] += int((gm ^ om).sum())
    print("config0 teacher-forced, all 32 steps:", st)
    _save("config0_teacher_forced_all_steps", st)
    # Measured in round 2 (profiles/r02_parity.json): 0 of 256 committed text tokens differ; 16.6 % of the sampled image
    # tokens differ (the CFG combine c + 4(c - u) amplifies the 2 % logit noise of two bf16 evaluations five-fold over 8192
    # near-uniform classes) with the oracle's probability of the GPU's token never below 0.76 of its maximum; WHICH text
    # positions / image slots are kept is a rank over near-equal confidences (random weights: every soft-max maximum is
    # ~4e-4) and differs on 10 % of the slots, always within a few per cent of the oracle's own cut.
    assert st["worst_text_margin_sigma"] < 0.1, "a committed text token far from the oracle's arg-max is a kernel bug"
    assert st["worst_img_prob_ratio"] > 0.60, "a sampled image token far from the oracle's most probable one is a bug"
    assert st["worst_text_conf_gap"] < 0.25, "an unmasked text position far from the oracle's confidence cut is a bug"
    assert st["text_token_diff"] <= st["text_committed"] // 20
    assert st["img_token_diff"] <= st["img_slots"] * 3 // 10   # 16.6 % and 19.0 % on two boxes (the oracle's host CPU differs)
    assert st["img_mask_diff"] <= st["img_slots"] // 5


def test_free_running_tiny_trajectory_vs_reference_recording():
    """Free-running (no teacher forcing) generate_ti2ti on the tiny model vs the trajectory the REFERENCE recorded for the
    same weights and job (tests/golden/e2e_tiny.*.npz: every model call's ids).  With random weights many logits are
    near-ties, so id equality is REPORTED (first diverging call, final agreement), and asserted only as far as it has
    been observed to hold."""
    import numpy as np

    from helpers import golden_float, tiny_job, tiny_sd
    from mmada_parallel_amd import LLaDAForMultiModalGeneration, generate_ti2ti
    from helpers import save_parity as _save

    z, _ = golden_float("e2e_tiny", prefer="amx_bf16")
    calls_ref = torch.from_numpy(z["calls"])
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(synth.CFG_TINY), tiny_sd(), device=DEV)
    job = tiny_job()
    calls = []
    real_fb = model.forward_body

    def rec(ids, consumed=None):
        calls.extend(ids[b:b + 1].cpu().clone() for b in range(ids.shape[0]))
        return real_fb(ids, consumed=consumed)

    model.forward_body = rec
    vq, text, final = generate_ti2ti(model, job["input_ids"].to(DEV), job["text_start"], job["text_end"], job["image_start"],
                                     job["seq_len"], job["newline_every"], text_steps=8, timesteps=4, temperature=0.0,
                                     text_temperature=0.0, cfg_scale=0.0, cfg_img=4.0, uncon_text=job["uncon_text"],
                                     uncon_image=job["uncon_image"], return_state=True)
    model.forward_body = real_fb
    got = torch.cat(calls, 0)
    assert got.shape == calls_ref.shape
    same = (got == calls_ref).all(1)
    first_div = int((~same).nonzero()[0]) if not bool(same.all()) else -1
    pos = [i for i in range(job["image_start"], job["image_start"] + job["seq_len"] + job["seq_len"] // job["newline_every"])
           if int(job["input_ids"][0, i]) != synth.NEW_LINE]
    vq_ref, text_ref = z["vq"].tolist(), z["text"].tolist()
    vq_agree = sum(int(final[0, p]) != synth.MASK and vq[j] == vq_ref[j] for j, p in enumerate(pos)) / len(pos)
    text_agree = sum(a == b for a, b in zip(text, text_ref)) / len(text_ref)
    rep = {"model_calls": int(got.shape[0]), "first_diverging_call": first_div, "calls_identical": int(same.sum()),
           "ids_equal_fraction_over_all_calls": (got == calls_ref).float().mean().item(),
           "final_vq_agreement": vq_agree, "final_text_agreement": text_agree}
    # calibration: the reference ARITHMETIC itself (CPU oracle, bit-equal to the reference on the host that recorded the
    # fixture) run free on THIS host's CPU — another CPU's bf16 GEMM blocking is already enough to leave the recording
    from oracle import generate_oracle, llada_oracle

    sd, cfg = tiny_sd(), synth.CFG_TINY
    trace = []
    generate_oracle.generate(lambda x: llada_oracle.forward_logits(sd, cfg, x), job["input_ids"], job["text_start"],
                             job["text_end"], job["image_start"], job["seq_len"], job["newline_every"], 8, 4, 0.0, 4.0,
                             job["uncon_text"], job["uncon_image"], trace=trace)
    ora = torch.cat(trace, 0)
    osame = (ora == calls_ref).all(1)
    rep["oracle_on_this_host"] = {"first_diverging_call": int((~osame).nonzero()[0]) if not bool(osame.all()) else -1,
                                  "ids_equal_fraction_over_all_calls": (ora == calls_ref).float().mean().item(),
                                  "hip_vs_oracle_on_this_host_ids_equal": (got == ora).float().mean().item()}
    print("free-running tiny trajectory vs the reference's recording:", rep)
    _save("free_running_tiny_vs_reference", rep)
    # id equality of a free-running trajectory on random weights is reported, not asserted; the recorded
    # round-2 value is 0.79 of all ids over the 16 calls (the first call is identical by construction)
    assert rep["calls_identical"] >= 1 and rep["ids_equal_fraction_over_all_calls"] > 0.6


def test_graph_replay_bit_identical_at_config2_shapes(block8b):
    """hipGraph replay of the denoise step at BASELINE config-2 shapes (d = 4096, L = 2438, N = 1024, T = 256,
    V = 134656) with all three CFG branches contributing (cfg_scale 3 + cfg_img 4, configs[4]): bit-identical to eager."""
    from mmada_parallel_amd import generate_ti2ti

    _, _, model = block8b
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)

    def run(graph):
        return generate_ti2ti(model, job["input_ids"].to(DEV), job["text_start"], job["text_end"], job["image_start"],
                              job["seq_len"], job["newline_every"], text_steps=12, timesteps=6, temperature=0.0,
                              text_temperature=0.0, cfg_scale=3.0, cfg_img=4.0, uncon_text=job["uncon_text"],
                              uncon_image=job["uncon_image"], return_state=True, graph=graph)[2]

    model.graph_replays, model.graph_nodes = 0, {}
    eager = run(False)
    captured = run(True)
    assert torch.equal(eager, captured)
    assert model.graph_replays >= 12 - 3 and model.graph_nodes
    print("config-2 shapes, 1 block: nodes per captured step kind", model.graph_nodes)
