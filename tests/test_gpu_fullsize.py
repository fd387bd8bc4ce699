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
    assert err.max().item() < 2.0 ** -6 * scale and err.mean().item() < 2.0 ** -9 * scale
    # consumed LM-head rows: text span x full vocabulary, image positions x codebook slab
    ts, te = job["text_start"], job["text_end"]
    rows = torch.arange(ts, ts + 8, dtype=torch.int32, device=DEV)
    lg = model.head_rows(rows, 0, cfg["embedding_size"]).float().cpu()
    lr = llada_oracle.head(sd, cfg, llada_oracle.forward_hidden(sd, cfg, ids)[:, ts:ts + 8])[0].float()
    assert (lg - lr).abs().max().item() < 2.0 ** -5 * lr.abs().max().item()


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

        def head_rows(self, rows, c0, c1):   # low-rank random logits: cheap at [256, 134656]
            return (self._h[rows.long()] @ self._w[c0:c1].t()).to(torch.bfloat16).contiguous()

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
    """BASELINE configs[0] geometry (256x256 output: N = 256, newline every 16, L = 1654, text_steps 32, timesteps 16) on a
    2-block 8B-width model: every conditional call's text argmax / image argmax is compared with the CPU oracle evaluated
    on the SAME ids (teacher forcing: the GPU trajectory supplies the ids).  A differing argmax must be a near-tie."""
    from mmada_parallel_amd import LLaDAForMultiModalGeneration
    from mmada_parallel_amd import generate_ti2ti
    from oracle import llada_oracle

    cfg = dict(synth.CFG_8B, n_layers=2, d_model=1024, n_heads=8, n_kv_heads=8, mlp_hidden_size=2048)
    sd = synth.synthetic_state_dict(cfg, seed=5, device="cpu")
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=DEV, max_batch=2)
    job = synth.synthetic_job(256, 256, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids0 = job["input_ids"]
    assert ids0.shape[1] == 1654 and job["seq_len"] == 256 and job["newline_every"] == 16
    ts, te, N = job["text_start"], job["text_end"], job["seq_len"]
    pos = [i for i in range(job["image_start"], job["image_start"] + N + N // 16) if int(ids0[0, i]) != synth.NEW_LINE]
    snaps = []
    real_fb = model.forward_body

    def recording_fb(ids, consumed=None):
        if ids.shape[0] == 1:
            snaps.append(ids.cpu().clone())  # input of the conditional call of each step
        return real_fb(ids, consumed=consumed)

    model.forward_body = recording_fb
    _, _, final = generate_ti2ti(model, ids0.to(DEV), ts, te, job["image_start"], N, 16, text_steps=32, timesteps=16,
                                 temperature=0.0, text_temperature=0.0, cfg_scale=0.0, cfg_img=4.0,
                                 uncon_text=job["uncon_text"], uncon_image=job["uncon_image"], return_state=True)
    model.forward_body = real_fb
    snaps.append(final)
    assert len(snaps) == 33 and not bool((snaps[-1][0, ts:te] == synth.MASK).any())
    # teacher-forced check on three steps (first, an image step in the middle, the last)
    checked = 0
    for s in (0, 17, 31):
        before, after = snaps[s], snaps[s + 1]
        x = llada_oracle.forward_hidden(sd, cfg, before)
        lt = llada_oracle.head(sd, cfg, x[:, ts:te])[0].float()                       # [T, V]
        changed = (before[0, ts:te] != after[0, ts:te]).nonzero()[:, 0]
        for t in changed.tolist():
            tok = int(after[0, ts + t])
            top2 = lt[t].topk(2).values
            assert tok == int(lt[t].argmax()) or (lt[t, tok] >= top2[0] - 2.0 ** -5 * top2[0].abs().clamp_min(1.0)), \
                f"step {s}: committed text token {tok} is not the oracle's (near-)argmax"
            checked += 1
    assert checked > 0
