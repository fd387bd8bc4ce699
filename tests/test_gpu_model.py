"""GPU parity of the denoiser forward and of the full sampler loop (through the C-ABI).

Three tiers:
  (i)   sampler exactness   — generate_ti2ti driven by STUB logits must reproduce, bit for bit, the ids the
                              reference's generate_ti2ti handed to every model call (tests/golden/sampler_traj.npz);
  (ii)  model tolerance     — residual stream / logits of the HIP forward vs the CPU oracle AND the reference fixture;
  (iii) teacher-forced step — at every step of the reference's recorded tiny-model trajectory, feed the reference's
                              ids, take one step on the GPU and compare decisions; any disagreement must be a
                              near-tie of the oracle's logits (reported with margins).
"""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, ROOT, SAMPLER_CASES, STUB_CB, STUB_TEXT_VOCAB, bits, from_bits, golden_float, stub_logits, tiny_job, tiny_sd
from mmada_parallel_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def tiny_model():
    from mmada_parallel_amd import LLaDAForMultiModalGeneration

    cfg = synth.full_config(synth.CFG_TINY)
    return LLaDAForMultiModalGeneration.from_state_dict(cfg, tiny_sd(), device=DEV)


# ------------------------------------------------------------------------------------------------ (ii) model tolerance
def _relerr(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item(), \
           ((got - ref).abs().mean() / ref.abs().mean().clamp_min(1e-30)).item()


@pytest.mark.parametrize("B", [1, 2])
def test_block_stages_vs_oracle(tiny_model, B):
    """Every intermediate of block 0 (RMSNorm, q/k after RoPE, V, SDPA output, residual adds, SiLU-gated hidden)
    against the oracle's value of the same tensor; errors are relative to each tensor's own magnitude, so a wrong
    branch cannot hide behind the (much larger) residual stream."""
    import torch.nn.functional as F
    from mmada_parallel_amd import abi
    from oracle import llada_oracle as lo

    sd, cfg = tiny_sd(), synth.CFG_TINY
    job = tiny_job()
    ids = job["input_ids"].repeat(B, 1)
    if B > 1:
        ids[1, :8] = torch.arange(100, 108)
    L = ids.shape[1]
    H, Hkv, hd, eps = cfg["n_heads"], cfg["n_kv_heads"], 128, cfg["rms_norm_eps"]
    w = lo.layer_weights(sd, 0)
    x0 = F.embedding(ids, sd["model.transformer.wte.weight"])
    xn = lo.rms_norm(x0, w["attn_norm"], eps)
    q = F.linear(xn, w["q_proj"]).view(B, L, H, hd).transpose(1, 2)
    k = F.linear(xn, w["k_proj"]).view(B, L, Hkv, hd).transpose(1, 2)
    v = F.linear(xn, w["v_proj"]).view(B, L, Hkv, hd).transpose(1, 2)
    sin, cos = lo.rope_tables(L, hd, cfg["rope_theta"])
    q, k = lo.apply_rope(q, sin, cos), lo.apply_rope(k, sin, cos)
    att = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).contiguous().view(B, L, H * hd)
    x1 = x0 + F.linear(att, w["attn_out"])
    n2 = lo.rms_norm(x1, w["ff_norm"], eps)
    hmid = F.silu(F.linear(n2, w["ff_proj"])) * F.linear(n2, w["up_proj"])
    x2 = x1 + F.linear(hmid, w["ff_out"])

    lib, h = tiny_model._lib, tiny_model._handle
    ids_d = ids.to(DEV)
    tiny_model._ensure_ws(B, L)
    tiny_model._shape = (B, L)
    st = abi.stream_ptr()
    abi.check(lib.mmada_embed(h, ids_d.data_ptr(), B, L, st), "embed")
    e_x0 = _relerr(tiny_model.hidden_state(), x0)
    abi.check(lib.mmada_attn_partial(h, 0, st), "attn")
    Lp = (L + 7) // 8 * 8
    got = {
        "xn": tiny_model.debug_buffer(0).view(B, Lp, -1)[:, :L],
        "q": tiny_model.debug_buffer(1)[:, :, :L],
        "k": tiny_model.debug_buffer(2)[:, :, :L],
        "v": tiny_model.debug_buffer(3)[:, :, :, :L].transpose(2, 3),
        "att": tiny_model.debug_buffer(4).view(B, Lp, -1)[:, :L],
        "x1": tiny_model.hidden_state(),
    }
    got = {k_: v_.clone() for k_, v_ in got.items()}
    abi.check(lib.mmada_mlp_partial(h, 0, st), "mlp")
    got["n2"] = tiny_model.debug_buffer(0).view(B, Lp, -1)[:, :L].clone()
    got["h"] = tiny_model.debug_buffer(5).view(B, Lp, -1)[:, :L].clone()
    got["x2"] = tiny_model.hidden_state()
    ref = {"xn": xn, "q": q, "k": k, "v": v, "att": att, "x1": x1, "n2": n2, "h": hmid, "x2": x2}
    report = {"x0": e_x0}
    for name in ref:
        report[name] = _relerr(got[name], ref[name])
    print("stage errors (max/maxabs, mean/meanabs):", {k_: (f"{a:.2e}", f"{b:.2e}") for k_, (a, b) in report.items()})
    assert report["x0"][0] == 0.0
    # the branch deltas, not just the stream: attention and MLP contributions themselves
    report["d_attn"] = _relerr(got["x1"].float().cpu() - x0.float(), x1.float() - x0.float())
    report["d_mlp"] = _relerr(got["x2"].float().cpu() - got["x1"].float().cpu(), x2.float() - x1.float())
    print("branch deltas:", report["d_attn"], report["d_mlp"])
    for name, (emax, emean) in report.items():
        # bf16 storage everywhere: a few ulps (2^-8) of the tensor's magnitude; deltas are differences of bf16 values
        lim_max, lim_mean = (0.25, 0.05) if name.startswith("d_") else (2.0 ** -5, 2.0 ** -8)
        assert emax < lim_max and emean < lim_mean, f"{name}: {emax:.3e} {emean:.3e}"


def test_forward_hidden_and_logits_vs_oracle_and_reference(tiny_model):
    from oracle import llada_oracle

    z, _ = golden_float("forward_tiny", prefer="amx_bf16")
    ids = torch.from_numpy(z["ids"])
    tiny_model.forward_body(ids.to(DEV))
    hid = tiny_model.hidden_state().cpu().float()[0]
    ref_hidden = from_bits(z["hidden"])[-1].float()           # reference, last block (recorded on the build host)
    ora = llada_oracle.forward_hidden(tiny_sd(), synth.CFG_TINY, ids)[0].float()
    # the oracle is bit-equal to the reference on the host that generated the fixture (tests/test_oracle_golden.py);
    # another CPU's bf16 GEMM blocking changes the last bits, so here it is a tolerance check
    o_err = (ora - ref_hidden).abs().max().item() / ref_hidden.abs().max().item()
    print(f"oracle on this host vs reference fixture: rel err {o_err:.3e}")
    assert o_err < 2.0 ** -6
    scale = ref_hidden.abs().max().item()
    err = (hid - ref_hidden).abs()
    print(f"hidden: max|err|={err.max():.4g} mean|err|={err.mean():.4g} scale={scale:.4g}")
    # measured (MI355X, rounds 1-2): max 1.05e-2, mean 6.1e-4 of the stream's magnitude after the two blocks; limits are
    # measured x 1.5 (the numbers of every run are recorded in gpurun_out/r02_parity.json -> profiles/r02_parity.json)
    assert err.max().item() < 1.6e-2 * scale
    assert err.mean().item() < 1.0e-3 * scale

    pos = torch.from_numpy(z["pos"]).to(DEV)
    img = tiny_model.head_rows(pos.int(), synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK).cpu().float()
    img_ref = from_bits(z["img_logits"]).float()
    lscale = img_ref.abs().max().item()
    lerr = (img - img_ref).abs()
    print(f"image logits: max|err|={lerr.max():.4g} mean|err|={lerr.mean():.4g} scale={lscale:.4g}")
    from helpers import save_parity

    save_parity("tiny_forward_vs_reference_fixture", {
        "hidden_max_rel": err.max().item() / scale, "hidden_mean_rel": err.mean().item() / scale,
        "image_logits_max_rel": lerr.max().item() / lscale, "image_logits_mean_rel": lerr.mean().item() / lscale,
        "oracle_on_this_host_vs_fixture_max_rel": o_err})
    # measured: max 5.3e-3, mean 7.8e-4 of the logit magnitude
    assert lerr.max().item() < 1.0e-2 * lscale
    assert lerr.mean().item() < 1.5e-3 * lscale

    job = tiny_job()
    out = tiny_model(ids.to(DEV), infer=True, use_cache=False).logits   # drop-in contract: [B, L, V]
    assert out.shape == (1, ids.shape[1], synth.CFG_TINY["vocab_size"]) and out.dtype == torch.bfloat16
    th = out[0, job["text_start"]:job["text_end"], :4096].cpu().float()
    th_ref = from_bits(z["text_logits_head"]).float()
    assert (th - th_ref).abs().max().item() < 1.5e-2 * th_ref.abs().max().item()
    # argmax agreement with the reference (near-ties may flip: report, require a clear majority)
    agree = (out[0].argmax(-1).cpu().int() == torch.from_numpy(z["argmax"])).float().mean().item()
    print(f"argmax agreement with the reference: {agree:.3f}")
    assert agree > 0.85


def test_batched_forward_equals_single(tiny_model):
    """Stacking equal-length sequences on the batch axis must not change any row."""
    job = tiny_job()
    ids = job["input_ids"].to(DEV)
    other = ids.clone()
    other[0, :5] = torch.tensor([11, 12, 13, 14, 15], device=DEV)
    tiny_model.forward_body(ids)
    a = tiny_model.hidden_state().clone()
    tiny_model.forward_body(other)
    b = tiny_model.hidden_state().clone()
    tiny_model.forward_body(torch.cat([ids, other, ids], 0))
    c = tiny_model.hidden_state()
    assert torch.equal(c[0], a[0]) and torch.equal(c[1], b[0]) and torch.equal(c[2], a[0])


# ----------------------------------------------------------------------------------------------- (i) sampler exactness
def _stubbed(tiny_model, seed, V):
    """Test double: keeps the real handle (sampler kernels) but serves seeded stub logits instead of the forward."""
    from mmada_parallel_amd import LLaDAForMultiModalGeneration

    class Stub(LLaDAForMultiModalGeneration):
        def __init__(self):  # share the native handle of the real model; never destroyed by the stub
            self.__dict__.update({k: v for k, v in tiny_model.__dict__.items()})
            self.vocab = V
            self.calls, self.n = [], 0
            self._cur = None

        def __del__(self):
            pass

        def forward_body(self, ids, consumed=None):
            B = ids.shape[0]
            chunks = []
            for b in range(B):  # the reference calls the model once per sequence: one stub draw per sequence
                self.n += 1
                self.calls.append(ids[b:b + 1].cpu().clone())
                chunks.append(stub_logits(seed, self.n, 1, ids.shape[1], V))
            self._cur = torch.cat(chunks, 0).to(DEV)

        def head_rows(self, rows, c0, c1, out=None):
            flat = self._cur.view(-1, V)
            r = flat[rows.long(), c0:c1].contiguous()
            return r if out is None else out.copy_(r)

    return Stub()


@pytest.mark.parametrize("name", list(SAMPLER_CASES))
def test_generate_stub_trajectory_bit_exact(tiny_model, name):
    from mmada_parallel_amd import generate_ti2ti

    z = np.load(os.path.join(GOLDEN, "sampler_traj.npz"))
    calls_ref = torch.from_numpy(z[name + "_calls"])
    job, kw = tiny_job(), SAMPLER_CASES[name]
    V = STUB_TEXT_VOCAB + STUB_CB
    stub = _stubbed(tiny_model, int(z[name + "_seed"]), V)
    old = tiny_model.config.__dict__.copy()
    try:
        vq, text, final = generate_ti2ti(stub, job["input_ids"].to(DEV), job["text_start"], job["text_end"],
                                         job["image_start"], job["seq_len"], job["newline_every"], temperature=0.0,
                                         text_temperature=0.0, uncon_text=job["uncon_text"], uncon_image=job["uncon_image"],
                                         tokenizer=None, text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB,
                                         return_state=True, **kw)
    finally:
        tiny_model.config.__dict__.update(old)
    got = torch.cat(stub.calls, 0)
    assert got.shape == calls_ref.shape
    assert torch.equal(got, calls_ref), f"first differing model call: {(got != calls_ref).any(1).nonzero()[0].item()}"
    assert text == z[name + "_text"].tolist()
    vq_ref = z[name + "_vq"].tolist()
    pos = [i for i in range(job["image_start"], job["image_start"] + job["seq_len"] + job["seq_len"] // job["newline_every"])
           if int(job["input_ids"][0, i]) != synth.NEW_LINE]
    n_fill = 0
    for j, p in enumerate(pos):
        if int(final[0, p]) == synth.MASK:
            n_fill += 1
        else:
            assert vq[j] == vq_ref[j]
    assert n_fill == 1


@pytest.mark.parametrize("name", ["inpaint_img4", "outpaint_both"])
def test_generate_painting_mode_trajectory_bit_exact(tiny_model, name):
    """Painting mode: the output image span starts partly known (in- / out-painting rectangle): every model call's ids must
    equal the reference's recording, known cells are never touched (tests/golden/paint_traj.npz)."""
    from helpers import PAINT_CASES, paint_job
    from mmada_parallel_amd import generate_ti2ti

    z = np.load(os.path.join(GOLDEN, "paint_traj.npz"))
    calls_ref = torch.from_numpy(z[name + "_calls"])
    kind, kw = PAINT_CASES[name]
    job = paint_job(kind)
    V = STUB_TEXT_VOCAB + STUB_CB
    stub = _stubbed(tiny_model, int(z[name + "_seed"]), V)
    old = tiny_model.config.__dict__.copy()
    try:
        vq, text, final = generate_ti2ti(stub, job["input_ids"].to(DEV), job["text_start"], job["text_end"],
                                         job["image_start"], job["seq_len"], job["newline_every"], temperature=0.0,
                                         text_temperature=0.0, uncon_text=job["uncon_text"], uncon_image=job["uncon_image"],
                                         tokenizer=None, text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB,
                                         return_state=True, **kw)
    finally:
        tiny_model.config.__dict__.update(old)
    got = torch.cat(stub.calls, 0)
    assert got.shape == calls_ref.shape
    assert torch.equal(got, calls_ref), f"first differing model call: {(got != calls_ref).any(1).nonzero()[0].item()}"
    assert text == z[name + "_text"].tolist()
    pos = [i for i in range(job["image_start"], job["image_start"] + job["seq_len"] + job["seq_len"] // job["newline_every"])
           if int(job["input_ids"][0, i]) != synth.NEW_LINE]
    for j, p in enumerate(pos):
        if int(job["input_ids"][0, p]) != synth.MASK:
            assert int(final[0, p]) == int(job["input_ids"][0, p])
        if int(final[0, p]) != synth.MASK:
            assert vq[j] == int(z[name + "_vq"][j])


@pytest.mark.parametrize("name", ["rand_img4", "rand_both"])
def test_generate_random_remasking_trajectory_bit_exact(tiny_model, name):
    """remasking='random': with the reference's draws replayed from the same seeded GLOBAL CPU generator (uniform ranks of
    the text positions, the re-mask jitter's randn), every model call's ids equal the reference's recording."""
    from helpers import RANDOM_CASES, RANDOM_SEED, ReplayCpuRng
    from mmada_parallel_amd import generate_ti2ti

    z = np.load(os.path.join(GOLDEN, "random_traj.npz"))
    calls_ref = torch.from_numpy(z[name + "_calls"])
    job, kw = tiny_job(), RANDOM_CASES[name]
    V = STUB_TEXT_VOCAB + STUB_CB
    stub = _stubbed(tiny_model, int(z[name + "_seed"]), V)
    old = tiny_model.config.__dict__.copy()
    torch.manual_seed(RANDOM_SEED)
    try:
        vq, text, final = generate_ti2ti(stub, job["input_ids"].to(DEV), job["text_start"], job["text_end"],
                                         job["image_start"], job["seq_len"], job["newline_every"],
                                         uncon_text=job["uncon_text"], uncon_image=job["uncon_image"], tokenizer=None,
                                         remasking="random", text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB,
                                         return_state=True, rng=ReplayCpuRng(), **kw)
    finally:
        tiny_model.config.__dict__.update(old)
    got = torch.cat(stub.calls, 0)
    assert got.shape == calls_ref.shape
    assert torch.equal(got, calls_ref), f"first differing model call: {(got != calls_ref).any(1).nonzero()[0].item()}"
    assert text == z[name + "_text"].tolist()
    with pytest.raises(RuntimeError):  # the reference raises too with an explicit generator (torch.rand(dtype=int64))
        generate_ti2ti(stub, job["input_ids"].to(DEV), job["text_start"], job["text_end"], job["image_start"],
                       job["seq_len"], job["newline_every"], remasking="random", generator=torch.Generator(device=DEV),
                       text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB, **kw)


@pytest.mark.parametrize("name", ["cfg_without_uncond", "only_uncon_image", "text_done", "more_timesteps_than_steps", "single_step"])
def test_generate_edge_case_trajectory_bit_exact(tiny_model, name):
    """Edge cases of the loop (tests/helpers.py EDGE_CASES) against the reference's recordings: CFG scales without
    unconditional prompts (zero logits stand in), one prompt only, a complete text span, more image steps than steps, a
    single step."""
    from helpers import edge_job
    from mmada_parallel_amd import generate_ti2ti

    z = np.load(os.path.join(GOLDEN, "edge_traj.npz"))
    calls_ref = torch.from_numpy(z[name + "_calls"])
    job, kw = edge_job(name)
    V = STUB_TEXT_VOCAB + STUB_CB
    stub = _stubbed(tiny_model, int(z[name + "_seed"]), V)
    old = tiny_model.config.__dict__.copy()
    try:
        vq, text, final = generate_ti2ti(stub, job["input_ids"].to(DEV), job["text_start"], job["text_end"],
                                         job["image_start"], job["seq_len"], job["newline_every"], temperature=0.0,
                                         text_temperature=0.0, uncon_text=job["uncon_text"], uncon_image=job["uncon_image"],
                                         tokenizer=None, text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB,
                                         return_state=True, **kw)
    finally:
        tiny_model.config.__dict__.update(old)
    got = torch.cat(stub.calls, 0)
    assert got.shape == calls_ref.shape, f"{got.shape[0]} model calls, the reference made {calls_ref.shape[0]}"
    assert torch.equal(got, calls_ref), f"first differing model call: {(got != calls_ref).any(1).nonzero()[0].item()}"
    assert text == z[name + "_text"].tolist()
    pos = [i for i in range(job["image_start"], job["image_start"] + job["seq_len"] + job["seq_len"] // job["newline_every"])
           if int(job["input_ids"][0, i]) != synth.NEW_LINE]
    for j, p in enumerate(pos):
        if int(final[0, p]) != synth.MASK:
            assert vq[j] == int(z[name + "_vq"][j])


# --------------------------------------------------------------------------------------------- (iii) teacher-forced e2e
def test_teacher_forced_tiny_trajectory(tiny_model):
    """Feed the reference's recorded ids of every conditional call; compare the GPU's per-position decisions
    (text argmax token + fp64 confidence, image argmax over the CFG-free codebook logits) with the oracle evaluated on
    the same ids.  A differing argmax must be a near-tie of the oracle's logits; confidences must agree closely.
    (Which POSITIONS get unmasked is a pure function of these confidences and is checked bit-exactly, on identical
    inputs, by tests/test_gpu_kernels.py and the stub-trajectory test above.)"""
    from mmada_parallel_amd import abi
    from oracle import llada_oracle
    from oracle import sampler_oracle as so

    z, _ = golden_float("e2e_tiny", prefer="amx_bf16")
    calls = torch.from_numpy(z["calls"])          # per step: cond, then (uncond_text, uncond_img) on image steps
    job = tiny_job()
    ts, te = job["text_start"], job["text_end"]
    T = te - ts
    text_steps, timesteps = 8, 4
    img_steps = set(torch.linspace(text_steps // 4, text_steps - 1, timesteps).round().int().tolist())
    pos = [i for i in range(job["image_start"], job["image_start"] + job["seq_len"] + job["seq_len"] // job["newline_every"])
           if int(job["input_ids"][0, i]) != synth.NEW_LINE]
    lib, h = tiny_model._lib, tiny_model._handle
    sd, cfg = tiny_sd(), synth.CFG_TINY
    V = tiny_model.vocab
    ci = 0
    n_tok = n_tok_bad = n_img = n_img_bad = 0
    worst_margin, worst_conf = 0.0, 0.0
    for step in range(text_steps):
        ids = calls[ci:ci + 1].clone()
        ci += 3 if step in img_steps else 1
        ids_dev = ids.to(DEV)
        tiny_model.forward_body(ids_dev)
        rows = torch.arange(ts, te, dtype=torch.int32, device=DEV)
        tl = tiny_model.head_rows(rows, 0, V)
        k_dev = torch.zeros(1, dtype=torch.int32, device=DEV)   # k = 0: statistics only, ids untouched
        scratch = torch.empty(T * 16, dtype=torch.uint8, device=DEV)
        abi.check(lib.mmada_text_select(h, tl.data_ptr(), None, 1, T, V, V, ids_dev.data_ptr(), ids.shape[1], ts,
                                        k_dev.data_ptr(), scratch.data_ptr(), abi.stream_ptr()), "text")
        conf = scratch[: T * 8].view(torch.float64).cpu()
        x0 = scratch[T * 8: T * 12].view(torch.int32).cpu()
        ol = llada_oracle.head(sd, cfg, llada_oracle.forward_hidden(sd, cfg, ids)[:, ts:te])   # [1,T,V] bf16
        _, conf_o, x0_o = so.text_select(ol, None, ids, ts, [0])
        masked = torch.isfinite(conf_o[0])
        assert torch.equal(torch.isfinite(conf), masked)
        olf = ol[0].float()
        for t in masked.nonzero().flatten().tolist():
            n_tok += 1
            if int(x0[t]) != int(x0_o[0, t]):
                n_tok_bad += 1
                top2 = olf[t].topk(2).values
                worst_margin = max(worst_margin, ((top2[0] - top2[1]) / olf[t].std()).item())
            else:
                worst_conf = max(worst_conf, abs(conf[t].item() / conf_o[0, t].item() - 1.0))
        if step in img_steps:
            irows = torch.tensor(pos, dtype=torch.int32, device=DEV)
            il = tiny_model.head_rows(irows, synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK).cpu().float()
            oil = llada_oracle.head(sd, cfg, llada_oracle.forward_hidden(sd, cfg, ids)[:, pos], synth.TEXT_VOCAB,
                                    synth.TEXT_VOCAB + synth.CODEBOOK)[0].float()
            for n in range(len(pos)):
                n_img += 1
                if int(il[n].argmax()) != int(oil[n].argmax()):
                    n_img_bad += 1
                    top2 = oil[n].topk(2).values
                    worst_margin = max(worst_margin, ((top2[0] - top2[1]) / oil[n].std()).item())
    print(f"teacher-forced: text argmax {n_tok_bad}/{n_tok} differ, image argmax {n_img_bad}/{n_img} differ, "
          f"worst oracle top1-top2 margin at a disagreement = {worst_margin:.4f} sigma, worst confidence rel diff = {worst_conf:.3e}")
    assert worst_margin < 0.05, "a disagreement with a clear oracle margin is a kernel bug, not a near-tie"
    assert worst_conf < 0.05
    assert n_tok_bad <= n_tok // 4 and n_img_bad <= n_img // 4


# ------------------------------------------------------------------------------------------------- tensor parallel
@pytest.mark.parametrize("tp", [2, 4, 8])
def test_tensor_parallel_slices_on_one_gpu(tiny_model, tp):
    """TP=2/4/8 emulated on one GPU: one handle per tp_rank from the same checkpoint, the all-reduce replaced by an
    explicit sum of their partial residual streams.  Must agree with the TP=1 forward up to bf16 re-association."""
    from mmada_parallel_amd import LLaDAForMultiModalGeneration, abi

    if tp == 2:
        base, sd_, ref_model = synth.CFG_TINY, tiny_sd(), tiny_model
    else:  # 8 heads so that every rank owns at least one
        base = dict(synth.CFG_TINY, d_model=1024, n_heads=8, n_kv_heads=8, mlp_hidden_size=2048)
        sd_ = synth.synthetic_state_dict(base, seed=5)
        ref_model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(base), sd_, device=DEV)
    cfg = synth.full_config(base)
    ranks = [LLaDAForMultiModalGeneration.from_state_dict(cfg, sd_, device=DEV, tp_rank=r, tp_size=tp)
             for r in range(tp)]
    job = tiny_job()
    ids = job["input_ids"].repeat(2, 1).to(DEV)
    ids[1, :6] = torch.arange(50, 56, device=DEV)
    B, L = ids.shape
    st = abi.stream_ptr()
    for m in ranks:
        m._ensure_ws(B, L)
        m._shape = (B, L)
        abi.check(m._lib.mmada_embed(m._handle, ids.data_ptr(), B, L, st), "embed")
    for layer in range(cfg["n_layers"]):
        for seg in ("mmada_attn_partial", "mmada_mlp_partial"):
            for m in ranks:
                abi.check(getattr(m._lib, seg)(m._handle, layer, st), seg)
            views = [m._stream_view() for m in ranks]
            total = views[0].float()
            for v in views[1:]:
                total = total + v.float()
            total = total.to(torch.bfloat16)
            for v in views:
                v.copy_(total)
    got = ranks[0].hidden_state().float().cpu()
    ref_model.forward_body(ids)
    ref = ref_model.hidden_state().float().cpu()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print(f"TP={tp} vs TP=1 hidden rel err {err:.3e}")
    assert err < 2.0 ** -5   # a few bf16 ulps of the stream: the two partial sums are rounded before they are added
    rows = torch.arange(B * L, dtype=torch.int32, device=DEV)
    lg = ranks[0].head_rows(rows, synth.TEXT_VOCAB, synth.TEXT_VOCAB + 512).float().cpu()
    lr = ref_model.head_rows(rows, synth.TEXT_VOCAB, synth.TEXT_VOCAB + 512).float().cpu()
    assert (lg - lr).abs().max().item() < 2.0 ** -5 * lr.abs().max().item()
    if tp != 2:
        ref_model.forward_body(ids[:1])


def test_microbatched_overlap_path_matches_single_context(tiny_model, monkeypatch):
    """The tensor-parallel overlap schedule (two activation contexts over shared weights, async all-reduce of one
    micro-batch under the other's kernels) exercised on one GPU with a single-rank RCCL group: results must be
    bit-identical to the plain path, for forward_body / head_rows and for a whole generate_ti2ti run."""
    import torch.distributed as dist

    from mmada_parallel_amd import generate_ti2ti

    job = tiny_job()
    ids = job["input_ids"].repeat(3, 1).to(DEV)
    ids[1, :4] = torch.tensor([5, 6, 7, 8], device=DEV)
    ids[2, :4] = torch.tensor([9, 10, 11, 12], device=DEV)
    rows = (torch.arange(3, device=DEV)[:, None] * ids.shape[1] + torch.arange(10, 40, device=DEV)[None, :]).reshape(-1).int()
    tiny_model.forward_body(ids)
    ref_h = tiny_model.hidden_state().clone()
    ref_l = tiny_model.head_rows(rows, 100, 612).clone()
    kw = dict(text_steps=8, timesteps=4, temperature=0.0, text_temperature=0.0, cfg_scale=2.5, cfg_img=4.0,
              uncon_text=job["uncon_text"], uncon_image=job["uncon_image"], return_state=True)
    args = (job["text_start"], job["text_end"], job["image_start"], job["seq_len"], job["newline_every"])
    ref_final = generate_ti2ti(tiny_model, ids, *args, **kw)[2]

    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        monkeypatch.setenv("MMADA_MICROBATCH", "1")
        tiny_model.forward_body(ids)
        assert tiny_model._split == 2
        assert torch.equal(tiny_model.hidden_state(), ref_h)
        assert torch.equal(tiny_model.head_rows(rows, 100, 612), ref_l)
        got_final = generate_ti2ti(tiny_model, ids, *args, **kw)[2]
        assert torch.equal(got_final, ref_final)
    finally:
        monkeypatch.delenv("MMADA_MICROBATCH", raising=False)
        tiny_model.forward_body(ids[:1])  # back to the single-context state for the other tests
        if created:
            dist.destroy_process_group()


# ------------------------------------------------------------------------------------- M variant: interleave_generate
from helpers import M_CASES, M_SHAPE  # noqa: E402


@pytest.mark.parametrize("name", list(M_CASES))
def test_m_interleave_stub_trajectory_bit_exact(tiny_model, name):
    """interleave_generate on stub logits with the reference's (per-call seeded) random draws replayed: the ids of
    every batch-2 forward and the returned image / text ids must equal the reference's (tests/golden/m_traj.npz)."""
    from types import SimpleNamespace

    from mmada_parallel_amd import interleave_generate
    from oracle.interleave_oracle import SeededRng

    z = np.load(os.path.join(GOLDEN, "m_traj.npz"))
    sh, kw = M_SHAPE, dict(M_CASES[name])
    seed = int(z[name + "_seed"])
    V = sh["text_vocab"] + sh["CB"]
    stub = _stubbed(tiny_model, seed, V)
    calls = []

    def fb(ids, consumed=None):  # the M sampler runs cond+uncond as ONE batch-2 forward: one stub draw per forward
        stub.n += 1
        calls.append(ids.cpu().clone())
        stub._cur = stub_logits(seed, stub.n, ids.shape[0], ids.shape[1], V).to(DEV)

    stub.forward_body = fb

    class Tok:
        bos_token_id = sh["bos"]

        def __len__(self):
            return sh["text_vocab"]

    cfgobj = SimpleNamespace(model=SimpleNamespace(mmada=SimpleNamespace(num_vq_tokens=sh["N"], codebook_size=sh["CB"])),
                             dataset=SimpleNamespace(preprocessing=SimpleNamespace(max_seq_length=sh["T"])))
    img, text = interleave_generate(stub, torch.from_numpy(z[name + "_inp"]), torch.from_numpy(z[name + "_unc"]),
                                    reserved_token_mapping={"<|soi|>": sh["soi"], "<|eoi|>": sh["eoi"]}, config=cfgobj,
                                    uni_prompting=SimpleNamespace(text_tokenizer=Tok()), rng=SeededRng(seed), **kw)
    got = torch.stack(calls, 0)
    ref = torch.from_numpy(z[name + "_calls"])
    assert got.shape == ref.shape and torch.equal(got, ref)
    assert torch.equal(img.cpu(), torch.from_numpy(z[name + "_img"]))
    assert torch.equal(text.cpu(), torch.from_numpy(z[name + "_text"]))


def test_from_pretrained_reads_reference_checkpoint_layout(tmp_path, tiny_model):
    """config.json + *.safetensors with the reference's state-dict keys load unchanged."""
    import json

    from safetensors.torch import save_file

    from mmada_parallel_amd import LLaDAForMultiModalGeneration

    sd = tiny_sd()
    keys = sorted(sd)
    half = len(keys) // 2   # two shards, like a sharded HF checkpoint
    save_file({k: sd[k].contiguous() for k in keys[:half]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({k: sd[k].contiguous() for k in keys[half:]}, str(tmp_path / "model-00002-of-00002.safetensors"))
    cfg = {k: v for k, v in synth.full_config(synth.CFG_TINY).items() if isinstance(v, (int, float, str, bool, type(None)))}
    with open(tmp_path / "config.json", "w") as f:
        json.dump(cfg, f)
    m = LLaDAForMultiModalGeneration.from_pretrained(str(tmp_path), torch_dtype=torch.bfloat16, device_map="auto")
    assert m.config.d_model == 256 and m.device.type == "cuda"
    ids = tiny_job()["input_ids"].to(DEV)
    m.forward_body(ids)
    tiny_model.forward_body(ids)
    assert torch.equal(m.hidden_state(), tiny_model.hidden_state())
    assert getattr(m.config, "text_vocab_size", 126356) == 126356


# ----------------------------------------------------------------- Gradio sampler (app.py:143-398), token level
from helpers import STEPWISE_CASES  # noqa: E402


@pytest.mark.parametrize("name", list(STEPWISE_CASES))
def test_stepwise_generator_bit_exact(tiny_model, name):
    from mmada_parallel_amd import generate_ti2ti_stepwise

    z = np.load(os.path.join(GOLDEN, "stepwise_traj.npz"))
    job, kw = tiny_job(), STEPWISE_CASES[name]
    V = STUB_TEXT_VOCAB + STUB_CB
    stub = _stubbed(tiny_model, int(z[name + "_seed"]), V)
    shown = [0]
    for step, ids, sampled, show in generate_ti2ti_stepwise(
            stub, job["input_ids"].to(DEV), job["text_start"], job["text_end"], job["image_start"], job["seq_len"],
            job["newline_every"], temperature=0.0, text_temperature=0.0, uncon_text=job["uncon_text"],
            uncon_image=job["uncon_image"], text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB, **kw):
        assert ids.shape == job["input_ids"].shape
        if show:
            shown.append(step)
        if sampled is not None:
            assert sampled.shape == (1, job["seq_len"]) and int(sampled.min()) >= 0 and int(sampled.max()) < STUB_CB
    assert torch.equal(torch.cat(stub.calls, 0), torch.from_numpy(z[name + "_calls"]))
    assert shown == z[name + "_yields"].tolist()   # the reference's display cadence


# ------------------------------------------------------------------- generate_image (A text-to-image MaskGIT sampler)
from helpers import T2I_CASES, ReplayRng, t2i_job  # noqa: E402


@pytest.mark.parametrize("name", list(T2I_CASES))
def test_generate_image_stub_trajectory_bit_exact(tiny_model, name):
    """generate_image on stub logits: the ids of EVERY model call (cond and, with CFG, the shorter/longer uncond
    sequence) and the returned vq ids equal the reference's own generate_image (tests/golden/t2i_traj.npz); at
    temperature 1 the reference's bf16 uniform draws are replayed from the same seeded CPU generator."""
    from mmada_parallel_amd.generators.image_generation_generator import generate_image

    z = np.load(os.path.join(GOLDEN, "t2i_traj.npz"))
    job, kw = t2i_job(), T2I_CASES[name]
    V = STUB_TEXT_VOCAB + STUB_CB
    stub = _stubbed(tiny_model, int(z[name + "_seed"]), V)
    gen = torch.Generator().manual_seed(int(z[name + "_gen_seed"])) if kw["temperature"] > 0 else None
    vq = generate_image(stub, job["prompt"], seq_len=job["seq_len"], newline_every=job["newline_every"],
                        uncon_ids=job["uncon_ids"], code_start=job["code_start"], codebook_size=STUB_CB,
                        text_vocab_size=STUB_TEXT_VOCAB, generator=gen, rng=ReplayRng(), **kw)
    lens = z[name + "_calls_len"]
    assert len(stub.calls) == len(lens)
    for i, n in enumerate(lens):
        ref = torch.from_numpy(z[name + "_calls"][i, :n]).view(1, -1)
        assert torch.equal(stub.calls[i], ref), f"model call {i} differs"
    assert torch.equal(vq.cpu(), torch.from_numpy(z[name + "_vq"]))


def test_generate_image_real_tiny_model_runs_and_is_deterministic(tiny_model):
    from mmada_parallel_amd import generate_image

    job = t2i_job(side=8)
    kw = dict(seq_len=job["seq_len"], newline_every=job["newline_every"], uncon_ids=job["uncon_ids"],
              code_start=job["code_start"], timesteps=5, temperature=0.0, cfg_scale=2.0)
    a = generate_image(tiny_model, job["prompt"], **kw)
    b = generate_image(tiny_model, job["prompt"], **kw)
    assert a.shape == (1, 64) and torch.equal(a, b)
    assert int(a.min()) >= synth.TEXT_VOCAB and int(a.max()) < synth.TEXT_VOCAB + synth.CODEBOOK  # every slot was filled
    with pytest.raises(TypeError):
        generate_image(object(), job["prompt"], **kw)
    # use_cache=True: the reference switches its dLLM cache bookkeeping on (model.caching / empty_cache,
    # generators/image_generation_generator.py:65-68,105-108) but never hands the model a compute mask (:128,141), so the
    # arithmetic is unchanged — same tokens here, and the model class has the methods the reference's own loop calls
    c = generate_image(tiny_model, job["prompt"], use_cache=True, **kw)
    assert torch.equal(a, c)
    tiny_model.caching(True)
    ids = job["prompt"].to(DEV)
    l0 = tiny_model(ids, infer=True, use_cache=False).logits
    l1 = tiny_model(ids, infer=True, use_cache=True).logits
    tiny_model.empty_cache()
    tiny_model.caching(False)
    assert torch.equal(l0, l1)


# ------------------------------------------------------------------------------- mmu_generate (M block-wise text sampler)
from helpers import MMU_CASES, MMU_SHAPE  # noqa: E402


@pytest.mark.parametrize("name", list(MMU_CASES))
def test_mmu_generate_stub_trajectory_bit_exact(tiny_model, name):
    """mmu_generate on stub logits (B = 2; CFG runs cond and prompt-masked copies as one batch-4 forward): ids of every
    forward and the returned sequence equal the reference's MMadaModelLM.mmu_generate (tests/golden/mmu_traj.npz); the
    float64 Gumbel noise of the temperature case is replayed from the same per-call seeded generators."""
    from mmada_parallel_amd.generators.mmu_generator import mmu_generate
    from oracle.interleave_oracle import SeededRng

    z = np.load(os.path.join(GOLDEN, "mmu_traj.npz"))
    seed, kw, sh = int(z[name + "_seed"]), MMU_CASES[name], MMU_SHAPE
    stub = _stubbed(tiny_model, seed, sh["V"])

    def fb(ids, consumed=None):  # one stub draw per forward, whatever its batch size (like the reference's single model call)
        stub.n += 1
        stub.calls.append(ids.cpu().clone())
        stub._cur = stub_logits(seed, stub.n, ids.shape[0], ids.shape[1], sh["V"]).to(DEV)

    stub.forward_body = fb
    x = mmu_generate(stub, torch.from_numpy(z[name + "_idx"]), mask_id=sh["mask_id"], rng=SeededRng(seed), **kw)
    assert torch.equal(torch.stack(stub.calls, 0), torch.from_numpy(z[name + "_calls"]))
    assert torch.equal(x.cpu(), torch.from_numpy(z[name + "_x"]))


def test_mmu_generate_real_tiny_model(tiny_model):
    from mmada_parallel_amd.generators.mmu_generator import mmu_generate, mmu_generate_fast

    idx = torch.randint(0, 1000, (2, 9), generator=torch.Generator().manual_seed(2))
    a = mmu_generate(tiny_model, idx, max_new_tokens=16, steps=8, block_length=8)
    b = mmu_generate(tiny_model, idx, max_new_tokens=16, steps=8, block_length=8)
    assert a.shape == (2, 25) and torch.equal(a, b) and not bool((a == synth.MASK).any())
    assert torch.equal(a[:, :9].cpu(), idx)
    c = mmu_generate_fast(tiny_model, idx, max_new_tokens=16, steps=8, block_length=8, eot_token=int(a[0, 16]))
    assert c.shape == (2, 25)
    # the reference's attention_bias is dead code in its model (unmasked attention whatever the mask): same ids
    d = mmu_generate(tiny_model, idx, max_new_tokens=16, steps=8, block_length=8,
                     attention_mask=torch.tensor([[0] + [1] * 8, [1] * 9]))
    assert torch.equal(a, d)


# ---------------------------------------------------------------------- t2i_generate (M MaskGIT text-to-image sampler)
from helpers import M_T2I_CASES, M_T2I_SHAPE, m_t2i_job  # noqa: E402


@pytest.mark.parametrize("name", list(M_T2I_CASES))
def test_m_t2i_generate_stub_trajectory_bit_exact(tiny_model, name):
    """t2i_generate on stub logits with the reference's multinomial / uniform draws replayed: ids of every forward
    (cond + uncond as one batch), the returned codebook ids and the in-place updated input_ids equal the reference's
    MMadaModelLM.t2i_generate (tests/golden/m_t2i_traj.npz), incl. B = 2 and pre-filled (known) image tokens."""
    from types import SimpleNamespace

    from mmada_parallel_amd.generators.t2i_generator import t2i_generate
    from oracle.interleave_oracle import SeededRng

    z = np.load(os.path.join(GOLDEN, "m_t2i_traj.npz"))
    seed, kw, sh = int(z[name + "_seed"]), M_T2I_CASES[name], M_T2I_SHAPE
    V = sh["text_vocab"] + sh["CB"]
    inp, unc = m_t2i_job(seed, kw["B"], kw["known"])
    stub = _stubbed(tiny_model, seed, V)

    def fb(ids, consumed=None):
        stub.n += 1
        stub.calls.append(ids.cpu().clone())
        stub._cur = stub_logits(seed, stub.n, ids.shape[0], ids.shape[1], V).to(DEV)

    stub.forward_body = fb

    class Tok:
        def __len__(self):
            return sh["text_vocab"]

    ids = t2i_generate(stub, input_ids=inp, uncond_input_ids=unc if kw["uncond"] else None, temperature=kw["temperature"],
                       timesteps=kw["timesteps"], guidance_scale=kw["guidance_scale"], seq_len=sh["N"],
                       mask_token_id=sh["mask_id"], resolution=sh["resolution"], codebook_size=sh["CB"],
                       uni_prompting=SimpleNamespace(text_tokenizer=Tok()), rng=SeededRng(seed))
    assert torch.equal(torch.stack(stub.calls, 0), torch.from_numpy(z[name + "_calls"]))
    assert torch.equal(ids.cpu(), torch.from_numpy(z[name + "_ids"]))
    assert torch.equal(inp, torch.from_numpy(z[name + "_final_input"]))


# ------------------------------------------------------------------ multi-process tensor parallel, end to end (one GPU)
@pytest.mark.parametrize("world,plain", [(2, True), (4, False), (8, True)])
def test_bench_multi_rank_tensor_parallel_on_one_gpu(world, plain):
    """bench.py's N-rank path (one process per rank, TP = N, N jobs) launched as the driver launches it — as the PLAIN command
    `python bench.py --gpus N ...` (bench.py starts its own ranks under torch.distributed.run) or under torch.distributed.run
    directly — except that all ranks share cuda:0 and the control plane runs over gloo (MMADA_BENCH_ONE_GPU=1): every rank
    must finish, sample identical tokens (tp_ranks_agree) and the ONE JSON line must be the last line of stdout."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, MMADA_BENCH_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    tail = ["--gpus", str(world), "--steps", "1", "--warmup", "0", "--layers", "2", "--text-steps", "8", "--timesteps", "4",
            "--no-cpu-baseline"]
    if plain:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
               "127.0.0.1", "--master-port", str(29600 + world), os.path.join(ROOT, "bench.py")] + tail
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    if plain:
        assert p.stdout.strip().splitlines()[-1] == lines[0]   # the line is the LAST line of stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["config"]["parallelism"] == f"tp{world}" and out["config"]["global_batch"] == world
    assert out["config"]["tp_ranks_agree"] is True
    assert out["value"] > 0 and "REDUCED" in out["config"]["workload"]
    assert ("self-launch" in out["config"]["launched_by"]) == plain
    # the exchange ran inside the library over hipIpc-mapped peer buffers (gloo only carried the handles), passed its
    # self-test on every rank, and no hand-off timed out
    assert out["config"]["tp_collective"] == "pull", out["config"]["tp_collective"]
    probe = out["config"]["allreduce_probe"]
    assert probe and probe["status"]["error"] == 0 and probe["status"]["mode"] == "pull"
    # one rank per device is what RCCL needs: on the one-GPU rig no RCCL communicator exists, and the line says so
    assert out["config"]["rccl_nranks"] == 0 and out["config"]["library_rccl_nranks"] == 0
    ex = out["config"]["exchange_exposure_probe"]
    assert ex and ex["forward_ms_with_exchange"] > 0 and ex["forward_ms_no_exchange_diagnostic"] > 0
    assert out["config"]["exposed_exchange_ms_per_forward"] == ex["exposed_exchange_ms_per_forward"]


# ------------------------------------------------------------------------------- consumed-row window of the last block
@pytest.mark.parametrize("B,win", [(1, (37, 60)), (2, (33, 70)), (1, (64, 70)), (2, (5, 40))])
def test_consumed_row_window_is_bit_identical_on_the_consumed_rows(tiny_model, B, win):
    """forward_body(consumed=(lo, hi)): the last block skips the other rows; logits of the consumed rows must equal the
    full forward's BIT FOR BIT (same K order in every GEMM, same 32-query wave grouping in attention)."""
    from mmada_parallel_amd import abi

    job = tiny_job()
    ids = job["input_ids"].repeat(B, 1).to(DEV)
    if B == 2:
        ids[1, :6] = torch.arange(50, 56, device=DEV)
    L = ids.shape[1]
    lo, hi = win
    assert hi <= L
    rows = (torch.arange(B, device=DEV)[:, None] * L + torch.arange(lo, hi, device=DEV)[None, :]).reshape(-1).to(torch.int32)
    tiny_model.forward_body(ids)
    full = tiny_model.head_rows(rows, synth.TEXT_VOCAB - 64, synth.TEXT_VOCAB + 448).clone()
    os.environ["MMADA_CHECK_ROWS"] = "1"
    try:
        tiny_model.forward_body(ids, consumed=(lo, hi))
        got = tiny_model.head_rows(rows, synth.TEXT_VOCAB - 64, synth.TEXT_VOCAB + 448)
        assert torch.equal(got, full)
        with pytest.raises(abi.MmadaError):
            tiny_model.hidden_state()  # the resident stream only holds the window
        with pytest.raises(AssertionError):
            tiny_model.head_rows(torch.tensor([max(0, (lo & ~31) - 1)], dtype=torch.int32, device=DEV) if lo >= 32 else
                                 torch.tensor([hi], dtype=torch.int32, device=DEV), 0, 64)
    finally:
        os.environ.pop("MMADA_CHECK_ROWS", None)
    tiny_model.forward_body(ids)  # the window is cleared by a plain forward
    assert torch.equal(tiny_model.head_rows(rows, synth.TEXT_VOCAB - 64, synth.TEXT_VOCAB + 448), full)
    tiny_model.hidden_state()


def test_generate_ti2ti_identical_with_and_without_row_window(tiny_model):
    from mmada_parallel_amd import generate_ti2ti

    job = tiny_job()

    def run():
        return generate_ti2ti(tiny_model, job["input_ids"].to(DEV), job["text_start"], job["text_end"], job["image_start"],
                              job["seq_len"], job["newline_every"], text_steps=8, timesteps=4, temperature=0.0,
                              text_temperature=0.0, cfg_scale=2.0, cfg_img=4.0, uncon_text=job["uncon_text"],
                              uncon_image=job["uncon_image"], return_state=True)

    a = run()
    os.environ["MMADA_NO_WINDOW"] = "1"
    try:
        b = run()
    finally:
        os.environ.pop("MMADA_NO_WINDOW", None)
    assert torch.equal(a[2], b[2]) and a[1] == b[1]


# ------------------------------------------------------------------ generate_ti2ti at temperature > 0 (README defaults)
from helpers import NOISY_CASES, ReplayCpuRng  # noqa: E402


@pytest.mark.parametrize("name", list(NOISY_CASES))
def test_generate_noisy_stub_trajectory_bit_exact(tiny_model, name):
    """temperature 1.0 / text_temperature 0.7 (the reference README's defaults): with the reference's torch.rand /
    torch.multinomial / torch.randn draws replayed from the same seeded CPU generator, the ids of every model call equal
    the reference's recorded run (tests/golden/sampler_noisy.npz) — Gumbel-noised text argmax, multinomial image tokens
    and the jittered re-mask all included."""
    from mmada_parallel_amd import generate_ti2ti

    z = np.load(os.path.join(GOLDEN, "sampler_noisy.npz"))
    calls_ref = torch.from_numpy(z[name + "_calls"])
    job, kw = tiny_job(), NOISY_CASES[name]
    V = STUB_TEXT_VOCAB + STUB_CB
    stub = _stubbed(tiny_model, int(z[name + "_seed"]), V)
    gen = torch.Generator().manual_seed(int(z[name + "_gen_seed"]))
    generate_ti2ti(stub, job["input_ids"].to(DEV), job["text_start"], job["text_end"], job["image_start"], job["seq_len"],
                   job["newline_every"], uncon_text=job["uncon_text"], uncon_image=job["uncon_image"], tokenizer=None,
                   text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB, generator=gen, rng=ReplayCpuRng(), **kw)
    got = torch.cat(stub.calls, 0)
    assert got.shape == calls_ref.shape
    assert torch.equal(got, calls_ref), f"first differing model call: {(got != calls_ref).any(1).nonzero()[0].item()}"


def test_m_t2i_stepwise_decodes_every_step_and_matches_t2i_generate(tiny_model):
    """t2i_generate_decoding_stepwise (modeling_mmada.py:768-875) = t2i_generate + a decode of the current samples after
    every step: same draws -> same final tokens; one PIL image per step at the decoder's output size."""
    from types import SimpleNamespace

    from mmada_parallel_amd import MAGVITv2, t2i_generate, t2i_generate_decoding_stepwise
    from oracle.interleave_oracle import SeededRng

    N, P = 64, 9
    g = torch.Generator().manual_seed(4)
    prompt = torch.randint(0, 2000, (1, P), generator=g)
    tail = torch.cat([torch.full((1, 1), 2040), torch.full((1, N), synth.MASK, dtype=torch.long), torch.full((1, 1), 2041)], 1)
    inp, unc = torch.cat([prompt, tail], 1), torch.cat([torch.randint(0, 2000, (1, P), generator=g), tail], 1)

    class Tok:
        def __len__(self):
            return synth.TEXT_VOCAB

    kw = dict(temperature=1.0, timesteps=4, guidance_scale=2.0, seq_len=N, resolution=6, codebook_size=synth.CODEBOOK,
              uni_prompting=SimpleNamespace(text_tokenizer=Tok()))
    a_in = inp.clone()
    a = t2i_generate(tiny_model, input_ids=a_in, uncond_input_ids=unc.clone(), rng=SeededRng(3), **kw)
    vq = MAGVITv2.from_state_dict(synth.synthetic_vq_state_dict(synth.VQ_CFG_TINY, 2), synth.VQ_CFG_TINY, device=DEV)
    b_in = inp.clone()
    frames = list(t2i_generate_decoding_stepwise(tiny_model, input_ids=b_in, uncond_input_ids=unc.clone(), vq_model=vq,
                                                 rng=SeededRng(3), **kw))
    assert [f[1] for f in frames] == [f"Step {i}/4" for i in range(1, 5)]
    assert all(f[0].size == (16, 16) for f in frames)          # 8x8 codes, 2-level decoder -> 16x16 pixels
    assert torch.equal(a_in, b_in)                              # same in-place final input_ids
    last = to_uint8 = None
    from mmada_parallel_amd.vq import to_uint8_image
    want = to_uint8_image(vq.decode_code(torch.clamp(a, 0, 8191)))[0].cpu().numpy()
    assert (np.asarray(frames[-1][0]) == want).all()


# ------------------------------------------------------------------------------------- hipGraph replay of a denoise step
@pytest.mark.parametrize("case", ["img4", "both", "nocfg"])
def test_graph_replayed_steps_are_bit_identical_to_eager(tiny_model, case):
    """generate_ti2ti(graph=True): every step kind is captured once (mmada_graph_*) and replayed; the trajectory must be
    bit-identical to the eager loop — same kernels, same buffers, same order — including the RNG stream position."""
    from mmada_parallel_amd import generate_ti2ti

    job, kw = tiny_job(), SAMPLER_CASES[case]

    def run(graph):
        torch.manual_seed(1234)
        out = generate_ti2ti(tiny_model, job["input_ids"].to(DEV), job["text_start"], job["text_end"], job["image_start"],
                             job["seq_len"], job["newline_every"], temperature=0.0, text_temperature=0.0,
                             uncon_text=job["uncon_text"], uncon_image=job["uncon_image"], return_state=True, graph=graph,
                             **kw)
        return out[2], torch.cuda.get_rng_state(0).clone()

    tiny_model.graph_replays, tiny_model.graph_nodes = 0, {}
    eager, rng_e = run(False)
    assert tiny_model.graph_replays == 0
    captured, rng_g = run(True)
    assert torch.equal(eager, captured)
    assert torch.equal(rng_e, rng_g), "the device RNG must have advanced exactly as in the eager loop"
    n_img = len(set(torch.linspace(kw["text_steps"] // 4, kw["text_steps"] - 1, kw["timesteps"]).round().int().tolist()))
    # the first step of each kind runs eagerly, every other step is a replay
    assert kw["text_steps"] - 4 <= tiny_model.graph_replays <= kw["text_steps"] - len(tiny_model.graph_nodes)
    assert all(n > 10 for n in tiny_model.graph_nodes.values()), tiny_model.graph_nodes
    print(f"{case}: {tiny_model.graph_replays} replays, nodes per step kind {tiny_model.graph_nodes}, image steps {n_img}")


def test_graph_with_a_batch_whose_unconditional_pair_outgrows_the_default_workspace():
    """Round-2 advisor finding: with the default max_batch (3) a batch of 2 with unconditional prompts runs its first 4-row
    forward only at the first image step — AFTER the text-only step graph was captured.  The workspace is now sized before
    the loop (and captured graphs are dropped if it ever moves): a fresh model, batch 2, graph on == eager, bit for bit."""
    from mmada_parallel_amd import LLaDAForMultiModalGeneration, generate_ti2ti

    job, kw = tiny_job(), SAMPLER_CASES["img4"]
    ids = job["input_ids"].repeat(2, 1)
    ids[1, :4] = torch.tensor([11, 12, 13, 14])
    ut, ui = job["uncon_text"].repeat(2, 1), job["uncon_image"].repeat(2, 1)

    def run(graph):
        model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(synth.CFG_TINY), tiny_sd(), device=DEV)
        assert model.max_batch == 3 and model._ws_bytes[0] == 0
        out = generate_ti2ti(model, ids.to(DEV), job["text_start"], job["text_end"], job["image_start"], job["seq_len"],
                             job["newline_every"], temperature=0.0, text_temperature=0.0, uncon_text=ut, uncon_image=ui,
                             return_state=True, graph=graph, **kw)
        return out[2], model

    eager, _ = run(False)
    captured, model = run(True)
    assert model.graph_replays > 0 and model._ws_epoch == 1, "one allocation, before the first capture"
    assert torch.equal(eager, captured)
