"""dLLM cache path on the GPU (mmada_cache_* / mmada_forward_cached behind LLaDAForMultiModalGeneration.forward(
use_cache=True, to_compute_mask=..., cat=...)) against the CPU oracle restatement of model/modeling_llada.py:593-600,
929-940,1244-1245,1406-1426 — which tests/test_oracle_golden.py pins to the reference bit for bit — plus the properties
that hold exactly: a prime call equals the plain forward, a mask over every token equals the plain forward, untouched rows
keep their logits, and a batch behaves as its sequences one by one."""
import time

import pytest
import torch

from helpers import save_parity, tiny_sd
from mmada_parallel_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def tiny_model():
    from mmada_parallel_amd import LLaDAForMultiModalGeneration

    return LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(synth.CFG_TINY), tiny_sd(), device=DEV)


def _logits(model, ids, **kw):
    return model(ids.to(DEV), infer=True, **kw).logits


@pytest.mark.parametrize("enable", [True, False])
def test_cache_script_vs_oracle(tiny_model, enable):
    """The fixture's call sequence (synth.dllm_cache_script): after every call the returned logit cache must agree with
    the oracle's within the bf16 re-association tolerance of the plain forward test; caching(False) reproduces the
    reference's rotary fallback for the queries (positions L-Tc..L-1)."""
    from oracle import llada_oracle

    sd, cfg = tiny_sd(), synth.CFG_TINY
    script = synth.dllm_cache_script() if enable else synth.dllm_cache_script()[:2]
    cache = llada_oracle.DllmCache(cfg["n_layers"])
    cache.caching(enable)
    tiny_model.caching(enable)
    worst = {"mean_rel": 0.0, "max_rel": 0.0, "argmax": 1.0}
    for n, (cat, ids, m) in enumerate(script):
        ref = llada_oracle.forward_logits_cached(sd, cfg, ids, cache, to_compute_mask=m, cat=cat)[0].float()
        got = _logits(tiny_model, ids, use_cache=True, to_compute_mask=m, cat=cat)[0].float().cpu()
        scale = ref.abs().max().item()
        err = (got - ref).abs()
        agree = (got.argmax(-1) == ref.argmax(-1)).float().mean().item()
        print(f"caching={enable} call {n} ({cat}, mask={None if m is None else int(m.sum())}): "
              f"max {err.max() / scale:.3e} mean {err.mean() / scale:.3e} argmax agreement {agree:.3f}")
        worst = {"mean_rel": max(worst["mean_rel"], err.mean().item() / scale),
                 "max_rel": max(worst["max_rel"], err.max().item() / scale), "argmax": min(worst["argmax"], agree)}
        # the plain tiny forward measures max 1.05e-2 / mean 6.1e-4 of the magnitude (tests/test_gpu_model.py); the cache
        # steps chain up to three forwards' worth of re-association noise through the cached keys / values
        assert err.max().item() < 3e-2 * scale and err.mean().item() < 2e-3 * scale
        assert agree >= 0.9
    save_parity(f"dllm_cache_script_caching_{'on' if enable else 'off'}", worst)
    tiny_model.empty_cache()


def test_prime_and_full_mask_equal_the_plain_forward_bitwise(tiny_model):
    (_, ids0, _), (_, ids1, m1) = synth.dllm_cache_script()[:2]
    tiny_model.caching(True)
    plain0 = _logits(tiny_model, ids0).clone()
    primed = _logits(tiny_model, ids0, use_cache=True, cat="c").clone()
    assert torch.equal(plain0, primed)
    plain1 = _logits(tiny_model, ids1).clone()
    allm = torch.ones_like(m1)
    full = _logits(tiny_model, ids1, use_cache=True, to_compute_mask=allm, cat="c")
    assert torch.equal(plain1, full)
    tiny_model.empty_cache()


def test_untouched_rows_keep_their_logits_and_computed_rows_change(tiny_model):
    (_, ids0, _), (_, ids1, m1) = synth.dllm_cache_script()[:2]
    tiny_model.caching(True)
    before = _logits(tiny_model, ids0, use_cache=True, cat="c").clone()
    after = _logits(tiny_model, ids1, use_cache=True, to_compute_mask=m1, cat="c")
    keep = ~m1[0]
    assert torch.equal(after[0, keep.to(DEV)], before[0, keep.to(DEV)])
    assert not torch.equal(after[0, m1[0].to(DEV)], before[0, m1[0].to(DEV)])
    # a second slot is independent of the first
    other = _logits(tiny_model, ids0, use_cache=True, cat="d")
    assert torch.equal(other, before)
    again = _logits(tiny_model, ids1, use_cache=True, to_compute_mask=m1, cat="c")
    # the same step again: the keys / values it scatters are the ones already there -> identical logits
    assert torch.equal(again, after)
    tiny_model.empty_cache()


def test_never_computed_positions_are_zero_and_errors_are_loud(tiny_model):
    """A compute-mask step on a fresh slot: the reference starts the cache at zeros (torch.zeros_like), so positions that
    were never computed have zero logits."""
    from mmada_parallel_amd import abi

    (_, ids0, _), (_, ids1, m1) = synth.dllm_cache_script()[:2]
    tiny_model.caching(True)
    lg = _logits(tiny_model, ids1, use_cache=True, to_compute_mask=m1, cat="fresh")
    assert torch.count_nonzero(lg[0, (~m1[0]).to(DEV)]) == 0
    assert torch.count_nonzero(lg[0, m1[0].to(DEV)]) > 0
    with pytest.raises(ValueError):
        _logits(tiny_model, ids1, to_compute_mask=m1)                     # mask without use_cache
    ragged = torch.cat([m1, m1.roll(1, 1)], 0)
    ragged[1, 0] = ~ragged[1, 0]
    with pytest.raises(ValueError):                                       # unequal counts: the reference's .view(B, -1)
        _logits(tiny_model, torch.cat([ids1, ids1], 0), use_cache=True, to_compute_mask=ragged, cat="r")
    with pytest.raises(ValueError):                                       # slot "fresh" holds B = 1
        _logits(tiny_model, torch.cat([ids1, ids1], 0), use_cache=True, to_compute_mask=torch.cat([m1, m1], 0), cat="fresh")
    tiny_model.forward_cached(ids0.to(DEV), cat="fresh")
    with pytest.raises(abi.MmadaError):                                   # no plain forward is resident after a cache step
        tiny_model.head_rows(torch.zeros(1, dtype=torch.int32, device=DEV), 0, 16)
    tiny_model.empty_cache()


def test_batch_of_two_equals_the_sequences_one_by_one(tiny_model):
    """B = 2 with DIFFERENT positions per sequence (the reference's rotary q_mask only indexes one sequence; per-sequence
    positions are the natural extension): every sequence must come out exactly as when it runs alone."""
    (_, ids0, _), (_, ids1, m1), (_, ids2, m2) = synth.dllm_cache_script()[:3]
    n = int(m1.sum())
    m2b = m2.clone()
    extra = (~m2b[0]).nonzero()[: n - int(m2b.sum()), 0]       # same count as m1, other positions
    m2b[0, extra] = True
    tiny_model.caching(True)
    singles = []
    for ids_a, ids_b, m in ((ids0, ids1, m1), (ids1, ids2, m2b)):
        _logits(tiny_model, ids_a, use_cache=True, cat="s")
        singles.append(_logits(tiny_model, ids_b, use_cache=True, to_compute_mask=m, cat="s")[0].clone())
    _logits(tiny_model, torch.cat([ids0, ids1], 0), use_cache=True, cat="b")
    both = _logits(tiny_model, torch.cat([ids1, ids2], 0), use_cache=True, to_compute_mask=torch.cat([m1, m2b], 0), cat="b")
    # the deferred soft-max rescale is decided per 32-query wave, so a query's last bits may depend on its wave mates:
    # batch and single runs place the same queries in the same waves here (same compact order), hence bitwise
    assert torch.equal(both[0], singles[0]) and torch.equal(both[1], singles[1])
    tiny_model.empty_cache()


def test_cache_step_at_8b_shapes_vs_oracle_and_timing():
    """Two 8B-shape blocks at L = 2438 (BASELINE configs[1] geometry): prime, then a compute-mask step over the output
    image + text span (1314 of 2438 tokens) after some tokens were committed — vs the oracle; and what the step costs
    against a full forward."""
    from helpers import host_threads
    from mmada_parallel_amd import LLaDAForMultiModalGeneration
    from oracle import llada_oracle

    host_threads()
    cfg = dict(synth.CFG_8B, n_layers=2)
    sd = synth.synthetic_state_dict(cfg, seed=5, device="cpu")
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=DEV, max_batch=1)
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids0 = job["input_ids"]
    L = ids0.shape[1]
    ts, te, im = job["text_start"], job["text_end"], job["image_start"]
    ids1 = ids0.clone()
    g = torch.Generator().manual_seed(2)
    ids1[0, ts:ts + 32] = torch.randint(0, 50000, (32,), generator=g)
    sel = (ids1[0] == synth.MASK).nonzero()[:, 0]
    sel = sel[(sel >= im) & (sel < ts)][:100]
    ids1[0, sel] = synth.TEXT_VOCAB + torch.randint(0, synth.CODEBOOK, (sel.numel(),), generator=g)
    m = torch.zeros(1, L, dtype=torch.bool)
    m[0, im:L] = True
    Tc = int(m.sum())
    model.caching(True)
    cache = llada_oracle.DllmCache(cfg["n_layers"])
    cache.caching(True)
    rows = torch.cat([torch.arange(ts, ts + 16), torch.arange(im + 1, im + 17), torch.arange(0, 8)]).to(torch.int32)

    def oracle_rows(ids, mask):
        # the oracle's step without materialising [L, V]: run its blocks, then the head on the probed rows of the cached
        # final stream (a logit row is a function of its residual row alone)
        n_heads, eps = cfg["n_heads"], cfg.get("rms_norm_eps", 1e-5)
        llada_oracle._THETA[0] = cfg.get("rope_theta", 10000.0)
        x = torch.nn.functional.embedding(ids[mask].view(1, -1) if mask is not None else ids, sd["model.transformer.wte.weight"])
        for i in range(cfg["n_layers"]):
            x = llada_oracle.block_forward_cached(x, llada_oracle.layer_weights(sd, i), n_heads, cfg.get("n_kv_heads") or n_heads,
                                                  eps, cache, i, "c", mask)
        if mask is None:
            cache.xfin = x.clone()
        else:
            cache.xfin[mask] = x.view(-1, x.shape[-1])
        return llada_oracle.head(sd, cfg, cache.xfin[:, rows.long()], 0, 4096)[0].float()

    ref0 = oracle_rows(ids0, None)
    model.forward_cached(ids0.to(DEV), cat="c")
    got0 = model.cache_head_rows("c", rows.to(DEV), 0, 4096).float().cpu()
    ref1 = oracle_rows(ids1, m)
    model.forward_cached(ids1.to(DEV), to_compute_mask=m, cat="c")
    got1 = model.cache_head_rows("c", rows.to(DEV), 0, 4096).float().cpu()
    rep = {}
    for name, got, ref in (("prime", got0, ref0), ("step", got1, ref1)):
        scale = ref.abs().max().item()
        err = (got - ref).abs()
        rep[name] = {"max_rel": err.max().item() / scale, "mean_rel": err.mean().item() / scale}
        print(f"8B-shape cache {name}: max {rep[name]['max_rel']:.3e} mean {rep[name]['mean_rel']:.3e} of the logit magnitude")
        assert err.max().item() < 3e-2 * scale and err.mean().item() < 3e-3 * scale
    assert torch.equal(got1[-8:], got0[-8:])        # prompt rows were not recomputed
    # cost of the step against a full forward (2 blocks; hipEvents on the current stream)
    ids1d, md = ids1.to(DEV), m.to(DEV)

    def timed(fn, n=9):
        """median of n synchronised calls after two warm-ups (a mean of five once recorded a 13 ms outlier of the host)"""
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[n // 2]

    t_full = timed(lambda: model.forward_body(ids1d))
    t_step = timed(lambda: model.forward_cached(ids1d, to_compute_mask=md, cat="c"))
    rep["timing_ms_2_blocks"] = {"full_forward": t_full, "cache_step": t_step, "computed_tokens": Tc, "L": L,
                                 "note": "the step includes the mask -> positions host work (nonzero) the reference also does"}
    print(f"8B-shape, 2 blocks, L={L}: full forward {t_full:.2f} ms, cache step over {Tc} tokens {t_step:.2f} ms")
    save_parity("dllm_cache_8b_shapes", rep)
    model.empty_cache()


def test_cache_head_rows_leaves_a_resident_plain_forward_intact(tiny_model):
    """Reading a slot's logits while a plain forward of ANOTHER shape is resident must not disturb that forward (the
    staging rows go to the resident carve's own gather buffer)."""
    (_, ids0, _), (_, ids1, _) = synth.dllm_cache_script()[:2]
    L = ids0.shape[1]
    tiny_model.caching(True)
    tiny_model.forward_cached(ids0.to(DEV), cat="c")
    rows1 = torch.arange(L, dtype=torch.int32, device=DEV)
    want = tiny_model.cache_head_rows("c", rows1, 0, 512).clone()
    tiny_model.forward_body(torch.cat([ids1, ids0], 0).to(DEV))            # B = 2 plain forward now resident
    rows2 = torch.arange(2 * L, dtype=torch.int32, device=DEV)
    before = tiny_model.head_rows(rows2, 0, 512).clone()
    got = tiny_model.cache_head_rows("c", rows1, 0, 512)
    after = tiny_model.head_rows(rows2, 0, 512)
    assert torch.equal(got, want) and torch.equal(before, after)
    assert torch.equal(after[L:], want)                                      # the same sequence, plain vs primed slot
    tiny_model.empty_cache()
