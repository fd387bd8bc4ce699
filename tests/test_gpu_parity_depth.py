"""Float parity of the denoiser at FULL depth and full size, with the measured errors recorded, not hidden.

`north_star` asks for "logits within 1e-3 bf16".  bf16 carries 8 significand bits (1 ulp = 2^-8 = 3.9e-3 relative), so
1e-3 is below one ulp: it can only hold per OPERATION on identical inputs (fp32 accumulation in another order, then ONE
rounding to bf16), never across 32 blocks whose every intermediate is re-rounded to bf16 — the reference's own bf16
forward is itself ~1e-2 away from exact arithmetic.  The suite therefore pins three things:

  (1) per-operation, identical bf16 inputs, 8B shapes: mean |err| / mean |ref| < 1e-3 for every contraction of the block,
      the RMSNorm, the attention and the LM head (`test_each_op_within_1e3_on_identical_inputs`);
  (2) full depth (32 blocks, d = 4096, L = 2438, BASELINE configs[1]): the HIP forward vs the CPU oracle per block and on
      the consumed LM-head rows, asserted against MEASURED values + headroom, and (unless MMADA_PARITY_FP32=0) next to the
      oracle's own bf16-vs-fp32 envelope — the HIP path must be as close to exact arithmetic as the reference's bf16
      path is (`test_full_depth_8b_forward_vs_oracle`);
  (3) every arg-max disagreement on the consumed rows is printed with the oracle's top-1/top-2 margin and must be a
      near-tie.

Measured numbers go to gpurun_out/r02_parity.json (copied to profiles/r02_parity.json when refreshed).
"""
import json
import os
import time

import pytest
import torch

from helpers import ROOT, host_threads as _host_threads, save_parity as _save, tp_each, tp_group
from mmada_parallel_amd import abi, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(got, ref):
    """(mean|err| / mean|ref|, max|err| / max|ref|) in fp32 on the CPU."""
    got, ref = got.float().cpu(), ref.float().cpu()
    e = (got - ref).abs()
    return (e.mean() / ref.abs().mean().clamp_min(1e-30)).item(), (e.max() / ref.abs().max().clamp_min(1e-30)).item()


# ------------------------------------------------------------------------------------------------ (1) per operation
def test_each_op_within_1e3_on_identical_inputs():
    """Every kernel family of the block on IDENTICAL bf16 inputs at 8B shapes (M = 2438 rows): the literal north-star
    tolerance, mean relative error < 1e-3 (well under one bf16 ulp on average: only round-to-nearest flips remain)."""
    import torch.nn.functional as F

    _host_threads()
    lib = abi.lib()
    st = abi.stream_ptr()
    g = torch.Generator().manual_seed(11)
    M, d, Fh = 2438, 4096, 12288
    out = {}

    def gemm(name, K, N, a_std=1.0):
        A = (torch.randn(M, K, generator=g) * a_std).to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16)
        ref = F.linear(A, W)                                  # CPU bf16 linear == the reference's nn.Linear arithmetic
        Ad, Wd = A.to(DEV), W.to(DEV)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        abi.check(lib.mmada_gemm_bt(Ad.data_ptr(), Wd.data_ptr(), C.data_ptr(), M, N, K, st), "gemm")
        out[name] = _rel(C, ref)

    gemm("q_proj_like_gemm_K4096_N4096", d, d)
    gemm("gate_up_like_gemm_K4096_N12288", d, Fh)
    gemm("down_gemm_K12288_N4096", Fh, d, a_std=0.3)
    # LM head: consumed text rows x full vocabulary (north_star: "logits within 1e-3")
    T, V = 256, 134656
    A = torch.randn(T, d, generator=g).to(torch.bfloat16)
    W = (torch.randn(V, d, generator=g) * d ** -0.5).to(torch.bfloat16)
    ref = F.linear(A, W)
    C = torch.empty(T, V, dtype=torch.bfloat16, device=DEV)
    Ad, Wd = A.to(DEV), W.to(DEV)
    abi.check(lib.mmada_gemm_bt(Ad.data_ptr(), Wd.data_ptr(), C.data_ptr(), T, V, d, st), "lm head")
    out["lm_head_logits_256x134656"] = _rel(C, ref)
    out["lm_head_argmax_agreement"] = (C.float().argmax(-1).cpu() == ref.float().argmax(-1)).float().mean().item()
    del Wd, W, C
    # RMSNorm (cast-then-scale order of the reference)
    from oracle import llada_oracle as lo

    x = (torch.randn(M, d, generator=g) * 0.7).to(torch.bfloat16)
    w = (1 + 0.02 * torch.randn(d, generator=g)).to(torch.bfloat16)
    y = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    xd, wd = x.to(DEV), w.to(DEV)
    abi.check(lib.mmada_rmsnorm(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), M, d, 1e-5, st), "rmsnorm")
    out["rmsnorm"] = _rel(y, lo.rms_norm(x, w, 1e-5))
    # attention: 32 heads x L = 2438 x 128, unmasked non-causal
    from mmada_parallel_amd import LLaDAForMultiModalGeneration

    cfg = dict(synth.CFG_8B, n_layers=1)
    sd = synth.synthetic_state_dict(cfg, seed=3, device=DEV)
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=DEV, max_batch=1)
    H, L = 32, 2438
    q = torch.randn(1, H, L, 128, generator=g).to(torch.bfloat16)
    k = torch.randn(1, H, L, 128, generator=g).to(torch.bfloat16)
    v = torch.randn(1, H, L, 128, generator=g).to(torch.bfloat16)
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(1, L, H * 128)      # reference arithmetic (bf16)
    exact = F.scaled_dot_product_attention(q.float(), k.float(), v.float()).transpose(1, 2).reshape(1, L, H * 128)
    o = torch.empty(1, L, H * 128, dtype=torch.bfloat16, device=DEV)
    model._ensure_ws(1, L)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    abi.check(lib.mmada_sdpa(model._handle, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), o.data_ptr(), 1, H, H, L, st), "sdpa")
    # P is rounded to bf16 before the PV product in the CPU kernel and in ours, against different running maxima, so the
    # two bf16 results are two independent roundings of the exact value (~2e-3 apart); both are compared with fp32
    out["attention_L2438_H32_vs_cpu_bf16"] = _rel(o, ref)
    out["attention_L2438_H32_vs_fp32"] = _rel(o, exact)
    out["attention_cpu_bf16_vs_fp32"] = _rel(ref, exact)
    print("per-op (mean rel, max rel):", json.dumps(out))
    _save("per_op_identical_inputs_8b_shapes", out)
    for name, v_ in out.items():
        if name.endswith("agreement"):
            assert v_ > 0.995
        elif name.startswith("attention"):
            continue
        else:
            assert v_[0] < 1e-3, f"{name}: mean relative error {v_[0]:.3e} is above the north-star 1e-3"
    # attention: as close to exact arithmetic as the reference's own bf16 kernel (measured 1.0x), and within 1 bf16 ulp of it
    # measured (round 6, 16x16x32 kernel): ours-vs-fp32 2.259e-3 = 1.031 x the CPU bf16 kernel's 2.191e-3; vs the CPU kernel 2.163e-3
    assert out["attention_L2438_H32_vs_fp32"][0] <= 1.06 * out["attention_cpu_bf16_vs_fp32"][0]
    assert out["attention_L2438_H32_vs_cpu_bf16"][0] < 2.4e-3


# ------------------------------------------------------------------------------------------------ (2)+(3) full depth
# Measured on MI355X in round 2 (profiles/r02_parity.json), limits = measured + ~25 % headroom.  Measured: residual-stream
# mean rel err HIP-vs-oracle 4.5e-3 after block 0 growing to 2.77e-2 after block 31, while BOTH are 2.85e-2 from exact fp32
# arithmetic (HIP 2.852e-2, reference bf16 2.854e-2); logits: mean |err| 0.0222 sigma, max 0.18 sigma, arg-max agreement
# 94.5 % (text) / 93.3 % (image) HIP-vs-oracle — the reference's own bf16 path agrees with exact arithmetic on 93.0 % / 92.6 %.
# measured + 10 % (round 6): stream 2.77e-2; text / image logits 2.22e-2 sigma; HIP-vs-fp32 = 1.000 x oracle-vs-fp32
LIM = dict(stream_mean_rel_last=3.1e-2, text_logit_mean_abs_in_sigma=2.45e-2, image_logit_mean_abs_in_sigma=2.45e-2,
           argmax_agree_min=0.90, envelope_ratio=1.03)


def _logit_report(name, got, ref, f32=None):
    got, ref = got.float().cpu(), ref.float().cpu()
    sigma = ref.std().item()
    e = (got - ref).abs()
    am_g, am_r = got.argmax(-1), ref.argmax(-1)
    bad = (am_g != am_r).nonzero().flatten().tolist()
    margins = []
    for r in bad:
        top2 = ref[r].topk(2).values
        margins.append({"row": r, "oracle_top1_minus_top2_sigma": ((top2[0] - top2[1]) / sigma).item(),
                        "oracle_logit_of_hip_choice_below_top1_sigma": ((top2[0] - ref[r, am_g[r]]) / sigma).item()})
    rep = {"rows": ref.shape[0], "cols": ref.shape[1], "logit_sigma": sigma, "mean_abs_err": e.mean().item(),
           "max_abs_err": e.max().item(), "mean_abs_err_in_sigma": e.mean().item() / sigma,
           "argmax_agreement": 1.0 - len(bad) / ref.shape[0], "disagreements": margins}
    if f32 is not None:
        f32 = f32.float().cpu()
        rep["envelope"] = {"hip_vs_fp32_mean_abs": (got - f32).abs().mean().item(),
                           "oracle_bf16_vs_fp32_mean_abs": (ref - f32).abs().mean().item(),
                           "hip_argmax_vs_fp32": (am_g == f32.argmax(-1)).float().mean().item(),
                           "oracle_bf16_argmax_vs_fp32": (am_r == f32.argmax(-1)).float().mean().item()}
    print(f"{name}: mean|err|={rep['mean_abs_err']:.4g} ({rep['mean_abs_err_in_sigma']:.3e} sigma) max|err|="
          f"{rep['max_abs_err']:.4g} argmax agreement {rep['argmax_agreement']:.4f}; disagreements (oracle margin in sigma): "
          f"{[round(m['oracle_top1_minus_top2_sigma'], 4) for m in margins][:12]}" + (f" envelope {rep['envelope']}" if f32 is not None else ""))
    return rep


def test_full_depth_8b_forward_vs_oracle():
    """32 blocks, d = 4096, F = 12288, L = 2438 (BASELINE configs[1] conditional forward): residual stream after every
    block and the consumed LM-head rows (text span x vocabulary, image positions x codebook slab) vs the CPU oracle."""
    from mmada_parallel_amd import LLaDAForMultiModalGeneration
    from oracle import llada_oracle

    _host_threads()
    want_f32 = os.environ.get("MMADA_PARITY_FP32", "1") != "0"   # exact-arithmetic envelope (adds ~1 min of host time)
    cfg = dict(synth.CFG_8B)
    n_layers = int(os.environ.get("MMADA_PARITY_LAYERS", cfg["n_layers"]))
    cfg["n_layers"] = n_layers
    want_tp = [int(t) for t in os.environ.get("MMADA_PARITY_TP", "2,4,8").split(",") if t]   # in-process rank groups
    want_cfg = os.environ.get("MMADA_PARITY_CFG", "1") != "0"   # one image step's dual-CFG decisions (2 more oracle forwards)
    sd_dev = synth.synthetic_state_dict(cfg, seed=3, device=DEV)
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd_dev, device=DEV, max_batch=2)
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids = job["input_ids"]
    B, L = ids.shape
    assert L == 2438
    ts, te, N = job["text_start"], job["text_end"], job["seq_len"]
    pos = [i for i in range(job["image_start"], job["image_start"] + N + N // job["newline_every"])
           if int(ids[0, i]) != synth.NEW_LINE]

    # ---- HIP: block by block through the segment entry points (the launch sequence of mmada_forward_body) ----
    lib, h = model._lib, model._handle
    ids_d = ids.to(DEV)
    model._ensure_ws(B, L)
    model._shape, model._split = (B, L), None
    abi.check(lib.mmada_set_consumed_rows(h, 0, 0), "rows")
    st = abi.stream_ptr()
    abi.check(lib.mmada_embed(h, ids_d.data_ptr(), B, L, st), "embed")
    taps_hip = []
    for i in range(n_layers):
        abi.check(lib.mmada_attn_partial(h, i, st), "attn")
        abi.check(lib.mmada_mlp_partial(h, i, st), "mlp")
        taps_hip.append(model.hidden_state().cpu())
    trow = torch.arange(ts, te, dtype=torch.int32, device=DEV)
    irow = torch.tensor(pos, dtype=torch.int32, device=DEV)
    V = cfg["embedding_size"]
    text_hip = model.head_rows(trow, 0, V).cpu()
    img_hip = model.head_rows(irow, synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK).cpu()
    # the production call (one launch sequence, last block on the consumed window) must give the same consumed rows
    model.forward_body(ids_d, consumed=(pos[0], te))
    assert torch.equal(model.head_rows(trow, 0, V).cpu(), text_hip)
    assert torch.equal(model.head_rows(irow, synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK).cpu(), img_hip)

    # ---- the unconditional pair of an image step (reference :243-274): [uncon_text | ids], [uncon_image | ids] ----
    unc = ids.repeat(2, 1)
    unc[0, :job["uncon_text"].shape[1]] = job["uncon_text"][0]
    unc[1, :job["uncon_image"].shape[1]] = job["uncon_image"][0]
    unc_hip = None
    if want_cfg:
        model.forward_body(unc.to(DEV))
        irow2 = torch.cat([irow, irow + L])
        unc_hip = model.head_rows(irow2, synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK).cpu().view(2, len(pos), -1)

    # ---- tensor-parallel forwards of the same sequence: the ranks of a TP group as handles of this process ----
    tp_hip = {}
    os.environ.setdefault("MMADA_TP_TIMEOUT_S", "20")
    for tp in want_tp:
        ranks, streams = tp_group(cfg, sd_dev, tp, (2 if want_cfg else 1) * ((L + 7) // 8 * 8))
        tp_each(ranks, streams, lambda m: m.forward_body(ids_d))
        hid = tp_each(ranks, streams, lambda m: m.hidden_state())
        tl = tp_each(ranks, streams, lambda m: m.head_rows(trow, 0, V))
        il = tp_each(ranks, streams, lambda m: m.head_rows(irow, synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK))
        for m in ranks:  # a timed-out hand-off (rig: two live streams on one hardware queue) voids everything below
            assert m.comm_status()["error"] == 0, f"TP={tp} rank {m.tp_rank}: {m.comm_status()}"
        for r in range(1, tp):  # every rank holds the same all-gathered rows
            assert torch.equal(hid[0], hid[r]), f"TP={tp}: residual stream of rank {r} differs from rank 0"
            assert torch.equal(tl[0], tl[r]) and torch.equal(il[0], il[r]), f"TP={tp}: logits of rank {r} differ"
        ul = None
        if want_cfg:   # the unconditional pair of the image step through the SAME rank group (a batch-2 forward)
            unc_d = unc.to(DEV)
            tp_each(ranks, streams, lambda m: m.forward_body(unc_d))
            irow2_tp = torch.cat([irow, irow + L])
            ul = tp_each(ranks, streams, lambda m: m.head_rows(irow2_tp, synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK))
            for m in ranks:
                assert m.comm_status()["error"] == 0, f"TP={tp} rank {m.tp_rank}: {m.comm_status()}"
            for r in range(1, tp):
                assert torch.equal(ul[0], ul[r]), f"TP={tp}: unconditional logits of rank {r} differ"
            ul = ul[0].cpu().view(2, len(pos), -1)
        tp_hip[tp] = (hid[0].cpu(), tl[0].cpu(), il[0].cpu(), ul)
        del ranks, streams, hid, tl, il
        torch.cuda.empty_cache()
    sd = {k: v.cpu() for k, v in sd_dev.items()}
    del sd_dev
    torch.cuda.empty_cache()

    # ---- oracle (reference arithmetic: bf16 storage, CPU) ----
    t0 = time.perf_counter()
    taps_ref = []
    x = llada_oracle.forward_hidden(sd, cfg, ids, taps=taps_ref)
    t_oracle = time.perf_counter() - t0
    text_ref = llada_oracle.head(sd, cfg, x[:, ts:te])[0]
    img_ref = llada_oracle.head(sd, cfg, x[:, pos], synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK)[0]
    taps_f32 = text_f32 = img_f32 = None
    if want_f32:  # exact-arithmetic envelope: same bf16 weights, every op in fp32
        sd32 = {k: v.float() for k, v in sd.items()}
        taps_f32 = []
        x32 = llada_oracle.forward_hidden(sd32, cfg, ids, taps=taps_f32)
        text_f32 = llada_oracle.head(sd32, cfg, x32[:, ts:te])[0]
        img_f32 = llada_oracle.head(sd32, cfg, x32[:, pos], synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK)[0]
        del sd32

    per_block = []
    for i in range(n_layers):
        row = {"block": i, "hip_vs_oracle": _rel(taps_hip[i], taps_ref[i])}
        if want_f32:
            row["hip_vs_fp32"] = _rel(taps_hip[i], taps_f32[i])
            row["oracle_bf16_vs_fp32"] = _rel(taps_ref[i], taps_f32[i])
        per_block.append(row)
    print("residual stream, mean rel err (HIP vs oracle) after block 0/7/15/23/last:",
          [f"{per_block[min(i, n_layers - 1)]['hip_vs_oracle'][0]:.3e}" for i in (0, 7, 15, 23, n_layers - 1)])
    if want_f32:
        print("   same vs fp32: HIP", [f"{per_block[min(i, n_layers - 1)]['hip_vs_fp32'][0]:.3e}" for i in (0, 7, 15, 23, n_layers - 1)],
              "oracle bf16", [f"{per_block[min(i, n_layers - 1)]['oracle_bf16_vs_fp32'][0]:.3e}" for i in (0, 7, 15, 23, n_layers - 1)])
    text_rep = _logit_report("text logits [256 x 134656]", text_hip, text_ref, text_f32)
    img_rep = _logit_report("image logits [1024 x 8192]", img_hip, img_ref, img_f32)
    _save("full_depth_8b_L2438" if n_layers == 32 else f"depth_{n_layers}_8b_L2438",
          {"n_layers": n_layers, "L": L, "oracle_forward_seconds": t_oracle, "host_threads": torch.get_num_threads(),
           "per_block_stream": per_block, "text_logits": text_rep, "image_logits": img_rep, "limits_asserted": LIM})

    last = per_block[-1]["hip_vs_oracle"]
    assert last[0] < LIM["stream_mean_rel_last"], f"residual stream mean rel err {last[0]:.3e}"
    assert text_rep["mean_abs_err_in_sigma"] < LIM["text_logit_mean_abs_in_sigma"]
    assert img_rep["mean_abs_err_in_sigma"] < LIM["image_logit_mean_abs_in_sigma"]
    for rep in (text_rep, img_rep):
        assert rep["argmax_agreement"] >= LIM["argmax_agree_min"]
        # a flipped arg-max must be a near-tie of the oracle's own logits: two logits cannot swap order unless their
        # errors add up to the margin, so the margin is bounded by twice the largest logit error of this very run
        bound = 2.0 * rep["max_abs_err"] / rep["logit_sigma"]
        for m in rep["disagreements"]:
            assert m["oracle_top1_minus_top2_sigma"] <= bound, (m, bound)
    if want_f32:  # the HIP path is as close to exact arithmetic as the reference's own bf16 path
        for rep in (text_rep, img_rep):
            env = rep["envelope"]
            assert env["hip_vs_fp32_mean_abs"] <= LIM["envelope_ratio"] * env["oracle_bf16_vs_fp32_mean_abs"], env

    # ---- tensor parallelism at full width and depth: bf16 partial sums must not leave the reference's own envelope ----
    # (round-2 review: on the 2-block tiny model TP = 2 was 6x further from TP = 1 than TP = 1 from the oracle; what decides
    # whether the reduce-scatter half must carry fp32 is the distance from EXACT arithmetic after 32 blocks at d = 4096)
    tp_rep = {}
    for tp, (hid, tl, il, _ul) in tp_hip.items():
        row = {"stream_vs_oracle": _rel(hid, taps_ref[-1]), "stream_vs_tp1": _rel(hid, taps_hip[-1]),
               "text_logits": _logit_report(f"TP={tp} text logits", tl, text_ref, text_f32),
               "image_logits": _logit_report(f"TP={tp} image logits", il, img_ref, img_f32)}
        if want_f32:
            row["stream_vs_fp32"] = _rel(hid, taps_f32[-1])
            row["oracle_bf16_stream_vs_fp32"] = _rel(taps_ref[-1], taps_f32[-1])
            row["tp1_stream_vs_fp32"] = _rel(taps_hip[-1], taps_f32[-1])
        tp_rep[f"tp{tp}"] = row
        print(f"TP={tp}: residual stream after block {n_layers - 1}: vs oracle {row['stream_vs_oracle'][0]:.3e}, vs TP=1 "
              f"{row['stream_vs_tp1'][0]:.3e}" + (f", vs fp32 {row['stream_vs_fp32'][0]:.3e} (TP=1 {row['tp1_stream_vs_fp32'][0]:.3e}, "
                                                  f"reference bf16 {row['oracle_bf16_stream_vs_fp32'][0]:.3e})" if want_f32 else ""))
    if tp_rep:
        _save("tp_depth_8b" if n_layers == 32 else f"tp_depth_{n_layers}_8b", {"n_layers": n_layers, "L": L, "partials": "bf16", **tp_rep})
    # TP = k differs from TP = 1 only by the bf16 rounding of k partial sums per row-parallel GEMM: its consumed logits must
    # sit as close to TP = 1's as TP = 1's sit to the oracle's (both are one re-association of the same bf16 arithmetic)
    tp1_img_vs_oracle = (img_hip.float() - img_ref.float()).abs().mean().item()
    for tp, (_hid, _tl, il, ul) in tp_hip.items():
        d_img = (il.float() - img_hip.float()).abs().mean().item()
        tp_rep[f"tp{tp}"]["image_logits_vs_tp1_mean_abs"] = d_img
        tp_rep[f"tp{tp}"]["tp1_image_logits_vs_oracle_mean_abs"] = tp1_img_vs_oracle
        assert d_img <= 1.5 * tp1_img_vs_oracle, (tp, d_img, tp1_img_vs_oracle)
        if ul is not None and unc_hip is not None:
            d_unc = (ul.float() - unc_hip.float()).abs().mean().item()
            tp_rep[f"tp{tp}"]["uncond_pair_logits_vs_tp1_mean_abs"] = d_unc
            assert d_unc <= 1.5 * tp1_img_vs_oracle, (tp, d_unc, tp1_img_vs_oracle)
    if tp_rep:
        _save("tp_depth_8b" if n_layers == 32 else f"tp_depth_{n_layers}_8b", {"n_layers": n_layers, "L": L, "partials": "bf16", **tp_rep})
    for name, row in tp_rep.items():
        assert row["text_logits"]["argmax_agreement"] >= LIM["argmax_agree_min"], name
        assert row["image_logits"]["argmax_agreement"] >= LIM["argmax_agree_min"], name
        if want_f32:  # hip_tp_vs_fp32 <= 1.1 x oracle_bf16_vs_fp32, on the stream and on the consumed logits
            assert row["stream_vs_fp32"][0] <= 1.06 * row["oracle_bf16_stream_vs_fp32"][0], (name, row["stream_vs_fp32"])   # measured 1.032-1.039
            for rep in (row["text_logits"], row["image_logits"]):
                env = rep["envelope"]
                assert env["hip_vs_fp32_mean_abs"] <= 1.1 * env["oracle_bf16_vs_fp32_mean_abs"], (name, env)

    # ---- one image step at full depth through the dual-CFG combine (reference :282-295,311): c + 4 (c - u_img) ----
    if want_cfg:
        from mmada_parallel_amd.generators.parallel_generator import mask_len_schedule
        from oracle import sampler_oracle as so

        xu = llada_oracle.forward_hidden(sd, cfg, unc)
        unc_ref = llada_oracle.head(sd, cfg, xu[:, pos], synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK).contiguous()
        Nq = len(pos)
        sets = {"oracle": (img_ref.view(1, Nq, -1).contiguous(), unc_ref), "hip": (img_hip.view(1, Nq, -1).contiguous(), unc_hip)}
        if want_f32:  # exact arithmetic through the same combine: what the reference's own bf16 evaluation agrees with
            sd32 = {k: v.float() for k, v in sd.items()}
            xu32 = llada_oracle.forward_hidden(sd32, cfg, unc)
            unc_f32 = llada_oracle.head(sd32, cfg, xu32[:, pos], synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK).contiguous()
            del sd32, xu32
            sets["fp32"] = (img_f32.view(1, Nq, -1).contiguous(), unc_f32)
            unc_env = {"hip_vs_fp32_mean_abs": (unc_hip.float() - unc_f32.float()).abs().mean().item(),
                       "oracle_bf16_vs_fp32_mean_abs": (unc_ref.float() - unc_f32.float()).abs().mean().item()}
        for tp, (_, _, il, ul) in tp_hip.items():   # the whole image step under tensor parallelism: all three branches
            sets[f"hip_tp{tp}"] = (il.view(1, Nq, -1).contiguous(), ul if ul is not None else unc_hip)
        res = {}
        for name, (c, u) in sets.items():
            am, pm, probs = so.image_probs(c.to(torch.bfloat16), u[0:1].contiguous().to(torch.bfloat16),
                                           u[1:2].contiguous().to(torch.bfloat16), 0.0, 4.0, want_probs=True)
            res[name] = (am, pm, probs)
        mlen = mask_len_schedule(Nq, 128)
        ids_img = ids.clone()
        cfg_rep = {}
        am_o, pm_o, probs_o = res["oracle"]
        for name in [n for n in res if n != "oracle"]:
            am, pm, _ = res[name]
            diff = (am != am_o)[0]
            if "fp32" in res and name != "fp32":
                am_x = res["fp32"][0]
            # oracle probability of the token the HIP logits chose, relative to the oracle's own maximum
            ratio = (probs_o[0, torch.arange(Nq), am[0].long()].float() / pm_o[0].float().clamp_min(1e-30))
            row = {"slots": Nq, "post_cfg_argmax_agreement": 1.0 - diff.float().mean().item(),
                   "worst_oracle_prob_ratio_of_hip_token": ratio.min().item()}
            if "fp32" in res and name != "fp32":   # the envelope: this evaluation vs exact, next to the reference's bf16 vs exact
                row["argmax_agreement_with_fp32"] = (am == am_x).float().mean().item()
                row["oracle_bf16_argmax_agreement_with_fp32"] = (am_o == am_x).float().mean().item()
            for step in (32, 64, 96):      # three cut levels of the cosine schedule (config 1: 128 steps)
                zero = torch.zeros((1, Nq), dtype=torch.bfloat16)
                keep_o = so.image_commit(ids_img, pos, am_o, pm_o, zero, 0.0, mlen[step])[0, pos] == synth.MASK
                keep_h = so.image_commit(ids_img, pos, am, pm, zero, 0.0, mlen[step])[0, pos] == synth.MASK
                assert int(keep_o.sum()) == int(keep_h.sum())
                row[f"remasked_set_agreement_step{step}"] = 1.0 - (keep_o ^ keep_h).float().mean().item()
            cfg_rep[name] = row
            print(f"post-CFG decisions, {name} vs oracle logits:", row)
        if want_f32:
            cfg_rep["uncond_pair_logits_envelope"] = unc_env
        _save("post_cfg_full_depth_8b" if n_layers == 32 else f"post_cfg_depth_{n_layers}_8b", cfg_rep)
        # Measured in round 3 (profiles/r03_parity.json): the combine c + 4 (c - u) amplifies the ~0.023 sigma noise of three
        # independent bf16 forwards five-fold over 8192 near-uniform classes (random weights: every soft-max maximum is ~4e-4):
        # HIP and oracle agree on 57.5 % of the post-CFG arg-maxima and on 71-89 % of the re-masked set; the oracle's
        # probability of the HIP token is never below 0.44 of its own maximum.  What is ASSERTED is the envelope: the logits
        # of the unconditional pair are as close to exact (fp32) arithmetic as the reference's own bf16 evaluation (like the
        # conditional ones above), and the post-CFG arg-max agrees with the exact one about as often (measured 63.0 % vs
        # the reference's 68.4 % on 1024 slots — one standard deviation of that difference is 2.1 points).
        if want_f32:
            assert unc_env["hip_vs_fp32_mean_abs"] <= LIM["envelope_ratio"] * unc_env["oracle_bf16_vs_fp32_mean_abs"], unc_env
        for name, row in cfg_rep.items():
            if name in ("fp32", "uncond_pair_logits_envelope"):
                continue
            assert row["worst_oracle_prob_ratio_of_hip_token"] > 0.3, (name, row)
            assert row["post_cfg_argmax_agreement"] > 0.45, (name, row)
            if "argmax_agreement_with_fp32" in row:
                assert row["argmax_agreement_with_fp32"] >= row["oracle_bf16_argmax_agreement_with_fp32"] - 0.10, (name, row)
