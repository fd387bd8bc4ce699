"""N>1 path on CPU: world_size-2 gloo processes evaluate one block with the tensor-parallel shard plan
(mmada_parallel_amd/tp.py — the slicing the HIP repack kernels implement) and all-reduce the partial residual streams;
the result must equal the unsharded oracle block up to bf16 re-association of the two partial sums."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from helpers import ROOT, tiny_job, tiny_sd
from mmada_parallel_amd import synth, tp


def _partial_block(x, w, cfg, rank, size, sin, cos):
    """What one rank computes for a block: (x on the rows this rank owns the residual of, else 0) + local partial,
    for attention then MLP, with the all-reduce in between (same sequence as mmada_attn_partial / all_reduce /
    mmada_mlp_partial; row m's residual belongs to rank (m >> 4) % size, kernels.h GemmArgs)."""
    from oracle import llada_oracle as lo

    B, T, D = x.shape
    hd = D // cfg["n_heads"]
    ws = tp.shard_layer_weights(w, cfg, rank, size)
    hq = ws["q_proj"].shape[0] // hd
    hkv = ws["k_proj"].shape[0] // hd
    eps = cfg["rms_norm_eps"]
    xn = lo.rms_norm(x, ws["attn_norm"], eps)
    q = F.linear(xn, ws["q_proj"]).view(B, T, hq, hd).transpose(1, 2)
    k = F.linear(xn, ws["k_proj"]).view(B, T, hkv, hd).transpose(1, 2)
    v = F.linear(xn, ws["v_proj"]).view(B, T, hkv, hd).transpose(1, 2)
    q, k = lo.apply_rope(q, sin, cos), lo.apply_rope(k, sin, cos)
    att = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).contiguous().view(B, T, hq * hd)
    part = F.linear(att, ws["attn_out"])
    own = (((torch.arange(B * T) >> 4) % size) == rank).view(B, T, 1)
    y = torch.where(own, x + part, part)
    dist.all_reduce(y)
    h = lo.rms_norm(y, ws["ff_norm"], eps)
    hm = F.silu(F.linear(h, ws["ff_proj"])) * F.linear(h, ws["up_proj"])
    part = F.linear(hm, ws["ff_out"])
    z = torch.where(own, y + part, part)
    dist.all_reduce(z)
    return z


def _worker(rank, size, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    torch.set_num_threads(2)
    from oracle import llada_oracle as lo

    cfg, sd, job = synth.CFG_TINY, tiny_sd(), tiny_job()
    ids = job["input_ids"]
    x = F.embedding(ids, sd["model.transformer.wte.weight"]).float()  # fp32 stream: isolates the sharding math
    sd32 = {k: v.float() for k, v in sd.items()}
    sin, cos = lo.rope_tables(ids.shape[1], 128, cfg["rope_theta"])
    w = lo.layer_weights(sd32, 0)
    z = _partial_block(x, w, cfg, rank, size, sin, cos)
    ref = lo.block_forward(x, w, cfg["n_heads"], cfg["n_kv_heads"], cfg["rms_norm_eps"], sin, cos)
    err = ((z - ref).abs().max() / ref.abs().max()).item()
    out[rank] = err
    dist.destroy_process_group()


def test_tp2_partial_sums_reproduce_the_block():
    size, port = 2, 29541 + os.getpid() % 200
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(size, port, out), nprocs=size, join=True)
    assert len(out) == size
    for r in range(size):
        assert out[r] < 1e-5, f"rank {r}: rel err {out[r]}"   # fp32 math: only summation order differs


def test_shard_plan_covers_every_feature_once():
    cfg = synth.CFG_8B
    for size in (1, 2, 4, 8):
        cover = {}
        for r in range(size):
            for name, (dim, sl) in tp.layer_shards(cfg, r, size).items():
                cover.setdefault(name, []).append((sl.start, sl.stop))
        for name, spans in cover.items():
            spans.sort()
            assert spans[0][0] == 0
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
        assert cover["q_proj"][-1][1] == cfg["d_model"] and cover["ff_out"][-1][1] == cfg["mlp_hidden_size"]
    with pytest.raises(ValueError):
        tp.layer_shards(dict(cfg, n_heads=30, d_model=3840), 0, 4)


# ---- the library's exchange (reduce-scatter -> residual + RMSNorm on owned rows -> all-gather) over gloo ----------------
def _exchange_worker(rank, size, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    torch.set_num_threads(2)
    from oracle import llada_oracle as lo

    g = torch.Generator().manual_seed(7)           # same data on every rank
    M, d, V = 200, 256, 1000
    x = torch.randn(M, d, generator=g).to(torch.bfloat16)
    parts = [torch.randn(M, d, generator=g).to(torch.bfloat16) for _ in range(size)]
    w = (1 + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16)
    # what csrc/tp_comm.hip does, rank-locally: sum the partial slices of MY rows in rank order (fp32), round, add the
    # residual, normalise, publish; then collect everybody's rows
    xn = torch.zeros(M, d, dtype=torch.bfloat16)
    mine = torch.zeros(M, dtype=torch.bool)
    for (m0, m1, sl), (r0, r1) in zip(tp.chunk_slices(M, size), tp.owned_rows(M, rank, size)):
        mine[r0:r1] = True
        pieces = [torch.zeros(r1 - r0, d) for _ in range(size)]
        # the "pull": every rank hands the owner its partial of the owner's rows (gloo stands in for the mapped buffers)
        for owner in range(size):
            o0, o1 = min(m1, m0 + owner * sl), min(m1, m0 + (owner + 1) * sl)
            buf = [torch.zeros(o1 - o0, d) for _ in range(size)] if rank == owner else None
            dist.gather(parts[rank][o0:o1].float(), buf, dst=owner)
            if rank == owner:
                pieces = buf
        acc = torch.zeros(r1 - r0, d)
        for j in range(size):
            acc = acc + pieces[j]
        x_new = (x[r0:r1].float() + acc.to(torch.bfloat16).float()).to(torch.bfloat16)
        xn[r0:r1] = lo.rms_norm(x_new, w, 1e-5)
    got = [torch.zeros(M, d, dtype=torch.float32) for _ in range(size)]
    dist.all_gather(got, torch.where(mine[:, None], xn.float(), torch.zeros(M, d)))
    full = sum(got).to(torch.bfloat16)
    total = sum(p.float() for p in parts).to(torch.bfloat16)
    want = lo.rms_norm((x.float() + total.float()).to(torch.bfloat16), w, 1e-5)
    ok_rows = bool(torch.equal(full, want))
    cover = torch.zeros(M)
    dist.all_reduce(cover.add_(mine.float()))
    # vocabulary-parallel text statistics: {max, first arg-max, sum-exp} of my columns, combined like tp_text_combine_kernel
    logits = torch.randn(6, V, generator=g).to(torch.bfloat16)
    logits[2, 700] = logits[2].max()   # an exact tie across two vocabulary slices: the lowest column must win
    logits[2, 100] = logits[2].max()
    v0, v1 = tp.vocab_slice(V, rank, size)
    loc = logits[:, v0:v1].double()
    rec = torch.stack([loc.max(1).values, (loc.argmax(1) + v0).double(), (loc - loc.max(1, keepdim=True).values).exp().sum(1)], 1)
    recs = [torch.zeros_like(rec) for _ in range(size)]
    dist.all_gather(recs, rec)
    mx = torch.stack([r[:, 0] for r in recs]).max(0).values
    arg = torch.zeros(6, dtype=torch.long)
    tot = torch.zeros(6, dtype=torch.double)
    for j in reversed(range(size)):
        arg = torch.where(recs[j][:, 0] == mx, recs[j][:, 1].long(), arg)   # lowest rank holding the maximum wins
    tot = sum(recs[j][:, 2] * (recs[j][:, 0] - mx).exp() for j in range(size))
    p_ref = torch.softmax(logits.double(), -1)
    ok_text = bool(torch.equal(arg, logits.float().argmax(-1))) and \
        bool(((1.0 / tot - p_ref.gather(1, arg[:, None])[:, 0]).abs() / (1.0 / tot)).max() < 1e-12)
    out[rank] = (ok_rows, bool((cover == 1).all()), ok_text)
    dist.destroy_process_group()


@pytest.mark.parametrize("size", [2, 3])
def test_exchange_plan_over_gloo(size):
    """reduce-scatter -> residual add + RMSNorm on the owner's rows -> all-gather, and the vocabulary-parallel text
    statistics, with world_size gloo processes standing in for the mapped peer buffers: every row is owned exactly once
    and the result equals the one-process computation bit for bit."""
    port = 29741 + os.getpid() % 200 + size
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_exchange_worker, args=(size, port, out), nprocs=size, join=True)
    assert len(out) == size
    for r in range(size):
        assert out[r] == (True, True, True), f"rank {r}: {out[r]}"


def test_chunk_slices_cover_rows_once_and_match_padding_rules():
    for size in (2, 4, 8):
        for M in (8, 72, 216, 2440, 4880, 19520, 39040):
            for nch in (1, 2):
                seen = torch.zeros(M, dtype=torch.int32)
                plan = tp.chunk_slices(M, size, nch)
                for k, (m0, m1, sl) in enumerate(plan):
                    assert sl % 8 == 0 and size * sl >= m1 - m0 and size * sl <= (m1 - m0) + 8 * size
                    if k < len(plan) - 1:
                        assert (m1 - m0) == size * sl   # only the last chunk is padded
                for r in range(size):
                    for r0, r1 in tp.owned_rows(M, r, size, nch):
                        seen[r0:r1] += 1
                assert bool((seen == 1).all()), (size, M, nch)
    assert tp.vocab_slice(134656, 7, 8) == (117824, 134656) and tp.vocab_slice(134656, 0, 8) == (0, 16832)


def test_exchange_plan_degenerates_cleanly_to_one_rank():
    """The single-rank test switch of the library (mmada_set_option("tp_allow_single_rank"): tests/test_gpu_tp.py runs every
    transport on a one-rank group) relies on the exchange plan being well formed for size 1: one owner that owns every row of
    each chunk, chunks that tile [0, M) without overlap, and a vocabulary slice that is the whole head."""
    from mmada_parallel_amd import tp

    for M in (8, 24, 96, 2440, 4880):
        for nch in (1, 2):
            sl = tp.chunk_slices(M, 1, nch)
            assert sl[0][0] == 0 and sl[-1][1] == M and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            own = tp.owned_rows(M, 0, 1, nch)
            assert [(m0, m1) for m0, m1, _ in sl] == own, "the one rank owns every row"
            for m0, m1, s in sl:
                assert s >= m1 - m0 and s % 8 == 0
    assert tp.vocab_slice(134656, 0, 1) == (0, 134656)
