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
