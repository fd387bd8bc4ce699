"""Tensor-parallel shard plan of one LLaDA block (the slicing the bind-time repack kernels implement,
csrc/elementwise.hip pack_qkv/pack_gate_up/pack_cols) as plain index arithmetic, so it can be checked on CPU.

Megatron-style: q/k/v and ff_proj/up_proj are column-parallel (output-feature slices: whole heads / MLP columns),
attn_out and ff_out are row-parallel (input-feature slices).  Each rank produces a partial sum of the two
row-parallel outputs, rounded to bf16; the residual of row m is added by exactly one rank — its owner
`residual_owner(m, size)` = (m >> 4) % size, i.e. ownership rotates over the ranks in 16-row groups so every rank reads
1/size of the residual stream (csrc/gemm.hip EPI_RESID, GemmArgs::resid_mod/resid_rank) — so the sum over ranks of the
per-rank buffers IS the new residual stream (reference block: model/modeling_llada.py:906-972).
Because every rank rounds its partial to bf16 before the sum, TP=k results differ from TP=1 by bf16 rounding of the
partials (not bit-identical; the tolerance is stated in INTEGRATION.md and asserted in tests/test_gpu_model.py).
"""
from __future__ import annotations

from typing import Dict, Tuple


def residual_owner(row: int, size: int) -> int:
    """Rank whose row-parallel GEMM epilogue adds the residual of stream row `row` (16-row groups, round robin)."""
    return (row >> 4) % size


def head_range(n_heads: int, rank: int, size: int) -> Tuple[int, int]:
    if n_heads % size:
        raise ValueError(f"{n_heads} heads not divisible by tp_size {size}")
    per = n_heads // size
    return rank * per, (rank + 1) * per


def layer_shards(cfg: dict, rank: int, size: int) -> Dict[str, Tuple[int, slice]]:
    """name -> (dim, slice) of the checkpoint tensor `blocks.{i}.<name>.weight` kept by `rank`."""
    hd = cfg["d_model"] // cfg["n_heads"]
    n_kv = cfg.get("n_kv_heads") or cfg["n_heads"]
    F = cfg["mlp_hidden_size"]
    if F % size:
        raise ValueError("mlp_hidden_size not divisible by tp_size")
    q0, q1 = head_range(cfg["n_heads"], rank, size)
    k0, k1 = head_range(n_kv, rank, size)
    f0, f1 = rank * (F // size), (rank + 1) * (F // size)
    return {
        "q_proj": (0, slice(q0 * hd, q1 * hd)), "k_proj": (0, slice(k0 * hd, k1 * hd)),
        "v_proj": (0, slice(k0 * hd, k1 * hd)), "attn_out": (1, slice(q0 * hd, q1 * hd)),
        "ff_proj": (0, slice(f0, f1)), "up_proj": (0, slice(f0, f1)), "ff_out": (1, slice(f0, f1)),
    }


def shard_layer_weights(weights: dict, cfg: dict, rank: int, size: int) -> dict:
    """Apply layer_shards to a dict of full block tensors (norm weights are replicated)."""
    plan = layer_shards(cfg, rank, size)
    out = {}
    for name, w in weights.items():
        if name in plan:
            dim, sl = plan[name]
            out[name] = w[sl] if dim == 0 else w[:, sl]
        else:
            out[name] = w
    return out


# ---- the library's exchange step (csrc/tp_comm.hip) as plain index arithmetic --------------------------------------------
def chunk_slices(M: int, size: int, n_chunks: int = 2):
    """Row plan of one exchange over M stream rows: [(m0, m1, slice)] per chunk; inside a chunk rank r owns rows
    [m0 + r*slice, min(m1, m0 + (r+1)*slice)).  Every chunk but the last is a multiple of 8*size rows, so only the last one
    is padded (chunk_slice in csrc/tp_comm.hip)."""
    unit = 8 * size
    if n_chunks == 2 and M >= 4 * unit:
        first = (M // 2 + unit - 1) // unit * unit
        bounds = [(0, min(first, M)), (min(first, M), M)]
    else:
        bounds = [(0, M)]
    out = []
    for m0, m1 in bounds:
        rows = m1 - m0
        sl = max(8, ((rows + size - 1) // size + 7) // 8 * 8)
        out.append((m0, m1, sl))
    return out


def owned_rows(M: int, rank: int, size: int, n_chunks: int = 2):
    """Row ranges of the residual stream rank `rank` keeps (and normalises) — one per chunk."""
    return [(min(m1, m0 + rank * sl), min(m1, m0 + (rank + 1) * sl)) for m0, m1, sl in chunk_slices(M, size, n_chunks)]


def vocab_slice(V: int, rank: int, size: int) -> Tuple[int, int]:
    """Columns of ff_out.weight rank `rank` multiplies in the vocabulary-parallel text head (mmada_text_select_tp)."""
    w = ((V + size - 1) // size + 7) // 8 * 8
    v0 = min(V, rank * w)
    return v0, min(V, v0 + w)
