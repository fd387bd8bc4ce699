"""CPU ORACLE (test infrastructure — never imported by the product path).

Plain-PyTorch-on-CPU restatement of the reference denoiser forward for the TI2TI path, with every rounding point
of the reference spelled out.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Follows (paths relative to /root/reference/MMaDA-Parallel-A):
    LLaDAForMultiModalGeneration.forward(infer=True)   model/modeling_xllmx_dimoo.py:41-72
    LLaDAModel.forward                                 model/modeling_llada.py:1201-1415
    LLaDALlamaBlock.forward                            model/modeling_llada.py:906-972
    LLaDABlock.attention / SDPA                        model/modeling_llada.py:643-744
    RotaryEmbedding                                    model/modeling_llada.py:363-435
    RMSLayerNorm.forward                               model/modeling_llada.py:301-329
    dLLM cache (use_cache / to_compute_mask / cat)     model/modeling_llada.py:593-600,929-940,1244-1245,1406-1426
Parity is PINNED: tests/test_oracle_golden.py checks this file against fixtures produced by importing and running
the reference itself (oracle/gen_golden.py -> tests/golden/).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    # modeling_llada.py:315-329: fp32 normalise -> cast to og dtype -> weight * x (no bias)
    og = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(og)


def rope_tables(seq_len: int, head_dim: int, theta: float):
    # modeling_llada.py:391-397 (fp32)
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    seq = torch.arange(seq_len, dtype=torch.float)
    freqs = torch.einsum("i , j -> i j", seq, inv_freq)
    positions = torch.cat((freqs, freqs), dim=-1)
    return positions.sin()[None, None, :, :], positions.cos()[None, None, :, :]


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    # modeling_llada.py:402-406
    B, nh, T, hs = x.size()
    x = x.view(B, nh, T, 2, hs // 2)
    x1, x2 = x.unbind(dim=-2)
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(t: torch.Tensor, sin: torch.Tensor, cos: torch.Tensor) -> torch.Tensor:
    # modeling_llada.py:408-435 with rope_full_precision=True: fp32 math, cast back
    tf = t.float()
    return ((tf * cos) + (rotate_half(tf) * sin)).to(t.dtype)


def block_forward(x: torch.Tensor, w: Dict[str, torch.Tensor], n_heads: int, n_kv_heads: int, eps: float,
                  sin: torch.Tensor, cos: torch.Tensor) -> torch.Tensor:
    # modeling_llada.py:906-972 (dropout p=0 is the identity)
    B, T, D = x.shape
    hd = D // n_heads
    xn = rms_norm(x, w["attn_norm"], eps)
    q = F.linear(xn, w["q_proj"])
    k = F.linear(xn, w["k_proj"])
    v = F.linear(xn, w["v_proj"])
    q = q.view(B, T, n_heads, hd).transpose(1, 2)
    k = k.view(B, T, n_kv_heads, hd).transpose(1, 2)
    v = v.view(B, T, n_kv_heads, hd).transpose(1, 2)
    q, k = apply_rope(q, sin, cos), apply_rope(k, sin, cos)
    if n_kv_heads != n_heads:  # :666-669
        k = k.repeat_interleave(n_heads // n_kv_heads, dim=1)
        v = v.repeat_interleave(n_heads // n_kv_heads, dim=1)
    att = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)  # :672-679
    att = att.transpose(1, 2).contiguous().view(B, T, D)
    x = x + F.linear(att, w["attn_out"])  # :741-744, :953
    og = x
    h = rms_norm(x, w["ff_norm"], eps)
    g, u = F.linear(h, w["ff_proj"]), F.linear(h, w["up_proj"])  # :962
    h = F.silu(g) * u  # :966-967
    return og + F.linear(h, w["ff_out"])  # :968-970


def layer_weights(sd: Dict[str, torch.Tensor], i: int) -> Dict[str, torch.Tensor]:
    p = f"model.transformer.blocks.{i}."
    names = ["attn_norm", "ff_norm", "q_proj", "k_proj", "v_proj", "attn_out", "ff_proj", "up_proj", "ff_out"]
    return {n: sd[p + n + ".weight"] for n in names}


@torch.no_grad()
def forward_hidden(sd: Dict[str, torch.Tensor], cfg: dict, input_ids: torch.Tensor,
                   taps: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """Residual stream after the last block, [B, L, d] (before ln_f).  `taps` collects the stream after each block."""
    n_heads = cfg["n_heads"]
    n_kv = cfg.get("n_kv_heads") or n_heads
    eps = cfg.get("rms_norm_eps", 1e-5)
    x = F.embedding(input_ids, sd["model.transformer.wte.weight"])  # :1265
    L = input_ids.shape[1]
    sin, cos = rope_tables(L, cfg["d_model"] // n_heads, cfg.get("rope_theta", 10000.0))
    sin, cos = sin.to(x.device), cos.to(x.device)
    for i in range(cfg["n_layers"]):
        x = block_forward(x, layer_weights(sd, i), n_heads, n_kv, eps, sin, cos)
        if taps is not None:
            taps.append(x)
    return x


@torch.no_grad()
def head(sd: Dict[str, torch.Tensor], cfg: dict, x: torch.Tensor, col_begin: int = 0,
         col_end: Optional[int] = None) -> torch.Tensor:
    """ln_f + LM head (modeling_llada.py:1392, 1399-1404) restricted to a column range."""
    xn = rms_norm(x, sd["model.transformer.ln_f.weight"], cfg.get("rms_norm_eps", 1e-5))
    w = sd["model.transformer.ff_out.weight"]
    return F.linear(xn, w[col_begin:col_end])


@torch.no_grad()
def forward_logits(sd: Dict[str, torch.Tensor], cfg: dict, input_ids: torch.Tensor) -> torch.Tensor:
    """Full [B, L, V] logits, i.e. model(ids, infer=True).logits of the reference."""
    return head(sd, cfg, forward_hidden(sd, cfg, input_ids))


# ---- dLLM cache (model/modeling_llada.py:593-600,929-940,1244-1245,1406-1426) -----------------------------------------
class DllmCache:
    """State of LLaDAModel's caches: per block {'k': {cat: [B,L,D]}, 'v': {...}} (:593-597) and logit_cache {cat: [B,L,V]}
    (:1420), plus the blocks' use_cache flag that caching(enable) sets (:598-600)."""

    def __init__(self, n_layers: int):
        self.n_layers = n_layers
        self.use_cache = False
        self.empty_cache()

    def caching(self, enable: bool = True):  # :1417-1421
        self.use_cache = enable
        self.empty_cache()

    def empty_cache(self):  # :1423-1426
        self.k = [dict() for _ in range(self.n_layers)]
        self.v = [dict() for _ in range(self.n_layers)]
        self.logits = {}


def block_forward_cached(x, w, n_heads, n_kv_heads, eps, cache: DllmCache, layer: int, cat, to_compute_mask):
    """LLaDALlamaBlock.forward with use_cache=True (:906-972): the fresh k / v rows replace the cached rows at the masked
    positions (kept UN-rotated, [B, L, D]); the whole cached k is rotated by its positions, q by the masked positions when
    the block's use_cache flag is on, else by the LAST T positions (:714-716,416-428)."""
    B, T, D = x.shape
    hd = D // n_heads
    xn = rms_norm(x, w["attn_norm"], eps)
    q = F.linear(xn, w["q_proj"])
    k = F.linear(xn, w["k_proj"])
    v = F.linear(xn, w["v_proj"])
    if cat not in cache.k[layer]:  # :930-932 — zeros_like(x): the cache assumes k / v are as wide as the stream
        cache.k[layer][cat] = torch.zeros_like(x)
        cache.v[layer][cat] = torch.zeros_like(x)
    if to_compute_mask is not None:  # :933-937
        cache.k[layer][cat][to_compute_mask] = k.view(-1, D)
        cache.v[layer][cat][to_compute_mask] = v.view(-1, D)
        k, v = cache.k[layer][cat], cache.v[layer][cat]
    else:  # :938-940
        cache.k[layer][cat], cache.v[layer][cat] = k, v
    q = q.view(B, -1, n_heads, hd).transpose(1, 2)
    k = k.view(B, -1, n_kv_heads, hd).transpose(1, 2)
    v = v.view(B, -1, n_kv_heads, hd).transpose(1, 2)
    key_len, query_len = k.shape[-2], q.shape[-2]
    sin, cos = rope_tables(key_len, hd, _THETA[0])
    if cache.use_cache and to_compute_mask is not None:  # :714-716 -> :426-431
        idx = to_compute_mask.nonzero(as_tuple=True)[1]
        q = apply_rope(q, sin[:, :, idx, :], cos[:, :, idx, :])
    else:  # :421-425
        q = apply_rope(q, sin[:, :, key_len - query_len:key_len, :], cos[:, :, key_len - query_len:key_len, :])
    k = apply_rope(k, sin, cos)
    if n_kv_heads != n_heads:
        k = k.repeat_interleave(n_heads // n_kv_heads, dim=1)
        v = v.repeat_interleave(n_heads // n_kv_heads, dim=1)
    att = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
    att = att.transpose(1, 2).contiguous().view(B, T, D)
    x = x + F.linear(att, w["attn_out"])
    og = x
    h = rms_norm(x, w["ff_norm"], eps)
    g, u = F.linear(h, w["ff_proj"]), F.linear(h, w["up_proj"])
    h = F.silu(g) * u
    return og + F.linear(h, w["ff_out"])


_THETA = [10000.0]


@torch.no_grad()
def forward_logits_cached(sd, cfg, input_ids, cache: DllmCache, to_compute_mask=None, cat=""):
    """LLaDAModel.forward(input_ids, use_cache=True, to_compute_mask=..., cat=...) -> the WHOLE logit cache [B, L, V]
    (:1244-1245 token gather, :1352-1381 blocks, :1406-1413 logit scatter)."""
    n_heads = cfg["n_heads"]
    n_kv = cfg.get("n_kv_heads") or n_heads
    eps = cfg.get("rms_norm_eps", 1e-5)
    _THETA[0] = cfg.get("rope_theta", 10000.0)
    B = input_ids.shape[0]
    ids = input_ids[to_compute_mask].view(B, -1) if to_compute_mask is not None else input_ids  # :1244-1245
    x = F.embedding(ids, sd["model.transformer.wte.weight"])
    for i in range(cfg["n_layers"]):
        x = block_forward_cached(x, layer_weights(sd, i), n_heads, n_kv, eps, cache, i, cat, to_compute_mask)
    logits = head(sd, cfg, x)
    if cat not in cache.logits:  # :1407-1408
        cache.logits[cat] = torch.zeros_like(logits)
    if to_compute_mask is not None:  # :1409-1411
        cache.logits[cat][to_compute_mask] = logits.view(-1, logits.shape[-1])
        logits = cache.logits[cat]
    else:
        cache.logits[cat] = logits
    return logits
