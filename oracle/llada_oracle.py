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
