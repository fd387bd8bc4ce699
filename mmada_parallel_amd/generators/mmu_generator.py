"""MI355X-native `mmu_generate` / `mmu_generate_fast` — the MMaDA-Parallel-M text (multimodal-understanding) sampler
(MMaDA-Parallel-M/models/modeling_mmada.py:618-766): semi-autoregressive LLaDA decoding, `num_blocks` blocks of
`block_length` masked tokens appended to the prompt, `steps / num_blocks` denoising steps per block.

Per step: one forward (two sequences when cfg_scale > 0: the prompt-masked copy is the unconditional branch, :660-666),
`un + (cfg+1) * (cond - un)` in bf16, argmax (float64 Gumbel-max when temperature > 0, :49-60), float64 softmax
confidence, and the `k` most confident masked positions of the CURRENT block are committed (:684-689) — positions of
later blocks are excluded (:677), earlier blocks and the prompt are no longer masked.  The LM head runs only on the
current block's rows.

`attention_mask`: the reference turns it into an `attention_bias` tensor (:625-629) and hands that to the model, whose
forward accepts the argument but never uses it (MMaDA-Parallel-M/models/modeling_llada.py:1164-1345: the merge code is
commented out and the blocks only receive `attention_mask`, which these samplers do not pass).  Attention is therefore
UNMASKED in the reference whatever the padding, and so it is here: the argument is accepted and has no effect.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import abi
from ..model import LLaDAForMultiModalGeneration
from .interleave_generator import TorchRng, get_num_transfer_tokens
from .parallel_generator import check_tp_exchange


@torch.no_grad()
def mmu_generate(model, idx=None, input_embeddings=None, max_new_tokens=128, steps=128, block_length=128,
                 temperature=0.0, top_k=None, eot_token=None, cfg_scale=0.0, remasking='low_confidence', mask_id=126336,
                 attention_mask=None, rng=None, trace: Optional[list] = None, _stop_on_eot: bool = False):
    """Returns x [B, P + max_new_tokens] like the reference (:692).  `eot_token` is ignored here, as in the reference's
    mmu_generate; mmu_generate_fast honours it."""
    if not isinstance(model, LLaDAForMultiModalGeneration):
        raise TypeError("mmu_generate (MI355X) needs mmada_parallel_amd.LLaDAForMultiModalGeneration")
    if input_embeddings is not None:
        raise NotImplementedError("input_embeddings")
    if remasking != 'low_confidence':
        raise NotImplementedError(remasking)  # 'random' is torch.rand plumbing only; not on any call path of the reference
    del attention_mask  # dead in the reference (see the module docstring): attention is unmasked
    if int(model.config.get("mask_token_id", 126336)) != mask_id:
        raise ValueError("mask_id differs from the model's")
    rng = rng or TorchRng()
    lib, h, device = model._lib, model._handle, model.device
    idx = idx.to(device)
    B, P = idx.shape
    if bool((idx == mask_id).any()):
        raise ValueError("the prompt must not contain mask tokens")
    assert max_new_tokens % block_length == 0
    num_blocks = max_new_tokens // block_length
    assert steps % num_blocks == 0
    steps = steps // num_blocks
    L, V, BL = P + max_new_tokens, model.vocab, block_length
    x = torch.full((B, L), mask_id, dtype=torch.long, device=device)
    x[:, :P] = idx
    use_cfg = cfg_scale > 0.0
    scratch = torch.empty(B * BL * 16, dtype=torch.uint8, device=device)
    ar = torch.arange(BL, dtype=torch.int32, device=device)
    boff = (torch.arange(B, dtype=torch.int32, device=device) * L)[:, None]

    for nb in range(num_blocks):
        start = P + nb * BL
        block_mask = (x[:, start:start + BL] == mask_id).cpu()
        k_dev = get_num_transfer_tokens(block_mask, steps).t().contiguous().to(device=device, dtype=torch.int32)  # [steps, B]
        rows = (boff + start + ar[None, :]).reshape(-1).contiguous()          # cond rows, batch-major
        for i in range(steps):
            if use_cfg:
                un_x = x.clone()
                un_x[:, :P] = mask_id                                          # un_x[prompt_index] = mask_id (:662)
                both = torch.cat([x, un_x], dim=0).contiguous()
                if trace is not None:
                    trace.append(both.cpu().clone())
                model.forward_body(both, consumed=(start, start + BL))   # only the current block is decoded
                lg = model.head_rows(torch.cat([rows, rows + B * L]), 0, V)   # [2*B*BL, V]: cond rows then uncond rows
                cond, unc = lg[:B * BL], lg[B * BL:]
                base, other, scale = unc, cond, float(cfg_scale + 1)          # un + (cfg+1) * (cond - un)  (:666)
            else:
                if trace is not None:
                    trace.append(x.cpu().clone())
                model.forward_body(x, consumed=(start, start + BL))
                cond = model.head_rows(rows, 0, V)
                base, other, scale = cond, cond, 0.0
            x0_in = None
            if temperature != 0:
                # add_gumbel_noise (:49-60) draws float64 noise for the WHOLE [B, L, V] logits; only the current block's
                # rows are consumed, but the draw keeps the reference's shape so the RNG stream stays the reference's
                comb = (base + scale * (other - base)) if use_cfg else cond    # bf16 tensor ops, as in the reference
                noise = rng.rand_f64((B, L, V), device)[:, start:start + BL]
                l64 = comb.view(B, BL, V).to(torch.float64)
                x0_in = torch.argmax(l64.exp() / ((-torch.log(noise)) ** temperature), dim=-1).to(torch.int32).contiguous()
            abi.check(lib.mmada_text_select_cfg(h, base.data_ptr(), other.data_ptr(), scale, abi.ptr(x0_in), B, BL, V, V,
                                                x.data_ptr(), L, start, k_dev[i].data_ptr(), scratch.data_ptr(),
                                                abi.stream_ptr()), "mmada_text_select_cfg")
        if _stop_on_eot and eot_token is not None:                             # mmu_generate_fast :756-761
            last = start + BL - 1
            if last < L and bool((x[:, last] == eot_token).all()):
                break
    check_tp_exchange(model)   # tensor parallel: a timed-out hand-off raises instead of returning void tokens
    return x


@torch.no_grad()
def mmu_generate_fast(model, idx=None, **kw):
    """modeling_mmada.py:694-766: mmu_generate that stops after a block whose last token is `eot_token` in every row."""
    return mmu_generate(model, idx, _stop_on_eot=True, **kw)
