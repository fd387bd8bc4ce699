"""CPU ORACLE (test infrastructure only): restatement of the generate_ti2ti loop
(/root/reference/MMaDA-Parallel-A/generators/parallel_generator.py:102-368) on top of the C sampler oracle.

`model_fn(ids) -> logits [B, L, V] bf16` stands for `model(ids, infer=True, use_cache=False).logits` (:178,263,264).
Every model call's input ids are appended to `trace` so tests can compare trajectories call by call against the
fixtures recorded from the reference (tests/golden/sampler_traj.npz, e2e_tiny.<host class>.npz).
B == 1 only, like the reference's image branch (:166,224,340).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional

import torch

from . import sampler_oracle as so

MASK_TOKEN, NEW_LINE = 126336, 126084


def get_num_transfer_tokens(n_masked: int, steps: int) -> List[int]:
    # :78-99
    out, remaining = [], n_masked
    for s in range(steps):
        target = int(n_masked * (1 - (s + 1) / steps))
        n = max(0, remaining - target)
        out.append(n)
        remaining -= n
    return out


def generate(model_fn: Callable[[torch.Tensor], torch.Tensor], input_ids: torch.Tensor, text_start: int, text_end: int,
             image_start: int, seq_len: int, newline_every: int, text_steps: int, timesteps: int,
             cfg_scale: float, cfg_img: float, uncon_text: Optional[torch.Tensor], uncon_image: Optional[torch.Tensor],
             text_vocab_size: int = 126356, codebook_size: int = 8192, trace: Optional[list] = None,
             image_step_list: Optional[list] = None, temperature: float = 0.0, text_temperature: float = 0.0,
             generator=None, remasking: str = "low_confidence", tie_order: str = "stable",
             commit_trace: Optional[list] = None) -> torch.Tensor:
    """Returns the final ids before the random fill (:360-362).  temperature / text_temperature > 0 draw from
    `generator` (a CPU generator) with the reference's calls in the reference's order: torch.rand for the text Gumbel
    noise (:13-16), torch.multinomial for the image tokens (:297-302), torch.randn for the re-mask jitter (:30-33)."""
    # tie_order: which of several EXACTLY tied bf16 confidences stay masked at the re-mask cut (:41).  The reference calls
    # torch.sort without stable=True; on CPU that is stable for small rows (N <= 16 observed) and NOT stable beyond (PyTorch
    # 2.10, AVX-512: 20 of 20 tie-heavy rows of N >= 64 come out in another order than the stable sort) — the order of ties
    # is a property of the PyTorch build and CPU, not of the algorithm.  "stable" = lowest index first (the C oracle and the
    # HIP kernel); "torch" = call torch.sort here exactly as the reference does (same host + same PyTorch = same order),
    # which is what pins this file against a reference recording with many ties (tests/golden/peaked_traj.*.npz).
    ids = input_ids.clone()
    assert ids.shape[0] == 1
    image_end = image_start + seq_len + seq_len // newline_every
    n_text_masked = int((ids[0, text_start:text_end] == MASK_TOKEN).sum())
    k_sched = get_num_transfer_tokens(n_text_masked, text_steps)
    img_steps = torch.linspace(text_steps // 4, text_steps - 1, timesteps).round().int().tolist()  # :157-159
    if image_step_list is not None:  # app.py:162-164 (Gradio sampler): linspace(0, text_steps-1, int(0.3*text_steps))
        img_steps = list(image_step_list)
    pos = [i for i in range(image_start, image_end) if int(ids[0, i]) != NEW_LINE]  # :164-169
    assert len(pos) == seq_len
    lo, hi = text_vocab_size, text_vocab_size + codebook_size

    def call(x):
        if trace is not None:
            trace.append(x.clone())
        return model_fn(x)

    for step in range(text_steps):
        cond = call(ids)  # :177-178
        if int((ids[0, text_start:text_end] == MASK_TOKEN).sum()) > 0:  # :183
            tl = cond[:, text_start:text_end, :].contiguous()
            noisy = None
            if text_temperature != 0:  # add_gumbel_noise :8-20 (bf16 tensor ops)
                u = torch.rand(tl.shape, dtype=tl.dtype, generator=generator)
                noisy = (tl + text_temperature * (-torch.log(-torch.log(u + 1e-10) + 1e-10))).contiguous()
            if remasking == "random":  # :194-198 with generator=None: the rank of a masked position is a uniform draw
                assert generator is None, "the reference raises here (torch.rand with dtype=int64, SURVEY A.6b)"
                x0 = torch.argmax(noisy if noisy is not None else tl, dim=-1)
                u = torch.rand((x0.shape[0], x0.shape[1]))
                masked = ids[:, text_start:text_end] == MASK_TOKEN
                conf = torch.where(masked, u, torch.tensor(-float("inf")))
                ids = ids.clone()
                if k_sched[step] > 0:
                    _, sel = torch.topk(conf[0], k=k_sched[step])  # :212
                    ids[0, text_start + sel] = x0[0, sel]
            else:
                ids, _, _ = so.text_select(tl, noisy, ids, text_start, [k_sched[step]])
        if step in img_steps:  # :220
            cond_vq = cond[:, pos, lo:hi]
            ut = ui = None
            if (cfg_scale > 0.0 and uncon_text is not None) or (cfg_img > 0.0 and uncon_image is not None):  # :243
                a, b = ids.clone(), ids.clone()
                if uncon_text is not None:
                    a[:, :uncon_text.shape[1]] = uncon_text
                if uncon_image is not None:
                    b[:, :uncon_image.shape[1]] = uncon_image
                ut, ui = call(a)[:, pos, lo:hi], call(b)[:, pos, lo:hi]
            else:
                ut = ui = torch.zeros_like(cond_vq)
            am, pm, probs = so.image_probs(cond_vq.contiguous(), ut.contiguous(), ui.contiguous(), cfg_scale, cfg_img,
                                           want_probs=temperature != 0)
            if temperature != 0:  # :297-302, then the probability of the drawn token (:311)
                s64 = torch.multinomial(probs.reshape(-1, codebook_size), 1, generator=generator)
                pm = torch.gather(probs.reshape(-1, codebook_size), -1, s64).view(1, seq_len)
                am = s64.view(1, seq_len).to(torch.int32)
            ratio = 1.0 * (step + 1) / text_steps
            mask_ratio = torch.cos(torch.tensor(ratio) * math.pi / 2)  # cosine_schedule :73-75
            mlen = int((seq_len * mask_ratio).floor().long())  # :321
            if temperature != 0:
                noise = torch.randn((1, seq_len), dtype=torch.bfloat16, generator=generator)  # :30-33
            else:
                if remasking == "random":  # keep the GLOBAL generator in step: the reference draws even at temperature 0 (A.2)
                    torch.randn((1, seq_len), dtype=torch.bfloat16)
                noise = torch.zeros((1, seq_len), dtype=torch.bfloat16)  # temperature 0: 0 * randn
            if tie_order == "torch":   # parallel_generator.py:304-344 + mask_by_random_topk :23-70 in torch ops
                cur = ids[0, pos]
                unknown = (cur == MASK_TOKEN)[None, :]
                sampled = torch.where(unknown, am.long(), (cur - text_vocab_size)[None, :]).clamp(0, codebook_size - 1)
                sel = torch.where(unknown, pm, torch.tensor(torch.finfo(torch.bfloat16).max, dtype=torch.bfloat16))
                k = max(1, min(int(unknown.sum()) - 1, mlen))
                conf = torch.log(sel + 1e-10) + (temperature * (1.0 - ratio)) * noise
                order = torch.sort(conf, dim=-1, descending=False)[1]
                masking = torch.zeros_like(sel, dtype=torch.bool)
                masking[0, order[0, :k]] = True
                ids = ids.clone()
                ids[0, pos] = torch.where(masking[0], torch.tensor(MASK_TOKEN), sampled[0] + text_vocab_size)
                if commit_trace is not None:
                    commit_trace.append((sel.clone(), masking.clone(), k))
            else:
                ids = so.image_commit(ids, pos, am, pm, noise, temperature * (1.0 - ratio), mlen, MASK_TOKEN, text_vocab_size,
                                      codebook_size)
    return ids
