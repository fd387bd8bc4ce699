"""CPU ORACLE (test infrastructure only): restatement of MMadaModelLM.interleave_generate
(/root/reference/MMaDA-Parallel-M/models/modeling_mmada.py:117-248) on top of the C sampler oracle.

`model_fn(ids[2,L]) -> logits [2,L,V] bf16` stands for `self(torch.cat([cond, uncond])).logits` (:171).
Random draws come from `rng` (multinomial / uniform_like / text_gumbel_argmax) so tests can replay the reference's.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch

from . import sampler_oracle as so


_REAL_MULTINOMIAL, _REAL_UNIFORM, _REAL_RAND = torch.multinomial, torch.Tensor.uniform_, torch.rand


class SeededRng:
    """Random draws of the M sampler from per-call seeded CPU generators: used to DRIVE the reference when the
    fixtures are generated (oracle/gen_golden.py patches torch.multinomial / Tensor.uniform_ / torch.rand_like with
    it) and to replay the same draws in the oracle and GPU tests."""

    def __init__(self, seed):
        self.seed, self.n = seed, 0

    def _g(self):
        self.n += 1
        return torch.Generator().manual_seed(self.seed * 7919 + self.n)

    def multinomial(self, probs2d, generator=None):
        return _REAL_MULTINOMIAL(probs2d.detach().cpu(), 1, generator=self._g())[:, 0].to(probs2d.device)

    def uniform_like(self, t, generator=None):
        return _REAL_UNIFORM(torch.zeros(t.shape, dtype=t.dtype), 0, 1, generator=self._g()).to(t.device)

    def rand_f64(self, shape, device="cpu"):
        return _REAL_RAND(shape, dtype=torch.float64, generator=self._g()).to(device)

    def text_gumbel_argmax(self, text_logits, temperature):
        l64 = text_logits.to(torch.float64)
        noise = self.rand_f64(l64.shape, l64.device)
        return torch.argmax(l64.exp() / ((-torch.log(noise)) ** temperature), dim=-1)


def _log(t, eps=1e-20):
    return torch.log(t.clamp(min=eps))  # models/sampling.py:11-12


def get_num_transfer_tokens(n_masked: int, steps: int):
    base, rem = n_masked // steps, n_masked % steps  # modeling_mmada.py:63-81
    return [base + (1 if s < rem else 0) for s in range(steps)]


def generate(model_fn: Callable, input_ids: torch.Tensor, uncond_input_ids: torch.Tensor, text_cfg: float,
             image_cfg: float, text_steps: int, image_steps: int, soi: int, eoi: int, bos: int, mask_id: int,
             text_vocab: int, num_vq_tokens: int, codebook_size: int, max_seq_length: int, image_temperature: float,
             rng, generator=None, text_temperature: float = 0.0, trace: Optional[list] = None):
    N, T, CB = num_vq_tokens, max_seq_length, codebook_size
    inp, unc_in = input_ids.unsqueeze(0), uncond_input_ids.unsqueeze(0)
    P = inp.shape[1]
    full = lambda n, v: torch.full((1, n), v, dtype=torch.long)  # noqa: E731
    ids = torch.cat([inp, full(1, soi), full(N, mask_id), full(1, eoi), full(1, bos), full(T - 1, mask_id)], dim=1)  # :133-144
    L = ids.shape[1]
    ts, i0 = L - T, P + 1
    k_sched = get_num_transfer_tokens(int((ids[0, ts:] == mask_id).sum()), text_steps)
    img_steps = torch.linspace(text_steps // 4, text_steps - 1, image_steps).round().int().tolist()  # :154
    pos = list(range(i0, i0 + N))
    sampled_ids = None
    for i in range(text_steps):
        unc = torch.cat([unc_in, ids[:, P:]], dim=1)  # :166-169
        both = torch.cat([ids, unc], dim=0)
        if trace is not None:
            trace.append(both.clone())
        logits = model_fn(both)
        cond, uncond = logits[0:1], logits[1:2]
        x0_in = None
        if text_temperature != 0:
            comb = cond[:, ts:] + text_cfg * (uncond[:, ts:] - cond[:, ts:])
            x0_in = rng.text_gumbel_argmax(comb, text_temperature).to(torch.int32)
        img_c = cond[:, i0:i0 + N, text_vocab:text_vocab + CB].contiguous()      # taken before the text update, like :216
        img_u = uncond[:, i0:i0 + N, text_vocab:text_vocab + CB].contiguous()
        ids, _, _ = so.text_select_cfg(cond[:, ts:].contiguous(), uncond[:, ts:].contiguous(), text_cfg, ids, ts,
                                       [k_sched[i]], x0_in=x0_in, mask_id=mask_id)
        if i in img_steps:  # :209
            _, _, probs = so.image_probs_m(img_c, img_u, image_cfg)
            drawn = rng.multinomial(probs.view(N, CB), generator).view(1, N)
            cur = ids[:, i0:i0 + N]
            unknown = cur == mask_id
            sampled_ids = torch.where(unknown, drawn, cur - text_vocab)
            p_sel = torch.gather(probs, -1, sampled_ids.long()[..., None]).squeeze(-1)
            p_sel = torch.where(unknown, p_sel, torch.finfo(p_sel.dtype).max)
            ratio = 1.0 * (i + 1) / text_steps
            mlen = int((N * torch.cos(torch.tensor(ratio) * math.pi * 0.5)).floor().long())
            temperature = image_temperature * (1.0 - ratio)
            gumbel = -_log(-_log(rng.uniform_like(p_sel, generator)))
            ids = so.image_commit_m(ids, pos, sampled_ids.to(torch.int32), p_sel, gumbel, temperature, mlen, mask_id,
                                    text_vocab)
    return sampled_ids, ids[:, ts:]
