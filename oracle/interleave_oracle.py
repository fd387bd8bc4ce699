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


def mmu_generate(model_fn: Callable, idx: torch.Tensor, max_new_tokens: int, steps: int, block_length: int,
                 temperature: float, cfg_scale: float, mask_id: int, rng=None, trace: Optional[list] = None,
                 eot_token=None, stop_on_eot: bool = False):
    """MMadaModelLM.mmu_generate / mmu_generate_fast (modeling_mmada.py:618-766), plain torch CPU ops in the reference's
    dtypes (bf16 CFG combine, float64 softmax / Gumbel-max, per-row topk).  Pinned by tests/golden/mmu_traj.npz."""
    import torch.nn.functional as F

    B, P = idx.shape
    x = torch.full((B, P + max_new_tokens), mask_id, dtype=torch.long)
    x[:, :P] = idx.clone()
    prompt_index = x != mask_id
    num_blocks = max_new_tokens // block_length
    steps = steps // num_blocks
    for nb in range(num_blocks):
        blk = x[:, P + nb * block_length:P + (nb + 1) * block_length] == mask_id
        ntt = [get_num_transfer_tokens(int(blk[j].sum()), steps) for j in range(B)]  # :63-81 per row
        for i in range(steps):
            mask_index = x == mask_id
            if cfg_scale > 0.0:  # :660-666
                un_x = x.clone()
                un_x[prompt_index] = mask_id
                both = torch.cat([x, un_x], dim=0)
                if trace is not None:
                    trace.append(both.clone())
                logits = model_fn(both)
                logits, un_logits = torch.chunk(logits, 2, dim=0)
                logits = un_logits + (cfg_scale + 1) * (logits - un_logits)
            else:
                if trace is not None:
                    trace.append(x.clone())
                logits = model_fn(x)
            if temperature == 0:  # add_gumbel_noise :49-60 + argmax
                x0 = torch.argmax(logits, dim=-1)
            else:
                x0 = rng.text_gumbel_argmax(logits, temperature)
            p = F.softmax(logits.to(torch.float64), dim=-1)
            x0_p = torch.squeeze(torch.gather(p, dim=-1, index=torch.unsqueeze(x0, -1)), -1)
            x0_p[:, P + (nb + 1) * block_length:] = -float("inf")  # :677
            x0 = torch.where(mask_index, x0, x)
            confidence = torch.where(mask_index, x0_p, -float("inf"))
            transfer = torch.zeros_like(x0, dtype=torch.bool)
            for j in range(B):
                _, sel = torch.topk(confidence[j], k=ntt[j][i])
                transfer[j, sel] = True
            x[transfer] = x0[transfer]
        if stop_on_eot and eot_token is not None:  # :756-761
            last = P + (nb + 1) * block_length - 1
            if last < x.shape[1] and bool(torch.all(x[:, last] == eot_token)):
                break
    return x


def t2i_generate(model_fn: Callable, input_ids: torch.Tensor, uncond_input_ids: Optional[torch.Tensor], temperature: float,
                 timesteps: int, guidance_scale: float, seq_len: int, mask_token_id: int, resolution: int, codebook_size: int,
                 tok_len: int, rng, generator=None, trace: Optional[list] = None):
    """MMadaModelLM.t2i_generate (modeling_mmada.py:264-359) + sampling.mask_by_random_topk (sampling.py:31-36), plain
    torch CPU ops in the reference's dtypes.  Pinned by tests/golden/m_t2i_traj.npz.  Mutates input_ids like the reference."""
    N = seq_len
    cur = input_ids[:, -(N + 1):-1].clone()
    cur = torch.where(cur == mask_token_id, mask_token_id, cur - tok_len)
    if uncond_input_ids is not None:
        uncond_prefix = uncond_input_ids[:, :resolution + 1]
    sampled_ids = None
    for step in range(timesteps):
        if uncond_input_ids is not None and guidance_scale > 0:
            uncond_input_ids = torch.cat([uncond_prefix, input_ids[:, resolution + 1:]], dim=1)
            model_input = torch.cat([input_ids, uncond_input_ids])
            if trace is not None:
                trace.append(model_input.clone())
            cond_logits, uncond_logits = torch.chunk(model_fn(model_input), 2, dim=0)
            logits = (1 + guidance_scale) * cond_logits - guidance_scale * uncond_logits
        else:
            if trace is not None:
                trace.append(input_ids.clone())
            logits = model_fn(input_ids)
        logits = logits[:, -(N + 1):-1, tok_len:tok_len + codebook_size]
        probs = logits.softmax(dim=-1)
        sampled_ids = rng.multinomial(probs.reshape(-1, logits.size(-1)), generator).view(*logits.shape[:-1])
        unknown_map = cur == mask_token_id
        sampled_ids = torch.where(unknown_map, sampled_ids, cur)
        ratio = 1.0 * (step + 1) / timesteps
        mask_ratio = torch.cos(torch.tensor(ratio) * math.pi * 0.5)
        selected_probs = torch.gather(probs, -1, sampled_ids.long()[..., None]).squeeze(-1)
        selected_probs = torch.where(unknown_map, selected_probs, torch.finfo(selected_probs.dtype).max)
        mask_len = (N * mask_ratio).floor().unsqueeze(0)
        mask_len = torch.max(torch.tensor([1]), torch.min(unknown_map.sum(dim=-1, keepdim=True) - 1, mask_len))
        temperature = temperature * (1.0 - ratio)
        confidence = _log(selected_probs) + temperature * (-_log(-_log(rng.uniform_like(selected_probs, generator))))
        cut_off = torch.gather(torch.sort(confidence, dim=-1).values, 1, mask_len.long())
        masking = confidence < cut_off
        input_ids[:, -(N + 1):-1] = torch.where(masking, mask_token_id, sampled_ids + tok_len)
        cur = torch.where(masking, mask_token_id, sampled_ids)
    return sampled_ids
