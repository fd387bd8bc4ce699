"""CPU ORACLE (test infrastructure only): restatement of the MMaDA-Parallel-A text-to-image MaskGIT sampler
    generate_image            /root/reference/MMaDA-Parallel-A/generators/image_generation_generator.py:14-251
    gumbel_noise / gumbel_max_sample / mask_by_random_topk / cosine_schedule
                              /root/reference/MMaDA-Parallel-A/utils/generation_utils.py:28-64
with plain torch CPU tensor ops in the reference's dtypes (bf16 logits -> bf16 softmax / log / compare), the sort taken
over the currently masked tokens only, exactly as the reference does.  `model_fn(ids) -> logits [1, L, V] bf16` stands
for `model(ids, infer=True).logits`; every call's ids are appended to `trace`.
Parity is PINNED: tests/test_oracle_golden.py compares the ids of every model call and the returned vq ids with
tests/golden/t2i_traj.npz, recorded from the reference's own generate_image (oracle/gen_golden.py: gen_t2i_traj).
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch


def cosine_schedule(t):  # utils/generation_utils.py:28-30
    return torch.cos(0.5 * math.pi * t)


def gumbel_noise(t, generator=None):  # :33-39
    u = torch.rand_like(t) if generator is None else torch.rand(t.shape, dtype=t.dtype, generator=generator)
    return -torch.log(-torch.log(u + 1e-20) + 1e-20)


def generate(model_fn: Callable, prompt: torch.Tensor, seq_len: int, timesteps: int, temperature: float, cfg_scale: float,
             uncon_ids: Optional[torch.Tensor], code_start: int, codebook_size: int, text_vocab_size: int,
             mask_token_id: int = 126336, newline_id: int = 126084, generator=None, trace: Optional[list] = None):
    x = prompt.clone()
    B = x.shape[0]
    assert B == 1
    vq_mask = x == mask_token_id
    unknown_cnt = vq_mask.sum(dim=1, keepdim=True)
    vq_len = unknown_cnt
    off = text_vocab_size

    def call(ids):
        if trace is not None:
            trace.append(ids.clone())
        return model_fn(ids)

    for step in range(timesteps):
        if unknown_cnt.item() == 0:  # :91
            break
        if step < timesteps - 1:  # :97-103
            frac = cosine_schedule(torch.tensor([(step + 1) / timesteps]))
            keep_n = (vq_len.float() * frac).floor().clamp_min(1).long()
        else:
            keep_n = torch.zeros_like(unknown_cnt)
        if cfg_scale > 0:  # :121-152
            uncond = torch.cat((uncon_ids, x[:, code_start - 2:]), dim=1)
            uncond_vq_mask = torch.cat((torch.zeros((1, uncon_ids.size(1)), dtype=torch.bool), vq_mask[:, code_start - 2:]), dim=1)
            cond_logits = call(x)[..., off:off + codebook_size]
            cond_mask_logits = cond_logits[vq_mask].view(B, -1, codebook_size)
            uncond_logits = call(uncond)[..., off:off + codebook_size]
            uncond_mask_logits = uncond_logits[uncond_vq_mask].view(B, -1, codebook_size)
            logits = (1 + cfg_scale) * cond_mask_logits - cfg_scale * uncond_mask_logits
        else:  # :154-157
            logits = call(x)[:, vq_mask[0], off:off + codebook_size]
        if temperature == 0.0:  # gumbel_max_sample :42-47
            sampled = logits.argmax(dim=-1)
        else:
            sampled = (logits / temperature + gumbel_noise(logits, generator)).argmax(dim=-1)
        sampled_full = sampled + off
        probs = torch.softmax(logits, dim=-1)  # :166
        conf = probs.gather(-1, sampled.unsqueeze(-1)).squeeze(-1)
        flat_idx = vq_mask.nonzero(as_tuple=False)[:, 1]  # :178
        x.view(-1)[flat_idx] = sampled_full.view(-1)
        # mask_by_random_topk :50-64
        g = gumbel_noise(conf, generator)
        confidence = torch.log(conf.clamp_min(1e-20)) + temperature * g
        sorted_conf = torch.sort(confidence, dim=-1).values
        k = keep_n.squeeze(1).long().unsqueeze(1).clamp_(0, conf.size(1) - 1)
        cut_off = torch.gather(sorted_conf, 1, k)
        mask_sel = confidence < cut_off
        x.view(-1)[flat_idx[mask_sel.view(-1)]] = mask_token_id  # :207
        vq_mask = x == mask_token_id
        unknown_cnt = vq_mask.sum(dim=1, keepdim=True)
    vq_ids = x[0, code_start:-2]  # :239-241
    return vq_ids[vq_ids != newline_id].view(1, seq_len)
