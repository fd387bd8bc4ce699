"""MI355X-native `t2i_generate` — the MMaDA-Parallel-M text-to-image MaskGIT sampler
(MMaDA-Parallel-M/models/modeling_mmada.py:264-359, models/sampling.py:10-36) on the HIP kernels.

The image tokens are the N positions before the last token of `input_ids` (:294).  Per step: one forward (cond and
uncond as ONE batch-2B forward when guidance_scale > 0, :300-312), `(1 + g) * cond - g * uncond` in bf16, bf16 softmax,
torch.multinomial draw, keep the already-known tokens, Gumbel-noised confidence and the cut-off re-mask with
mask_len = max(1, min(unknown - 1, floor(N * cos(ratio * pi / 2)))) (:343-352).  Reference quirks kept on purpose:
`temperature` is multiplied by (1 - ratio) cumulatively (:349), the caller's `input_ids` is updated in place (:353),
the returned ids are the last step's samples (:358), and the attention masks only feed an `attention_bias` that the
reference model ignores (see generators/mmu_generator.py) — attention is unmasked.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .. import abi
from ..model import LLaDAForMultiModalGeneration
from .interleave_generator import TorchRng, _log, cosine_schedule
from .parallel_generator import check_tp_exchange, mask_len_schedule


def _t2i_steps(
    model,
    input_ids: torch.LongTensor = None,
    uncond_input_ids: torch.LongTensor = None,
    attention_mask=None,
    uncond_attention_mask=None,
    temperature=1.0,
    timesteps=18,
    guidance_scale=0,
    noise_schedule: Callable = cosine_schedule,
    generator: torch.Generator = None,
    config=None,
    seq_len=1024,
    mask_token_id=126336,
    resolution=512,
    codebook_size=8192,
    rng=None,
    trace: Optional[list] = None,
    **kwargs,
):
    """Step generator shared by t2i_generate and t2i_generate_decoding_stepwise: yields (step, sampled_ids) right after
    the draw of every step (where the stepwise variant decodes, :841-848), then commits the step."""
    if not isinstance(model, LLaDAForMultiModalGeneration):
        raise TypeError("t2i_generate (MI355X) needs mmada_parallel_amd.LLaDAForMultiModalGeneration")
    if int(model.config.get("mask_token_id", 126336)) != mask_token_id:
        raise ValueError("mask_token_id differs from the model's")
    rng = rng or TorchRng()
    lib, h, device = model._lib, model._handle, model.device
    uni_prompting = kwargs.get("uni_prompting", None)
    tok_len = len(uni_prompting.text_tokenizer)
    N, CB = seq_len, codebook_size
    ids = input_ids.to(device).contiguous()
    B, L = ids.shape
    i0 = L - (N + 1)
    use_cfg = uncond_input_ids is not None and guidance_scale > 0
    if uncond_input_ids is not None:
        uncond_prefix = uncond_input_ids.to(device)[:, :resolution + 1]   # :298
        if use_cfg and uncond_prefix.shape[1] + max(0, L - (resolution + 1)) != L:
            raise ValueError("uncond_input_ids is shorter than resolution + 1 tokens")
    mlen_dev = torch.tensor(mask_len_schedule(N, timesteps, noise_schedule), dtype=torch.int32, device=device)
    pos_map = torch.arange(i0, i0 + N, dtype=torch.int32, device=device)
    rows1 = (torch.arange(B, dtype=torch.int32, device=device)[:, None] * L + pos_map[None, :]).reshape(-1)
    rows = torch.cat([rows1, rows1 + B * L]).contiguous() if use_cfg else rows1.contiguous()
    argmax = torch.empty((B, N), dtype=torch.int32, device=device)
    pmax = torch.empty((B, N), dtype=torch.bfloat16, device=device)
    probs = torch.empty((B * N, CB), dtype=torch.bfloat16, device=device)
    sampled_ids = None

    for step in range(timesteps):
        if use_cfg:
            unc_ids = torch.cat([uncond_prefix, ids[:, resolution + 1:]], dim=1)   # :303-304
            model_input = torch.cat([ids, unc_ids], dim=0).contiguous()
        else:
            model_input = ids
        if trace is not None:
            trace.append(model_input.cpu().clone())
        model.forward_body(model_input, consumed=(i0, i0 + N))   # only the image span is decoded
        il = model.head_rows(rows, tok_len, tok_len + CB)
        st = abi.stream_ptr()
        il_c, il_u = (il[:B * N], il[B * N:]) if use_cfg else (il, il)
        abi.check(lib.mmada_image_probs_m(h, il_c.data_ptr(), il_u.data_ptr(), B, N, CB,
                                          float(guidance_scale) if use_cfg else 0.0, probs.data_ptr(), argmax.data_ptr(),
                                          pmax.data_ptr(), st), "mmada_image_probs_m")
        drawn = rng.multinomial(probs, generator).view(B, N)                       # :319-320
        cur = ids[:, i0:i0 + N]
        unknown = cur == mask_token_id
        sampled_ids = torch.where(unknown, drawn, cur - tok_len)                   # :322-324
        yield step, sampled_ids
        ratio = 1.0 * (step + 1) / timesteps
        p_sel = torch.gather(probs.view(B, N, CB), -1, sampled_ids.long()[..., None]).squeeze(-1)
        p_sel = torch.where(unknown, p_sel, torch.finfo(p_sel.dtype).max)          # :334
        temperature = temperature * (1.0 - ratio)                                  # :349 (cumulative, as in the reference)
        gumbel = (-_log(-_log(rng.uniform_like(p_sel, generator)))).contiguous()   # sampling.py:14-16
        s32, p_c = sampled_ids.to(torch.int32).contiguous(), p_sel.contiguous()    # named: must outlive the launch
        abi.check(lib.mmada_image_commit_m(h, ids.data_ptr(), B, L, pos_map.data_ptr(), N, s32.data_ptr(), p_c.data_ptr(),
                                           gumbel.data_ptr(), float(temperature), mlen_dev[step:step + 1].data_ptr(),
                                           tok_len, st), "mmada_image_commit_m")
    if input_ids.data_ptr() != ids.data_ptr():
        input_ids.copy_(ids)                                                       # the reference mutates input_ids (:353)


@torch.no_grad()
def t2i_generate(model, input_ids=None, uncond_input_ids=None, attention_mask=None, uncond_attention_mask=None,
                 temperature=1.0, timesteps=18, guidance_scale=0, noise_schedule: Callable = cosine_schedule,
                 generator: torch.Generator = None, config=None, seq_len=1024, mask_token_id=126336, resolution=512,
                 codebook_size=8192, rng=None, trace: Optional[list] = None, **kwargs):
    """Returns sampled_ids [B, seq_len] (codebook ids) like the reference (:358); `input_ids` is updated in place."""
    sampled_ids = None
    for _step, sampled_ids in _t2i_steps(model, input_ids, uncond_input_ids, attention_mask, uncond_attention_mask,
                                         temperature, timesteps, guidance_scale, noise_schedule, generator, config, seq_len,
                                         mask_token_id, resolution, codebook_size, rng, trace, **kwargs):
        pass
    check_tp_exchange(model)   # tensor parallel: a timed-out hand-off raises instead of returning void tokens
    return sampled_ids


@torch.no_grad()
def t2i_generate_decoding_stepwise(model, input_ids=None, uncond_input_ids=None, attention_mask=None,
                                   uncond_attention_mask=None, temperature=1.0, timesteps=18, guidance_scale=0,
                                   noise_schedule: Callable = cosine_schedule, generator: torch.Generator = None, config=None,
                                   seq_len=1024, mask_token_id=126336, resolution=512, codebook_size=8192, vq_model=None,
                                   rng=None, **kwargs):
    """modeling_mmada.py:768-875: t2i_generate that decodes the current samples after every step and yields
    (PIL image of batch element 0, "Step i/T").  `vq_model` is anything with decode_code (mmada_parallel_amd.MAGVITv2)."""
    from PIL import Image

    for step, sampled_ids in _t2i_steps(model, input_ids, uncond_input_ids, attention_mask, uncond_attention_mask,
                                        temperature, timesteps, guidance_scale, noise_schedule, generator, config, seq_len,
                                        mask_token_id, resolution, codebook_size, rng, None, **kwargs):
        if step == timesteps - 1:
            check_tp_exchange(model)
        cur = torch.clamp(sampled_ids.clone(), 0, 8192 - 1)                         # :839-840 (constant as in the reference)
        images = torch.clamp((vq_model.decode_code(cur) + 1.0) / 2.0, min=0.0, max=1.0) * 255.0
        images = images.permute(0, 2, 3, 1).cpu().numpy().astype("uint8")
        yield Image.fromarray(images[0]), f"Step {step + 1}/{timesteps}"
