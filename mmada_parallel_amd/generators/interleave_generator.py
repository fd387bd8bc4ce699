"""MI355X-native `interleave_generate` — the MMaDA-Parallel-M sampler
(MMaDA-Parallel-M/models/modeling_mmada.py:117-248, models/sampling.py:31-36) on the same HIP kernels.

Differences from the A sampler that matter: cond and uncond run as ONE batch-2 forward every step;
text logits are CFG-combined (cond + text_cfg*(uncond-cond)); image logits are (1+cfg)*cond - cfg*uncond; the image
token is ALWAYS drawn with torch.multinomial; the re-mask uses Gumbel noise and a cut-off compare; no newline tokens
inside the image span; the returned image ids are the last image step's samples.

The random draws go through an `rng` object (default: torch's own RNG calls, in the reference's order) so that parity
tests can replay the reference's draws.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch

from .. import abi
from ..model import LLaDAForMultiModalGeneration
from .parallel_generator import check_tp_exchange, mask_len_schedule


def cosine_schedule(t):
    """models/sampling.py:39-40."""
    return torch.cos(t * math.pi * 0.5)


def get_num_transfer_tokens(mask_index: torch.Tensor, steps: int) -> torch.Tensor:
    """modeling_mmada.py:63-81: base + remainder schedule (differs from the A sampler's)."""
    mask_num = mask_index.sum(dim=1, keepdim=True)
    base = mask_num // steps
    remainder = mask_num % steps
    out = torch.zeros(mask_num.size(0), steps, dtype=torch.int64) + base
    for i in range(mask_num.size(0)):
        out[i, :remainder[i]] += 1
    return out


class TorchRng:
    """The reference's random draws, call for call (device RNG)."""

    def text_gumbel_argmax(self, text_logits: torch.Tensor, temperature: float) -> torch.Tensor:
        # add_gumbel_noise (:49-60) + argmax (:182), float64
        l64 = text_logits.to(torch.float64)
        noise = torch.rand_like(l64, dtype=torch.float64)
        return torch.argmax(l64.exp() / ((-torch.log(noise)) ** temperature), dim=-1)

    def rand_f64(self, shape, device) -> torch.Tensor:
        return torch.rand(tuple(shape), dtype=torch.float64, device=device)  # torch.rand_like(logits, dtype=float64)

    def multinomial(self, probs2d: torch.Tensor, generator) -> torch.Tensor:
        return torch.multinomial(probs2d, 1, generator=generator)[:, 0]  # :220-222

    def uniform_like(self, t: torch.Tensor, generator) -> torch.Tensor:
        return torch.zeros_like(t).uniform_(0, 1, generator=generator)  # sampling.py:15-16


def _log(t, eps=1e-20):
    return torch.log(t.clamp(min=eps))  # sampling.py:11-12


@torch.no_grad()
def interleave_generate(
    model,
    input_ids: torch.LongTensor = None,
    uncond_input_ids: torch.LongTensor = None,
    text_cfg: float = 0.0,
    image_cfg: float = 3.5,
    noise_schedule: Callable = cosine_schedule,
    text_steps: int = 100,
    image_steps: int = 100,
    reserved_token_mapping: Dict = None,
    generator: torch.Generator = None,
    config=None,
    remasking="low_confidence",
    text_temperature: float = 0.0,
    image_temperature: float = 1.0,
    rng=None,
    trace: Optional[list] = None,
    **kwargs,
):
    """Returns (image ids [1, num_vq_tokens], text ids [1, max_seq_length]) like the reference."""
    if not isinstance(model, LLaDAForMultiModalGeneration):
        raise TypeError("interleave_generate (MI355X) needs mmada_parallel_amd.LLaDAForMultiModalGeneration")
    if remasking != "low_confidence":
        raise NotImplementedError(remasking)
    if not (text_cfg or image_cfg):
        raise ValueError("text_cfg and image_cfg cannot be both 0")  # modeling_mmada.py:176-177
    rng = rng or TorchRng()
    lib, h, device = model._lib, model._handle, model.device
    uni_prompting = kwargs.get("uni_prompting", None)
    tok = uni_prompting.text_tokenizer
    text_vocab = len(tok)
    mask_id = int(model.config.get("mask_token_id", 126336))
    N = config.model.mmada.num_vq_tokens
    CB = config.model.mmada.codebook_size
    T = config.dataset.preprocessing.max_seq_length
    V = model.vocab

    input_ids = input_ids.to(device).unsqueeze(0)
    uncond_input_ids = uncond_input_ids.to(device).unsqueeze(0)
    P = input_ids.shape[1]

    def full(n, v):
        return torch.full((1, n), v, dtype=torch.long, device=device)

    out_ids = torch.cat([full(1, reserved_token_mapping['<|soi|>']), full(N, mask_id),
                         full(1, reserved_token_mapping['<|eoi|>']), full(1, tok.bos_token_id), full(T - 1, mask_id)], dim=1)
    ids = torch.cat([input_ids, out_ids], dim=1).contiguous()  # combined_input_ids (:144)
    L = ids.shape[1]
    Lu = uncond_input_ids.shape[1] + out_ids.shape[1]
    if Lu != L:
        raise ValueError("cond and uncond prompts must have equal length (the reference batches them, :171)")
    text_start, img_start = L - T, P + 1

    text_masked0 = (ids[:, text_start:] == mask_id).cpu()
    num_transfer = get_num_transfer_tokens(text_masked0, text_steps)
    img_steps = set(torch.linspace(text_steps // 4, text_steps - 1, image_steps).round().int().tolist())
    mlen = mask_len_schedule(N, text_steps, noise_schedule)
    k_dev = num_transfer.t().contiguous().to(device=device, dtype=torch.int32)
    mlen_dev = torch.tensor(mlen, dtype=torch.int32, device=device)
    pos_map = torch.arange(img_start, img_start + N, dtype=torch.int32, device=device)
    ar = torch.arange(text_start, L, dtype=torch.int32, device=device)
    text_rows = torch.cat([ar, ar + L])                       # cond rows then uncond rows (batch-major)
    img_rows = torch.cat([pos_map, pos_map + L])
    scratch = torch.empty(T * 16, dtype=torch.uint8, device=device)
    argmax = torch.empty((1, N), dtype=torch.int32, device=device)
    pmax = torch.empty((1, N), dtype=torch.bfloat16, device=device)
    sampled_ids = None

    for i in range(text_steps):
        unc = torch.cat([uncond_input_ids, ids[:, P:]], dim=1)  # :166-169
        both = torch.cat([ids, unc], dim=0).contiguous()
        if trace is not None:
            trace.append(both.cpu().clone())
        is_img = i in img_steps
        # rows read this step: the text span, plus the image span on image steps (the last block computes only those)
        model.forward_body(both, consumed=(img_start if is_img else text_start, L))   # one batch-2 forward (:171)
        st = abi.stream_ptr()
        tl = model.head_rows(text_rows, 0, V)                    # [2T, V]: cond rows, uncond rows
        il = model.head_rows(img_rows, text_vocab, text_vocab + CB) if is_img else None

        x0_in = None
        if text_temperature != 0:
            comb = tl[:T] + text_cfg * (tl[T:] - tl[:T])         # :173 on the text rows, bf16 tensor ops
            x0_in = rng.text_gumbel_argmax(comb.view(1, T, V), text_temperature).to(torch.int32).contiguous()
        tl_c, tl_u = tl[:T], tl[T:]  # views of `tl`
        abi.check(lib.mmada_text_select_cfg(h, tl_c.data_ptr(), tl_u.data_ptr(), float(text_cfg), abi.ptr(x0_in), 1, T,
                                            V, V, ids.data_ptr(), L, text_start, k_dev[i].data_ptr(), scratch.data_ptr(),
                                            st), "mmada_text_select_cfg")

        if is_img:
            probs = torch.empty((N, CB), dtype=torch.bfloat16, device=device)
            il_c, il_u = il[:N], il[N:]
            abi.check(lib.mmada_image_probs_m(h, il_c.data_ptr(), il_u.data_ptr(), 1, N, CB, float(image_cfg),
                                              probs.data_ptr(), argmax.data_ptr(), pmax.data_ptr(), st),
                      "mmada_image_probs_m")
            drawn = rng.multinomial(probs, generator).view(1, N)
            cur = ids[:, img_start:img_start + N]
            unknown = cur == mask_id
            sampled_ids = torch.where(unknown, drawn, cur - text_vocab)                       # :224-225
            p_sel = torch.gather(probs.view(1, N, CB), -1, sampled_ids.long()[..., None]).squeeze(-1)
            p_sel = torch.where(unknown, p_sel, torch.finfo(p_sel.dtype).max)                 # :233
            ratio = 1.0 * (i + 1) / text_steps
            temperature = image_temperature * (1.0 - ratio)
            gumbel = (-_log(-_log(rng.uniform_like(p_sel, generator)))).contiguous()          # sampling.py:15-17
            s32, p_c = sampled_ids.to(torch.int32).contiguous(), p_sel.contiguous()  # named: must outlive the launch
            abi.check(lib.mmada_image_commit_m(h, ids.data_ptr(), 1, L, pos_map.data_ptr(), N, s32.data_ptr(),
                                               p_c.data_ptr(), gumbel.data_ptr(), float(temperature),
                                               mlen_dev[i:i + 1].data_ptr(), text_vocab, st), "mmada_image_commit_m")

    check_tp_exchange(model)   # tensor parallel: a timed-out hand-off raises instead of returning void tokens
    return sampled_ids, ids[:, text_start:]
