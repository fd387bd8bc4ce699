"""MI355X-native `generate_image` — the MMaDA-Parallel-A text-to-image MaskGIT sampler
(MMaDA-Parallel-A/generators/image_generation_generator.py:14-251, utils/generation_utils.py:28-64) on the HIP kernels.

Same signature and return value as the reference.  What the loop does per step (B = 1, like the reference :55):
conditional forward (+ an unconditional forward of a different length when cfg_scale > 0, :120-152), logits of the
still-masked image slots over the codebook columns, `(1 + cfg) * cond - cfg * uncond` in bf16, Gumbel-max / argmax
sample, bf16 softmax confidence of the sampled token, write the samples, re-mask the `keep_n` least confident ones
(cut-off compare on log p + temperature * gumbel).  The reference's `use_cache` branch never hands a compute mask to the
model (:128,141 pass only `use_cache`), so its K/V "cache" is rewritten in full by every call and does not change the
arithmetic; the argument is accepted and has no effect here either.

Random draws (temperature > 0) go through an `rng` object (default: torch's RNG calls in the reference's order) so that
parity tests can replay the reference's draws; at temperature 0 the path is deterministic and bit-exact.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch

from .. import abi
from ..model import LLaDAForMultiModalGeneration
from .parallel_generator import check_tp_exchange


def cosine_schedule(t: torch.Tensor) -> torch.Tensor:
    """utils/generation_utils.py:28-30."""
    return torch.cos(0.5 * math.pi * t)


class TorchRng:
    """The reference's draws, call for call: gumbel_noise (utils/generation_utils.py:33-39) uses torch.rand in the
    tensor's dtype (bf16) on its device."""

    def rand(self, shape, dtype, device, generator):
        if generator is None:
            return torch.rand(shape, dtype=dtype, device=device)
        return torch.rand(shape, device=device, dtype=dtype, generator=generator)


def _gumbel(u: torch.Tensor) -> torch.Tensor:
    return -torch.log(-torch.log(u + 1e-20) + 1e-20)  # utils/generation_utils.py:39, tensor ops in u's dtype


def keep_schedule(vq_len: int, timesteps: int, noise_schedule: Callable = cosine_schedule):
    """keep_n of every step (:96-103): floor(vq_len * schedule((step+1)/T)) clamped to >= 1; 0 on the last step."""
    out = []
    for step in range(timesteps):
        if step < timesteps - 1:
            frac = noise_schedule(torch.tensor([(step + 1) / timesteps]))
            out.append(int((torch.tensor([[vq_len]]).float() * frac).floor().clamp_min(1).long().item()))
        else:
            out.append(0)
    return out


@torch.no_grad()
def generate_image(
    model,
    prompt: torch.LongTensor,
    *,
    seq_len: int = 1024,
    newline_every: int = 16,
    timesteps: int = 18,
    mask_token_id: int = 126336,
    newline_id: int = 126084,
    temperature: float = 1.0,
    cfg_scale: float = 0.0,
    uncon_ids: torch.LongTensor = None,
    code_start: Optional[int] = None,
    codebook_size: int = 8192,
    noise_schedule: Callable[[torch.Tensor], torch.Tensor] = cosine_schedule,
    text_vocab_size: Optional[int] = None,
    generator: Optional[torch.Generator] = None,
    use_cache=False,
    cache_ratio=0.9,
    refresh_interval=5,
    warmup_ratio=0.3,
    debug: bool = False,
    debug_log_dir: Optional[str] = None,
    max_print_tokens: int = 100,
    rng=None,
    trace: Optional[list] = None,
) -> torch.LongTensor:
    """Returns vq_ids [1, seq_len] (token ids in the full vocabulary, newlines removed), like the reference :239-250.
    `debug` only prints one line per step (the reference's per-step dumps are host-side diagnostics)."""
    if not isinstance(model, LLaDAForMultiModalGeneration):
        raise TypeError("generate_image (MI355X) needs mmada_parallel_amd.LLaDAForMultiModalGeneration")
    if temperature > 8.0:
        raise ValueError("temperature > 8 is not supported (known slots must out-rank every noisy confidence)")
    rng = rng or TorchRng()
    lib, h, device = model._lib, model._handle, model.device
    prompt = prompt.to(device)
    B, L = prompt.shape
    assert B == 1, "batch>1 not supported – wrap in loop if needed"  # :55
    if int(model.config.get("mask_token_id", 126336)) != mask_token_id:
        raise ValueError("mask_token_id differs from the model's")
    x = prompt.clone().contiguous()
    slots = (x[0] == mask_token_id).nonzero(as_tuple=False)[:, 0]      # the image slots: every initially masked position
    N = int(slots.numel())
    vq_len = N
    if text_vocab_size is None:
        text_vocab_size = model.vocab - codebook_size                   # :78-82 (size of the logits' last dimension)
    off, CB = text_vocab_size, codebook_size
    use_cfg = cfg_scale > 0
    if use_cfg:
        if uncon_ids is None or code_start is None:
            raise ValueError("cfg_scale > 0 needs uncon_ids and code_start")
        uncon_ids = uncon_ids.to(device)
        U = uncon_ids.shape[1]
        if bool((slots < code_start - 2).any()):
            raise ValueError("masked tokens before code_start - 2 are dropped from the unconditional sequence")
    keep = torch.tensor(keep_schedule(vq_len, timesteps, noise_schedule), dtype=torch.int32, device=device)
    win = (int(slots[0]), int(slots[-1]) + 1) if N else None   # the only rows ever decoded (one host read, before the loop)
    pos_map = slots.to(torch.int32).contiguous()
    argmax = torch.empty((1, N), dtype=torch.int32, device=device)
    pmax = torch.empty((1, N), dtype=torch.bfloat16, device=device)
    probs = torch.empty((N, CB), dtype=torch.bfloat16, device=device)
    zeros_g = torch.zeros((1, N), dtype=torch.bfloat16, device=device)

    for step in range(timesteps):
        masked = x[0, slots.long()] == mask_token_id                   # [N] over the slots
        n_unknown = int(masked.sum().item())
        if n_unknown == 0:                                             # :91-94
            break
        rows = slots[masked].to(torch.int32).contiguous()              # flat_idx (:178), ascending positions
        if trace is not None:
            trace.append(x.cpu().clone())
        model.forward_body(x, consumed=win)
        cond = model.head_rows(rows, off, off + CB)                    # cond_logits[vq_mask] (:130-133) / :155-157
        if use_cfg:
            uncond_ids = torch.cat((uncon_ids, x[:, code_start - 2:]), dim=1).contiguous()   # :123
            if trace is not None:
                trace.append(uncond_ids.cpu().clone())
            model.forward_body(uncond_ids, consumed=(win[0] - (code_start - 2) + U, win[1] - (code_start - 2) + U))
            urows = (rows - (code_start - 2) + U).contiguous()         # uncond_vq_mask (:124)
            unc = model.head_rows(urows, off, off + CB)
        else:
            unc = cond
        st = abi.stream_ptr()
        # (1 + cfg) * cond - cfg * uncond, softmax (bf16), first-index argmax and its probability (:152,165-167)
        abi.check(lib.mmada_image_probs_m(h, cond.data_ptr(), unc.data_ptr(), 1, n_unknown, CB,
                                          float(cfg_scale) if use_cfg else 0.0, probs.data_ptr(), argmax.data_ptr(),
                                          pmax.data_ptr(), st), "mmada_image_probs_m")
        if temperature == 0.0:                                         # gumbel_max_sample, tau = 0 (:44-45)
            sampled = argmax[0, :n_unknown].long()
            conf = pmax[0, :n_unknown]
        else:
            logits = ((1 + cfg_scale) * cond - cfg_scale * unc) if use_cfg else cond          # bf16 tensor ops (:152)
            g = _gumbel(rng.rand(logits.shape, logits.dtype, device, generator))
            sampled = (logits / temperature + g).argmax(dim=-1)
            conf = probs[:n_unknown].gather(-1, sampled.unsqueeze(-1)).squeeze(-1)
        # gumbel of mask_by_random_topk is drawn even at temperature 0 (utils/generation_utils.py:57): keep the RNG in step
        # (always drawn, also with generator=None — the reference's rand_like advances the global RNG either way)
        gm = _gumbel(rng.rand((1, n_unknown), torch.bfloat16, device, generator))
        # scatter the per-masked-slot results back to slot order for the commit kernel
        s_full = torch.zeros((1, N), dtype=torch.int32, device=device)
        p_full = torch.zeros((1, N), dtype=torch.bfloat16, device=device)
        g_full = zeros_g if gm is None else torch.zeros((1, N), dtype=torch.bfloat16, device=device)
        s_full[0, masked] = sampled.to(torch.int32)
        p_full[0, masked] = conf
        if gm is not None:
            g_full[0, masked] = gm[0]
        abi.check(lib.mmada_image_commit_g(h, x.data_ptr(), 1, L, pos_map.data_ptr(), N, s_full.data_ptr(), p_full.data_ptr(),
                                           g_full.data_ptr(), float(temperature), keep[step:step + 1].data_ptr(), off, st),
                  "mmada_image_commit_g")
        if debug:
            print(f"[generate_image] step {step}: unknown {n_unknown} -> keep {int(keep[step])}")

    check_tp_exchange(model)   # tensor parallel: a timed-out hand-off raises instead of returning void tokens
    vq_ids = x[0, code_start:-2]                                       # :239-241
    return vq_ids[vq_ids != newline_id].view(1, seq_len)
