"""MI355X-native `generate_ti2ti`: same signature and results as the reference sampler
(generators/parallel_generator.py:102-368), with the tensor math on HIP kernels and no host sync inside the loop.

Control flow (step schedule, which forwards run, CFG prefix overwrite, final read-out incl. the one random fill) is
the reference's, line for line in meaning; what changed is *where* the math runs:
  * model forward            -> libmmada_mi355x (mmada_forward_body), LM head only on the rows/columns consumed
  * text argmax/softmax/topk -> mmada_text_select        (reference :181-217)
  * VQ gather + CFG + softmax + argmax -> mmada_head_rows + mmada_image_probs   (:236-295)
  * keep-known / confidence / re-mask / write-back -> mmada_image_commit        (:221-233, :304-344, :23-70)
All per-step counts (text k, image mask_len) are schedule-determined and precomputed on the host.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
from typing import List

import torch

from .. import abi
from ..model import LLaDAForMultiModalGeneration

MASK_TOKEN = 126336
NEW_LINE = 126084


def cosine_schedule(t):
    """Cosine noise schedule (reference :73-75; inference.py:37-38)."""
    return torch.cos(t * math.pi / 2)


def get_num_transfer_tokens(text_masked_indices: torch.Tensor, text_steps: int) -> torch.Tensor:
    """Tokens to unmask per step (reference :78-99): remaining - int(total * (1 - (s+1)/S)), Python float math."""
    batch_size = text_masked_indices.shape[0]
    initial_masks = text_masked_indices.sum(dim=1).tolist()
    num_transfer = torch.zeros(batch_size, text_steps, dtype=torch.long)
    for b in range(batch_size):
        total = int(initial_masks[b])
        remaining = total
        for step in range(text_steps):
            ratio = (step + 1) / text_steps
            target_remaining = int(total * (1 - ratio))
            n = max(0, remaining - target_remaining)
            num_transfer[b, step] = n
            remaining -= n
    return num_transfer


def image_step_indices(text_steps: int, timesteps: int) -> List[int]:
    """Steps at which an image denoising step runs (reference :157-159); duplicates collapse via `in`."""
    return torch.linspace(text_steps // 4, text_steps - 1, timesteps).round().int().tolist()


def mask_len_schedule(num_vq_tokens: int, text_steps: int, noise_schedule=cosine_schedule) -> List[int]:
    """floor(N * noise_schedule(ratio)) per step as int (reference :318-321); fp32 on the host like the CPU oracle.

    Note cos(pi/2) is -4.37e-8 in fp32, so the last step yields -1."""
    out = []
    for step in range(text_steps):
        ratio = 1.0 * (step + 1) / text_steps
        mask_ratio = noise_schedule(torch.tensor(ratio))
        out.append(int((num_vq_tokens * mask_ratio).floor().long().item()))
    return out


def check_tp_exchange(model):
    """A hand-off of the tensor-parallel pull transport that timed out (a lost or stalled peer) sets a sticky device flag
    and the ranks run on unsynchronised; nothing else would report it.  Called where the sampler synchronises anyway
    (the read-out of the final ids): raises instead of returning an image computed from stale partial sums."""
    if getattr(model, "_comm_in_library", False) and hasattr(model, "comm_status"):
        st = model.comm_status()
        if st["mode"] == "no-exchange diagnostic":
            raise abi.MmadaError("tensor-parallel exchange: the no-exchange diagnostic (mmada_comm_set_mode 3) is still on; "
                                 "the generated tokens are void")
        if st["error"]:
            raise abi.MmadaError(f"tensor-parallel exchange: hand-off timed out waiting for rank {st['error'] - 1} "
                                 f"(transport {st['mode']}); the generated tokens are void")


class TorchRng:
    """The reference's random draws, call for call, from torch's RNG on the tensors' device (:13-16, :30-33, :297-302).
    Parity tests pass an object with the same three methods that replays the draws of a recorded reference run."""

    def rand(self, shape, dtype, device, generator):
        if generator is not None:
            return torch.rand(shape, dtype=dtype, device=device, generator=generator)
        return torch.rand(shape, dtype=dtype, device=device)

    def randn(self, shape, dtype, device, generator):
        if generator is not None:
            return torch.randn(shape, dtype=dtype, device=device, generator=generator)
        return torch.randn(shape, dtype=dtype, device=device)

    def multinomial(self, probs2d, generator):
        if generator is not None:
            return torch.multinomial(probs2d, 1, generator=generator)
        return torch.multinomial(probs2d, 1)


def add_gumbel_noise(logits, temperature=1.0, generator=None, rng=None):
    """Reference :8-20 verbatim in meaning (bf16 noise, torch RNG) — torch-ROCm plumbing for text_temperature > 0."""
    if temperature == 0:
        return logits
    u = (rng or TorchRng()).rand(logits.shape, logits.dtype, logits.device, generator)
    g = -torch.log(-torch.log(u + 1e-10) + 1e-10)
    return logits + temperature * g


def _ti2ti_steps(
    model,
    input_ids,
    text_start,
    text_end,
    image_start,
    seq_len,
    newline_every,
    text_steps=100,
    text_gen_length=256,
    text_block_length=64,
    timesteps=100,
    temperature=1.0,
    text_temperature=0.7,
    cfg_scale=0.0,
    cfg_img=4.0,
    uncon_text=None,
    uncon_image=None,
    tokenizer=None,
    remasking='low_confidence',
    noise_schedule=cosine_schedule,
    generator=None,
    text_vocab_size=126356,
    codebook_size=8192,
    image_step_list=None,
    rng=None,
    graph=None,
):
    """Generator core shared by generate_ti2ti and generate_ti2ti_stepwise: runs the reference loop and yields
    (step, ids, info) after every step; `info` carries the image step's sampled ids (before re-masking) when one ran.
    `image_step_list` overrides the schedule of image steps (default: reference :157-159).

    `graph` (None = env MMADA_GRAPH=1): replay each kind of step — text-only, image step — as ONE hipGraph
    (mmada_graph_*): the launches of a step are a fixed sequence over fixed buffers because k, mask_len and the set of
    forwards are schedule-determined.  The first step of each kind runs eagerly, the second is captured,
    the rest replay.  Only at temperature == text_temperature == 0 (the image re-mask temperature is a by-value kernel
    argument that changes every step otherwise) and when the forward issues no host-side collective."""
    if not isinstance(model, LLaDAForMultiModalGeneration):
        raise TypeError("generate_ti2ti (MI355X) needs mmada_parallel_amd.LLaDAForMultiModalGeneration; "
                        "there is no PyTorch fallback path")
    if remasking not in ('low_confidence', 'random'):
        raise NotImplementedError(remasking)
    if remasking == 'random' and generator is not None:
        # the reference asks torch.rand for an int64 uniform on this path (:195-196) and raises
        raise RuntimeError("remasking='random' with an explicit generator raises in the reference too "
                           "(torch.rand(dtype=int64)); pass generator=None")
    lib, h = model._lib, model._handle
    device = model.device
    rng = rng or TorchRng()
    ids = input_ids.to(device=device, dtype=torch.long).clone().contiguous()
    B, L = ids.shape
    V = model.vocab

    num_vq_tokens = seq_len
    total_image_len = seq_len + seq_len // newline_every
    image_end = image_start + total_image_len
    T = text_end - text_start

    # ---- host-side schedules (one D2H copy at setup, none inside the loop) ----
    ids_host = ids.cpu()
    text_masked0 = ids_host[:, text_start:text_end] == MASK_TOKEN
    num_transfer = get_num_transfer_tokens(text_masked0, text_steps)  # [B, steps]
    remaining_text = text_masked0.sum(dim=1)  # [B]
    img_steps = set(image_step_indices(text_steps, timesteps) if image_step_list is None else image_step_list)
    pos_list = [i for i in range(image_start, image_end) if ids_host[0, i] != NEW_LINE]
    assert len(pos_list) == num_vq_tokens, f"Expected {num_vq_tokens} VQ tokens, got {len(pos_list)}"
    mlen = mask_len_schedule(num_vq_tokens, text_steps, noise_schedule)

    k_dev = num_transfer.t().contiguous().to(device=device, dtype=torch.int32)  # [steps, B]
    mlen_dev = torch.tensor(mlen, dtype=torch.int32, device=device)
    pos_map = torch.tensor(pos_list, dtype=torch.int32, device=device)
    N = num_vq_tokens
    brow = torch.arange(B, dtype=torch.int32, device=device)[:, None] * L
    text_rows = (brow + torch.arange(text_start, text_end, dtype=torch.int32, device=device)[None, :]).reshape(-1)
    img_rows_1 = (brow + pos_map[None, :]).reshape(-1)                                     # cond batch  [B*N]
    brow2 = torch.arange(2 * B, dtype=torch.int32, device=device)[:, None] * L
    img_rows_2 = (brow2 + pos_map[None, :]).reshape(-1)                                    # uncond batch [2B*N]
    scratch = torch.empty(B * T * 16, dtype=torch.uint8, device=device)
    argmax = torch.empty((B, N), dtype=torch.int32, device=device)
    pmax = torch.empty((B, N), dtype=torch.bfloat16, device=device)

    want_ut = cfg_scale > 0.0 and uncon_text is not None
    want_ui = cfg_img > 0.0 and uncon_image is not None
    need_uncond = want_ut or want_ui
    if uncon_text is not None:
        uncon_text = uncon_text.to(device=device, dtype=torch.long)
    if uncon_image is not None:
        uncon_image = uncon_image.to(device=device, dtype=torch.long)

    # ---- every buffer a step touches exists before the loop (fixed addresses: a step can be captured and replayed) ----
    # That includes the library's activation workspace: an image step's unconditional pair is a 2B forward, and a workspace
    # that grows THEN would leave an already captured text-step graph pointing into the freed allocation.
    if hasattr(model, "_ensure_ws"):
        model._ensure_ws(2 * B if need_uncond else B, L)
    CBs = codebook_size
    k_cur = torch.zeros(B, dtype=torch.int32, device=device)
    mlen_cur = torch.zeros(1, dtype=torch.int32, device=device)
    # the [B*T, V] logits are then never materialised (random re-masking ranks by a uniform draw: the one-rank kernels)
    vp_text = text_temperature == 0 and remasking == 'low_confidence' and model.vocab_parallel_head()
    text_logits = None if vp_text else torch.empty((B * T, V), dtype=torch.bfloat16, device=device)
    cond_vq = torch.empty((B * N, CBs), dtype=torch.bfloat16, device=device) if img_steps else None
    unc = torch.empty((2 * B, L), dtype=torch.long, device=device) if need_uncond else None
    unc_vq = torch.empty((2 * B * N, CBs), dtype=torch.bfloat16, device=device) if need_uncond and img_steps else None
    zeros_vq = None
    if not need_uncond and (cfg_scale != 0.0 or cfg_img != 0.0) and img_steps:
        zeros_vq = torch.zeros((B * N, CBs), dtype=torch.bfloat16, device=device)  # reference :275-278
    noise_buf = torch.zeros((B, N), dtype=torch.bfloat16, device=device)
    probs = torch.empty((B * N, CBs), dtype=torch.bfloat16, device=device) if temperature != 0 and img_steps else None

    masked_left = remaining_text.clone()
    # rows of the residual stream each forward is read at (host-side): the last block only computes those
    windowed = os.environ.get("MMADA_NO_WINDOW") != "1"
    img_win = (pos_list[0], pos_list[-1] + 1)

    def text_part(is_img, need_text):
        """Conditional forward (reference :177-178) + text step (:181-217): launches only."""
        lo = min(img_win[0] if is_img else L, text_start if need_text or not is_img else L)
        hi = max(img_win[1] if is_img else 0, text_end if need_text or not is_img else 0)
        model.forward_body(ids, consumed=(lo, hi) if windowed else None)
        st = abi.stream_ptr()
        if is_img:
            model.head_rows(img_rows_1, text_vocab_size, text_vocab_size + codebook_size, out=cond_vq)
        if need_text and vp_text:
            # tensor parallel: each rank holds vocab/tp columns of the LM head; rows are reduced to {max, arg-max, sum-exp}
            # and only those 16 bytes per row travel (mmada_text_select_tp) — the [B*T, V] logits exist on no rank
            abi.check(lib.mmada_text_select_tp(h, text_rows.data_ptr(), B, T, ids.data_ptr(), L, text_start,
                                               k_cur.data_ptr(), scratch.data_ptr(), st), "mmada_text_select_tp")
        elif need_text:
            model.head_rows(text_rows, 0, V, out=text_logits)  # [B*T, V]
            noisy = None
            if text_temperature != 0:
                noisy = add_gumbel_noise(text_logits.view(B, T, V), temperature=text_temperature,
                                         generator=generator, rng=rng).contiguous()
            if remasking == 'random':  # x0_p = torch.rand((B, T), device=...) (:198), drawn AFTER the Gumbel noise
                u = (rng or TorchRng()).rand((B, T), torch.float32, device, None).contiguous()
                abi.check(lib.mmada_text_select_random(h, text_logits.data_ptr(), abi.ptr(noisy), u.data_ptr(), B, T, V, V,
                                                       ids.data_ptr(), L, text_start, k_cur.data_ptr(), scratch.data_ptr(),
                                                       st), "mmada_text_select_random")
            else:
                abi.check(lib.mmada_text_select(h, text_logits.data_ptr(), abi.ptr(noisy), B, T, V, V, ids.data_ptr(), L,
                                                text_start, k_cur.data_ptr(), scratch.data_ptr(), st),
                          "mmada_text_select")

    def image_forwards():
        """Unconditional forwards of an image step (reference :243-274) + dual-CFG soft-max / arg-max (:282-295)."""
        st = abi.stream_ptr()
        ut = ui = None
        if need_uncond:
            # unconditional sequences: prefix overwritten in place, same length (reference :250-259, A.3)
            unc[:B].copy_(ids)
            unc[B:].copy_(ids)
            if uncon_text is not None:
                unc[:B, :uncon_text.shape[1]] = uncon_text
            if uncon_image is not None:
                unc[B:, :uncon_image.shape[1]] = uncon_image
            # both uncond forwards run whenever either scale > 0 (reference :243); only their image rows are read
            model.forward_body(unc, consumed=img_win if windowed else None)
            model.head_rows(img_rows_2, text_vocab_size, text_vocab_size + codebook_size, out=unc_vq)
            ut, ui = unc_vq[:B * N], unc_vq[B * N:]
        elif zeros_vq is not None:
            ut = ui = zeros_vq  # reference :275-278: uncond logits are zeros when no uncond input exists
        abi.check(lib.mmada_image_probs(h, cond_vq.data_ptr(), abi.ptr(ut), abi.ptr(ui), B, N, codebook_size,
                                        float(cfg_scale), float(cfg_img), abi.ptr(probs), argmax.data_ptr(),
                                        pmax.data_ptr(), st), "mmada_image_probs")

    def image_commit(sampled, p_sel, img_temp):
        abi.check(lib.mmada_image_commit(h, ids.data_ptr(), B, L, pos_map.data_ptr(), N, sampled.data_ptr(),
                                         p_sel.data_ptr(), noise_buf.data_ptr(), float(img_temp),
                                         mlen_cur.data_ptr(), int(text_vocab_size),
                                         int(codebook_size), abi.stream_ptr()), "mmada_image_commit")

    if graph is None:
        graph = os.environ.get("MMADA_GRAPH") == "1"
    graph = (bool(graph) and temperature == 0 and text_temperature == 0 and remasking == 'low_confidence'
             and model.graph_capturable())  # a replayed step would replay its random draws
    graphs, seen = {}, set()
    ws_epoch = getattr(model, "_ws_epoch", 0)
    side = torch.cuda.Stream(device=device) if graph else None
    if graph:  # the legacy default stream cannot be captured: the loop runs on a side stream, ordered after the caller's
        side.wait_stream(torch.cuda.current_stream(device))
    try:
        for step in range(text_steps):
            is_img = step in img_steps
            need_text = int(masked_left.sum()) > 0
            info = {"image_step": is_img, "sampled": None}
            with torch.cuda.stream(side) if graph else contextlib.nullcontext():
                k_cur.copy_(k_dev[step])
                mlen_cur.copy_(mlen_dev[step:step + 1])
                early_noise = is_img and temperature == 0 and text_temperature == 0 and remasking == 'low_confidence'
                if early_noise:
                    # randn is drawn even at temperature 0 (reference :30-33, A.2) so the RNG stream advances identically;
                    # nothing else draws in such a step (no Gumbel noise, no random re-mask ranks), so it can be drawn
                    # ahead of the step's launches
                    noise_buf.copy_(rng.randn((B, N), torch.bfloat16, device, generator))
                key = (is_img, need_text)
                if graph and graphs and getattr(model, "_ws_epoch", 0) != ws_epoch:
                    # the workspace moved after all (a foreign forward in between): captured pointers are stale
                    for g_old in graphs.values():
                        lib.mmada_graph_destroy(g_old)
                    graphs.clear()
                    seen.clear()
                ws_epoch = getattr(model, "_ws_epoch", 0)
                if graph and key in graphs:
                    abi.check(lib.mmada_graph_launch(graphs[key], abi.stream_ptr()), "mmada_graph_launch")
                    model.graph_replays += 1
                else:
                    capture = graph and key in seen  # first step of a kind: eager (first calls set attributes / carve)
                    if capture:
                        abi.check(lib.mmada_graph_begin(abi.stream_ptr()), "mmada_graph_begin")
                    try:
                        text_part(is_img, need_text)
                        if is_img:
                            image_forwards()
                            if temperature == 0:
                                if not early_noise:  # the reference's draw order: text Gumbel noise first (:13-16, :30-33)
                                    noise_buf.copy_(rng.randn((B, N), torch.bfloat16, device, generator))
                                image_commit(argmax, pmax, 0.0)
                    except Exception:
                        if capture:
                            lib.mmada_graph_abort(abi.stream_ptr())
                        raise
                    if capture:
                        g = C.c_void_p()
                        abi.check(lib.mmada_graph_end(abi.stream_ptr(), C.byref(g)), "mmada_graph_end")
                        graphs[key] = g
                        model.graph_nodes[key] = lib.mmada_graph_num_nodes(g)
                        abi.check(lib.mmada_graph_launch(g, abi.stream_ptr()), "mmada_graph_launch")
                        model.graph_replays += 1
                    seen.add(key)
                if need_text:
                    masked_left = masked_left - num_transfer[:, step]
                if is_img:
                    if temperature == 0:
                        info["sampled"] = argmax
                    else:
                        s64 = rng.multinomial(probs, generator)
                        p_sel = torch.gather(probs, -1, s64).view(B, N).contiguous()
                        sampled = s64.view(B, N).to(torch.int32).contiguous()
                        ratio = 1.0 * (step + 1) / text_steps
                        noise_buf.copy_(rng.randn((B, N), torch.bfloat16, device, generator))  # reference :30-33
                        image_commit(sampled, p_sel, temperature * (1.0 - ratio))
                        info["sampled"] = sampled
            if graph:
                torch.cuda.current_stream(device).wait_stream(side)  # whoever consumes `ids` sees the finished step
            yield step, ids, info
            if graph:
                side.wait_stream(torch.cuda.current_stream(device))
    finally:
        for g in graphs.values():
            lib.mmada_graph_destroy(g)
    yield text_steps, ids, {"image_step": False, "sampled": None, "pos_list": pos_list}


@torch.no_grad()
def generate_ti2ti(
    model,
    input_ids,
    text_start,
    text_end,
    image_start,
    seq_len,
    newline_every,
    text_steps=100,
    text_gen_length=256,
    text_block_length=64,
    timesteps=100,
    temperature=1.0,
    text_temperature=0.7,
    cfg_scale=0.0,
    cfg_img=4.0,
    uncon_text=None,
    uncon_image=None,
    tokenizer=None,
    remasking='low_confidence',
    noise_schedule=cosine_schedule,
    generator=None,
    text_vocab_size=126356,
    codebook_size=8192,
    return_state=False,
    rng=None,
    graph=None,
):
    """Joint text+image generation; returns (List[int] vq ids, str | List[int] text) like the reference
    (generators/parallel_generator.py:102-368).

    `graph=True` (or MMADA_GRAPH=1) replays every step as one hipGraph (see _ti2ti_steps; BASELINE configs[4]).
    `return_state=True` additionally returns the final `combined_input_ids` *before* the random fill of
    still-masked image tokens (reference :360-362) — the quantity parity tests compare."""
    ids = pos_list = None
    for _step, ids, info in _ti2ti_steps(model, input_ids, text_start, text_end, image_start, seq_len, newline_every,
                                         text_steps, text_gen_length, text_block_length, timesteps, temperature,
                                         text_temperature, cfg_scale, cfg_img, uncon_text, uncon_image, tokenizer,
                                         remasking, noise_schedule, generator, text_vocab_size, codebook_size, rng=rng,
                                         graph=graph):
        pos_list = info.get("pos_list", pos_list)

    # ===== final read-out (reference :346-368) =====
    final_ids = ids.cpu()
    check_tp_exchange(model)
    text_tokens = [t for t in final_ids[0, text_start:text_end].tolist() if t != MASK_TOKEN]
    generated_text = tokenizer.decode(text_tokens, skip_special_tokens=True) if tokenizer is not None else text_tokens
    image_tokens = []
    for pos in pos_list:
        token = int(final_ids[0, pos])
        if token != MASK_TOKEN:
            image_tokens.append(max(0, min(token - text_vocab_size, codebook_size - 1)))
        else:
            # still masked -> sample randomly from the global CPU RNG, exactly like the reference (A.1)
            image_tokens.append(int(torch.randint(0, codebook_size, (1,)).item()))
    if return_state:
        return image_tokens, generated_text, final_ids
    return image_tokens, generated_text


def stepwise_image_steps(text_steps: int):
    """Image-step schedule of the reference's Gradio generator (app.py:162-164): 30 % of the steps, from step 0."""
    return torch.linspace(0, text_steps - 1, int(text_steps * 0.3)).round().int().tolist()


def generate_ti2ti_stepwise(
    model, input_ids, text_start, text_end, image_start, seq_len, newline_every,
    text_steps=100, temperature=1.0, text_temperature=0.7, cfg_scale=0.0, cfg_img=4.0,
    uncon_text=None, uncon_image=None, tokenizer=None, remasking='low_confidence',
    noise_schedule=cosine_schedule, generator=None, text_vocab_size=126356,
    codebook_size=8192, vqvae=None, image_height=512, image_width=512,
):
    """Token-level mirror of the reference's streaming sampler `generate_ti2ti_stepwise` (app.py:143-398): the same
    loop as generate_ti2ti with the Gradio schedule of image steps, yielding after every step

        (step + 1, combined_input_ids [B, L] on the device, sampled VQ ids [B, N] of this step's image update or None,
         show)   with show == the reference's display cadence (step % 5 == 0, image steps, last step).

    Turning the yielded ids into what the UI displays is the caller's work: text through the tokenizer, and the preview of
    an image step through `utils.decode_step_preview(sampled, masked_cells, vqvae, image_height, image_width)` (the
    reference's decode_vq_to_image + gray overlay of the re-masked cells, app.py:310-339) — `vqvae`, `image_height` and
    `image_width` are accepted here for signature compatibility only."""
    sched = stepwise_image_steps(text_steps)
    with torch.no_grad():
        for step, ids, info in _ti2ti_steps(model, input_ids, text_start, text_end, image_start, seq_len, newline_every,
                                            text_steps, 256, 64, 0, temperature, text_temperature, cfg_scale, cfg_img,
                                            uncon_text, uncon_image, tokenizer, remasking, noise_schedule, generator,
                                            text_vocab_size, codebook_size, image_step_list=sched):
            if step >= text_steps:
                return
            show = step % 5 == 0 or info["image_step"] or step == text_steps - 1
            if step == text_steps - 1:
                check_tp_exchange(model)   # the last yield is the result: raise rather than hand out void tokens
            yield step + 1, ids, info["sampled"], show
