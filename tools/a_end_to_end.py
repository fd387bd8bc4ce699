#!/usr/bin/env python
"""End-to-end MMaDA-Parallel-A edit on one MI355X, pixels in -> pixels out, the way the reference CLI runs a job
(MMaDA-Parallel-A/inference.py:109-245):

    prompt template (utils/prompt_utils.py:209-233) -> tokenizer -> PIL image -> encode_img_with_breaks (VQ encode) ->
    interleaved sequence (:129-161) -> generate_ti2ti (:169-193) -> decode_vq_to_image (:218-225) -> PNG (+ side-by-side)

on the mirror classes of this package: LLaDAForMultiModalGeneration, VQModel, utils.*.  No checkpoint or tokenizer is
available offline, so the weights are synthetic (8B shapes; the f16 / 8192-code VQModel geometry) and the tokenizer is a
deterministic stand-in with the two calls the reference makes (`tokenizer(text)["input_ids"]`, `.decode`).  The A tokenizer
(diffusers.VQModel) is NOT vendored in the reference tree: its arithmetic here is a restatement, PARITY UNPINNED (DESIGN.md
§6b) — this script demonstrates the call sequence and times its stages; the headline benchmark is bench.py.

    python tools/a_end_to_end.py [--layers N] [--height 512 --width 512] [--painting-mode inpainting] [--out DIR]
"""
import argparse
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mmada_parallel_amd import LLaDAForMultiModalGeneration, VQModel, generate_ti2ti, synth  # noqa: E402
from mmada_parallel_amd.utils import (build_ti2ti_sequence, calculate_vq_params, decode_vq_to_image, encode_img_with_breaks,  # noqa: E402
                                      encode_img_with_paint, generate_text_image_to_text_image_prompt)

SYSTEM_PROMPT = "Generate an image applying the following editing instruction based on the original image."


class FakeTokenizer:
    """Stand-in for the LLaDA tokenizer: one id per whitespace-separated piece (a stable hash below the special-token
    range), the template tags as single ids; `decode` returns a printable rendering of the ids."""

    TAGS = {"<system>": 126340, "</system>": 126341, "<user>": 126342, "</user>": 126343, "<uncondition>": 126351,
            "</answer>": 126355}

    def __call__(self, text, add_special_tokens=True):
        for tag in self.TAGS:
            text = text.replace(tag, f" {tag} ")
        return _Ids([self.TAGS[p] if p in self.TAGS else zlib.crc32(p.encode()) % 120000 for p in text.split()])

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(f"<{int(t)}>" for t in ids)


class _Ids(dict):
    """What a tokenizer call returns: `["input_ids"]` and `.input_ids` (the reference uses both, inference.py:115,147)."""

    def __init__(self, ids):
        super().__init__(input_ids=ids)
        self.input_ids = ids


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=None, help="fewer denoiser blocks (smoke runs)")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--text-steps", type=int, default=128)
    ap.add_argument("--timesteps", type=int, default=64)
    ap.add_argument("--text-gen-length", type=int, default=256)
    ap.add_argument("--cfg-scale", type=float, default=0.0)
    ap.add_argument("--cfg-img", type=float, default=4.0)
    ap.add_argument("--painting-mode", choices=["inpainting", "outpainting"], default=None)
    ap.add_argument("--prompt", default="Make the sky look like a watercolour painting at sunset.")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "a_e2e"))
    args = ap.parse_args()
    if args.painting_mode and (args.height, args.width) != (512, 512):
        ap.error("painting mode edits the 512x512 conditioning picture in place: the output grid is the input grid (inference.py:141-146)")
    from PIL import Image

    dev = "cuda:0"
    cfg = dict(synth.CFG_8B)
    if args.layers:
        cfg["n_layers"] = args.layers
    sd = synth.synthetic_state_dict(cfg, seed=0, device=dev)
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=dev, max_batch=2)
    del sd
    vqvae = VQModel.from_state_dict(synth.VQMODEL_CFG_A, synth.synthetic_vqmodel_state_dict(synth.VQMODEL_CFG_A, 2), device=dev)
    tokenizer = FakeTokenizer()

    # inference.py:109-127 — prompt strings, token ids, the conditioning picture (512 x 512 after the centre crop)
    input_prompt, uncon_text = generate_text_image_to_text_image_prompt(args.prompt, SYSTEM_PROMPT)
    prompt_ids = tokenizer(input_prompt)["input_ids"]
    uncon_text_ids = tokenizer(uncon_text)["input_ids"]
    pix = ((synth.synthetic_image(1, 512, 512, seed=5)[0] + 1.0) * 127.5).clamp(0, 255).permute(1, 2, 0).numpy().astype(np.uint8)
    img = Image.fromarray(pix, "RGB")

    def once():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        input_img_token = encode_img_with_breaks(img, vqvae)                                   # :127
        img_mask_token = None
        if args.painting_mode:                                                                  # :141-144
            img_mask_token, _ = encode_img_with_paint(img, vqvae=vqvae, mask_h_ratio=1.0, mask_w_ratio=0.2,
                                                      mask_mode=args.painting_mode)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        seq = build_ti2ti_sequence(prompt_ids, uncon_text_ids, input_img_token, args.height, args.width, args.text_gen_length,
                                   tokenizer("</answer>", add_special_tokens=False).input_ids,
                                   img_mask_token=img_mask_token)                               # :129-161
        con = torch.tensor(seq["input_ids"], device=dev).unsqueeze(0)
        ut = torch.tensor(seq["uncon_text"], device=dev).unsqueeze(0)
        ui = torch.tensor(seq["uncon_image"], device=dev).unsqueeze(0)
        tokens, text = generate_ti2ti(model, con, seq["text_start"], seq["text_end"], seq["image_start"], seq["seq_len"],
                                      seq["newline_every"], text_steps=args.text_steps, text_gen_length=args.text_gen_length,
                                      timesteps=args.timesteps, temperature=0.0, text_temperature=0.0, cfg_scale=args.cfg_scale,
                                      cfg_img=args.cfg_img, uncon_text=ut, uncon_image=ui, tokenizer=tokenizer)   # :169-193
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        os.makedirs(args.out, exist_ok=True)
        save_path = os.path.join(args.out, f"edit_{args.height}x{args.width}_t{args.timesteps}_ti2ti.png")
        out_img = decode_vq_to_image(torch.tensor(tokens, dtype=torch.long, device=dev).unsqueeze(0), save_path,
                                     image_height=args.height, image_width=args.width, vqvae=vqvae)      # :218-225
        canvas = Image.new("RGB", (img.size[0] + out_img.size[0], max(img.size[1], out_img.size[1])), "white")   # :227-233
        canvas.paste(img, (0, 0))
        canvas.paste(out_img, (img.size[0], 0))
        canvas.save(save_path.replace(".png", "_concat.png"))
        with open(save_path.replace(".png", "_thinking.txt"), "w", encoding="utf-8") as f:
            f.write(f"{text}\n")
        t3 = time.perf_counter()
        return (t1 - t0, t2 - t1, t3 - t2), con.shape[1], len(tokens), out_img.size, save_path

    once()
    (te, ts, td), L, ntok, size, path = once()
    gh = calculate_vq_params(args.height, args.width)
    print(f"A end to end (L={L}, {args.text_steps} text + {args.timesteps} image steps, output grid {gh[2]}x{gh[3]}): "
          f"encode_img_with_breaks {te * 1e3:.1f} ms | generate_ti2ti {ts:.2f} s | decode_vq_to_image + PNG {td * 1e3:.1f} ms | "
          f"total {te + ts + td:.2f} s/image = {1 / (te + ts + td):.4f} images/s; {ntok} VQ tokens -> {size[0]}x{size[1]} image at {path}")


if __name__ == "__main__":
    main()
