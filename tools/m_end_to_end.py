#!/usr/bin/env python
"""End-to-end MMaDA-Parallel-M edit on one MI355X, pixels in -> pixels out, at the reference's sizes
(MMaDA-Parallel-M/inference.py:73-130): 512x512 image -> MAGVITv2.get_code (1024 tokens) -> interleave_generate
(text_steps 128, image_steps 30, text_cfg 2.5, image_cfg 4.0; one batch-2 forward of the 8B denoiser per step) ->
MAGVITv2.decode_code -> uint8 image.  Synthetic weights / image / prompt (no checkpoint offline); stage timings only.
Measurement tool — the headline benchmark is bench.py."""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmada_parallel_amd import MAGVITv2, MMadaModelLM, synth  # noqa: E402
from mmada_parallel_amd.vq import to_uint8_image  # noqa: E402


def main(layers=None):
    dev = "cuda:0"
    cfg = dict(synth.CFG_8B)
    if layers:
        cfg["n_layers"] = layers
    sd = synth.synthetic_state_dict(cfg, seed=0, device=dev)
    model = MMadaModelLM.from_state_dict(synth.full_config(cfg), sd, device=dev, max_batch=2)
    del sd
    vsd = {"encoder." + k: v for k, v in synth.synthetic_vq_state_dict(synth.VQ_ENC_CFG_M, 1).items()}
    vsd.update({"decoder." + k: v for k, v in synth.synthetic_vq_state_dict(synth.VQ_CFG_M, 2).items()})
    vq = MAGVITv2(vsd, device=dev)
    text_vocab = synth.TEXT_VOCAB

    class Tok:
        bos_token_id = 126080

        def __len__(self):
            return text_vocab

    cfgobj = SimpleNamespace(model=SimpleNamespace(mmada=SimpleNamespace(num_vq_tokens=1024, codebook_size=8192)),
                             dataset=SimpleNamespace(preprocessing=SimpleNamespace(max_seq_length=256)))
    img = synth.synthetic_image(1, 512, 512, seed=5).to(dev)
    g = torch.Generator().manual_seed(3)
    text = torch.randint(0, 100000, (40,), generator=g).to(dev)
    un_text = torch.randint(0, 100000, (40,), generator=g).to(dev)

    def once():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tokens = vq.get_code(img) + text_vocab                                   # inference.py:79
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        head = torch.tensor([126340, 126084], device=dev)                        # <|interleave|>, <|soi|> stand-ins
        eoi = torch.tensor([126085], device=dev)
        inp = torch.cat([head, tokens[0], eoi, text])
        unc = torch.cat([head, torch.zeros_like(tokens[0]), eoi, un_text])
        out_img, out_text = model.interleave_generate(
            inp, unc, text_cfg=2.5, image_cfg=4.0, text_steps=128, image_steps=30, config=cfgobj,
            reserved_token_mapping={"<|soi|>": 126084, "<|eoi|>": 126085}, uni_prompting=SimpleNamespace(text_tokenizer=Tok()))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pix = to_uint8_image(vq.decode_code(out_img))
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        return (t1 - t0, t2 - t1, t3 - t2), pix, inp.numel()

    once()
    (te, ts, td), pix, P = once()
    L = P + 1 + 1024 + 1 + 256
    print(f"M end to end (L={L}, 128 batch-2 forwards): get_code {te*1e3:.1f} ms | interleave_generate {ts:.2f} s | "
          f"decode_code+uint8 {td*1e3:.1f} ms | total {te+ts+td:.2f} s/image = {1/(te+ts+td):.4f} images/s; "
          f"output {tuple(pix.shape)} {pix.dtype}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else None)
