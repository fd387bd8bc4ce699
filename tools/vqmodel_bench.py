#!/usr/bin/env python
"""A-variant image tokenizer (mmada_parallel_amd.VQModel, f16 / 8192-code geometry, synthetic weights): time of the two calls
the benchmark makes per 512 x 512 image.  MI355X: encode + quantise 17.9 ms, decode 29.1 ms."""
import sys, torch
sys.path.insert(0, ".")
from mmada_parallel_amd import VQModel, synth
cfg = synth.VQMODEL_CFG_A
vq = VQModel.from_state_dict(cfg, synth.synthetic_vqmodel_state_dict(cfg, 2), device="cuda:0")
x = ((synth.synthetic_image(1, 512, 512, seed=5) + 1) * 0.5).clamp(0, 1).cuda()
codes = torch.randint(0, 8192, (1, 32, 32), device="cuda")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("encode+quantize 512x512: %.2f ms" % t(lambda: vq.quantize(vq.encode(x).latents)))
print("decode 32x32 -> 512x512: %.2f ms" % t(lambda: vq.decode(codes, force_not_quantize=True)))
