"""Drop-in for the reference's `MMadaModelLM` (MMaDA-Parallel-M/models/modeling_mmada.py:108-116): the same denoiser
(`LLaDAModelLM`, state-dict keys `model.transformer.*`) with the M samplers as METHODS, because that is how the
reference calls them (`model.interleave_generate(...)`, MMaDA-Parallel-M/inference.py:113;
`model.mmu_generate(...)`, training/train_interleave.py:1254).  Everything runs on libmmada_mi355x.so.
"""
from __future__ import annotations

from .generators.interleave_generator import interleave_generate as _interleave_generate
from .generators.mmu_generator import mmu_generate as _mmu_generate, mmu_generate_fast as _mmu_generate_fast
from .generators.t2i_generator import t2i_generate as _t2i_generate
from .generators.t2i_generator import t2i_generate_decoding_stepwise as _t2i_stepwise
from .model import LLaDAForMultiModalGeneration


class MMadaModelLM(LLaDAForMultiModalGeneration):
    @classmethod
    def from_pretrained(cls, path, trust_remote_code=True, torch_dtype=None, **kw):
        import torch

        return super().from_pretrained(path, torch_dtype=torch_dtype or torch.bfloat16, **kw)

    def interleave_generate(self, input_ids=None, uncond_input_ids=None, **kw):   # modeling_mmada.py:117-248
        return _interleave_generate(self, input_ids, uncond_input_ids, **kw)

    def t2i_generate(self, input_ids=None, uncond_input_ids=None, **kw):          # :264-359
        return _t2i_generate(self, input_ids, uncond_input_ids, **kw)

    def t2i_generate_decoding_stepwise(self, input_ids=None, uncond_input_ids=None, **kw):   # :768-875
        return _t2i_stepwise(self, input_ids, uncond_input_ids, **kw)

    def mmu_generate(self, idx=None, **kw):                                       # :618-692
        return _mmu_generate(self, idx, **kw)

    def mmu_generate_fast(self, idx=None, **kw):                                  # :694-766
        return _mmu_generate_fast(self, idx, **kw)
