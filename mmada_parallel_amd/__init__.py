"""mmada_parallel_amd — MI355X-native hot path of the MMaDA-Parallel parallel text+image sampler.

Public surface mirrors the reference (tyfeld/MMaDA-Parallel, MMaDA-Parallel-A):
    from mmada_parallel_amd import LLaDAForMultiModalGeneration, generate_ti2ti
"""
from .model import LLaDAForMultiModalGeneration, LLaDAConfigLite  # noqa: F401
from .generators.parallel_generator import generate_ti2ti, generate_ti2ti_stepwise, cosine_schedule  # noqa: F401
from .generators.interleave_generator import interleave_generate  # noqa: F401
from .generators.image_generation_generator import generate_image  # noqa: F401
from .generators.mmu_generator import mmu_generate, mmu_generate_fast  # noqa: F401
from .generators.t2i_generator import t2i_generate, t2i_generate_decoding_stepwise  # noqa: F401
from .mmada import MMadaModelLM  # noqa: F401
from .vq import MAGVITv2  # noqa: F401
from .vqmodel import VQModel  # noqa: F401

__all__ = ["LLaDAForMultiModalGeneration", "LLaDAConfigLite", "generate_ti2ti", "generate_ti2ti_stepwise", "interleave_generate", "cosine_schedule", "MAGVITv2", "generate_image", "mmu_generate", "mmu_generate_fast", "t2i_generate", "t2i_generate_decoding_stepwise", "MMadaModelLM", "VQModel"]
