"""Contract 2 of SURVEY §8b, exercised by the reference itself: the UNMODIFIED `generate_ti2ti`
(/root/reference/MMaDA-Parallel-A/generators/parallel_generator.py:102-368, imported, never copied) drives this repo's model
class surface — `model(ids, infer=True, use_cache=False).logits` (`:178,263-264`), `model.config.{text_vocab_size,
codebook_size}` through `getattr` defaults (inference.py:87-89) and `model.device` (modeling_llada.py:1153-1159).

The GPU box has no /root/reference and this container has no GPU, so the two cannot meet in one process.  What CAN be checked
here, on the CPU: the reference loop runs against `mmada_parallel_amd.LLaDAForMultiModalGeneration.forward` ITSELF (argument
validation, row bookkeeping, `CausalLMOutputLite`, `LLaDAConfigLite`) with only the two device entry points underneath it
(`forward_body`, `head_rows` — the ctypes calls) replaced by recorded logits, and must reproduce the trajectory the reference
recorded with its own model stand-in (tests/golden/sampler_traj.npz, written by oracle/gen_golden.py) call for call.
"""
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN, SAMPLER_CASES, STUB_CB, STUB_TEXT_VOCAB, stub_logits, tiny_job
from mmada_parallel_amd.model import CausalLMOutputLite, LLaDAConfigLite, LLaDAForMultiModalGeneration

REF_A = "/root/reference/MMaDA-Parallel-A"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_A), reason="the reference tree is only mounted in the build container")


class RecordedForward(LLaDAForMultiModalGeneration):
    """The product's model class with the two ctypes entry points under `forward` served from recorded logits."""

    def __init__(self, seed, vocab):  # the real constructor needs a GPU and a checkpoint
        self.device = torch.device("cpu")
        self.vocab, self.tp_size, self._comm_in_library = vocab, 1, False
        self.config = LLaDAConfigLite(text_vocab_size=STUB_TEXT_VOCAB, codebook_size=STUB_CB, vocab_size=vocab)
        self._seed, self.calls = seed, []

    def forward_body(self, input_ids):
        self.calls.append(input_ids.clone())
        self._shape = tuple(input_ids.shape)
        self._logits = stub_logits(self._seed, len(self.calls), input_ids.shape[0], input_ids.shape[1], self.vocab)

    def head_rows(self, rows, col_begin, col_end, out=None):
        return self._logits.reshape(-1, self.vocab)[rows.long(), col_begin:col_end]

    def __del__(self):
        pass


def _reference_generate():
    sys.path.insert(0, REF_A)
    try:
        from generators.parallel_generator import generate_ti2ti
    finally:
        sys.path.remove(REF_A)
    return generate_ti2ti


@pytest.mark.parametrize("name", list(SAMPLER_CASES))
def test_reference_loop_drives_the_product_model_class(name):
    generate_ti2ti = _reference_generate()
    z = np.load(os.path.join(GOLDEN, "sampler_traj.npz"))
    job, kw = tiny_job(), SAMPLER_CASES[name]
    model = RecordedForward(int(z[name + "_seed"]), STUB_TEXT_VOCAB + STUB_CB)
    # the way inference.py:87-89 reads the vocabulary split
    text_vocab = getattr(model.config, "text_vocab_size", 126356)
    codebook = getattr(model.config, "codebook_size", 8192)
    assert (text_vocab, codebook) == (STUB_TEXT_VOCAB, STUB_CB)
    assert getattr(LLaDAConfigLite(), "text_vocab_size", 126356) == 126356   # an absent key takes the reference's default
    ids = job["input_ids"].to(model.device)
    torch.manual_seed(1234)  # pins the one torch.randint fill, as the recording did
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        vq, text = generate_ti2ti(model, ids, job["text_start"], job["text_end"], job["image_start"], job["seq_len"],
                                  job["newline_every"], uncon_text=job["uncon_text"], uncon_image=job["uncon_image"],
                                  tokenizer=None, generator=None, text_vocab_size=text_vocab, codebook_size=codebook,
                                  temperature=0.0, text_temperature=0.0, **kw)   # the recording's settings (gen_golden.py)
    calls = torch.cat(model.calls, 0)
    ref_calls = torch.from_numpy(z[name + "_calls"])
    assert calls.shape == ref_calls.shape and torch.equal(calls, ref_calls), "ids handed to the model, call for call"
    assert list(vq) == z[name + "_vq"].tolist()
    assert list(text) == z[name + "_text"].tolist()


def test_output_object_and_keyword_contract():
    """`forward` returns an object whose `.logits` is [B, L, V] in the model dtype; the reference's other keywords are accepted
    (`use_cache`, `to_compute_mask`, `cat`: modeling_llada.py:1468-1493) and the training path is refused loudly."""
    model = RecordedForward(3, STUB_TEXT_VOCAB + STUB_CB)
    ids = tiny_job()["input_ids"]
    out = model(ids, infer=True, use_cache=False)
    assert isinstance(out, CausalLMOutputLite) and out.logits.shape == (1, ids.shape[1], model.vocab)
    assert out.logits.dtype == torch.bfloat16
    assert torch.equal(out.logits, stub_logits(3, 1, 1, ids.shape[1], model.vocab))
    with pytest.raises(NotImplementedError):
        model(ids, infer=False)
    with pytest.raises(ValueError):
        model(ids, infer=True, to_compute_mask=torch.ones_like(ids, dtype=torch.bool))
