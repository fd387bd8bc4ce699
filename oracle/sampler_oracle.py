"""ctypes wrapper of oracle/sampler_oracle.c (CPU ORACLE — test infrastructure only, see the C file's header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "sampler_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_log_conf.restype = C.c_uint16
        _lib.oracle_log_conf.argtypes = [C.c_uint16]
    return _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _bits(t):  # bf16 tensor -> contiguous int16 view (raw bits) on CPU
    return None if t is None else t.detach().to("cpu", torch.bfloat16).contiguous().view(torch.int16)


def text_select(logits, noisy, ids, text_start, k, mask_id=126336):
    """logits/noisy: bf16 [B,T,V]; ids: int64 [B,L] (copied); k: int [B]. Returns (ids_out, conf f64 [B,T], x0 [B,T])."""
    B, T, V = logits.shape
    lb, nb = _bits(logits), _bits(noisy)
    ids = ids.detach().to("cpu", torch.long).clone().contiguous()
    kk = torch.as_tensor(k, dtype=torch.int32).contiguous()
    conf = torch.empty((B, T), dtype=torch.float64)
    x0 = torch.empty((B, T), dtype=torch.int32)
    lib().oracle_text_select(_p(lb), _p(nb), B, T, V, V, _p(ids), ids.shape[1], text_start, _p(kk), mask_id, _p(conf), _p(x0))
    return ids, conf, x0


def image_probs(cond, ut, ui, cfg_scale, cfg_img, want_probs=False):
    """cond/ut/ui: bf16 [B,N,CB]. Returns (argmax int32 [B,N], pmax bf16 [B,N], probs bf16 [B,N,CB] | None)."""
    B, N, CB = cond.shape
    cb, tb, ib = _bits(cond), _bits(ut), _bits(ui)
    probs = torch.empty((B, N, CB), dtype=torch.int16) if want_probs else None
    am = torch.empty((B, N), dtype=torch.int32)
    pm = torch.empty((B, N), dtype=torch.int16)
    lib().oracle_image_probs(_p(cb), _p(tb), _p(ib), B, N, CB, C.c_float(cfg_scale), C.c_float(cfg_img), _p(probs), _p(am), _p(pm))
    return am, pm.view(torch.bfloat16), (probs.view(torch.bfloat16) if want_probs else None)


def image_commit(ids, pos_map, sampled, p_sel, noise, remask_temp, mask_len_sched, mask_id=126336, text_vocab=126356,
                 codebook=8192):
    """Returns the updated ids (int64 [B,L])."""
    ids = ids.detach().to("cpu", torch.long).clone().contiguous()
    B, L = ids.shape
    pm = torch.as_tensor(pos_map, dtype=torch.int32).cpu().contiguous()
    N = pm.numel()
    s = sampled.detach().to("cpu", torch.int32).contiguous()
    pb, nb = _bits(p_sel), _bits(noise)
    lib().oracle_image_commit(_p(ids), B, L, _p(pm), N, _p(s), _p(pb), _p(nb), C.c_float(remask_temp), int(mask_len_sched),
                              mask_id, text_vocab, codebook)
    return ids


def log_conf_table() -> torch.Tensor:
    """bf16 log-confidence of every non-negative bf16 probability bit pattern 0..0x7f7f as int16 bits."""
    l = lib()
    return torch.tensor(np.array([l.oracle_log_conf(p) for p in range(0x7f80)], dtype=np.uint16).view(np.int16))


def lfq_gather(idx, nbits):
    idx = idx.detach().to("cpu", torch.long).contiguous()
    B, N = idx.shape
    out = torch.empty((B, nbits, N), dtype=torch.float32)
    lib().oracle_lfq_gather(_p(idx), B, N, nbits, _p(out))
    return out


# ---- M variant (MMaDA-Parallel-M/models/modeling_mmada.py:117-248) ------------------------------------------------
def text_select_cfg(cond, unc, text_cfg, ids, text_start, k, x0_in=None, mask_id=126336):
    B, T, V = cond.shape
    cb, ub = _bits(cond), _bits(unc)
    ids = ids.detach().to("cpu", torch.long).clone().contiguous()
    kk = torch.as_tensor(k, dtype=torch.int32).contiguous()
    x0i = None if x0_in is None else x0_in.detach().to("cpu", torch.int32).contiguous()
    conf = torch.empty((B, T), dtype=torch.float64)
    x0 = torch.empty((B, T), dtype=torch.int32)
    lib().oracle_text_select_cfg(_p(cb), _p(ub), C.c_float(text_cfg), _p(x0i), B, T, V, V, _p(ids), ids.shape[1],
                                 text_start, _p(kk), mask_id, _p(conf), _p(x0))
    return ids, conf, x0


def image_probs_m(cond, unc, image_cfg):
    B, N, CB = cond.shape
    cb, ub = _bits(cond), _bits(unc)
    probs = torch.empty((B, N, CB), dtype=torch.int16)
    am = torch.empty((B, N), dtype=torch.int32)
    pm = torch.empty((B, N), dtype=torch.int16)
    lib().oracle_image_probs_m(_p(cb), _p(ub), B, N, CB, C.c_float(image_cfg), _p(probs), _p(am), _p(pm))
    return am, pm.view(torch.bfloat16), probs.view(torch.bfloat16)


def image_commit_m(ids, pos_map, sampled, p_sel, gumbel, remask_temp, mask_len_sched, mask_id=126336, text_vocab=126356):
    ids = ids.detach().to("cpu", torch.long).clone().contiguous()
    B, L = ids.shape
    pm = torch.as_tensor(pos_map, dtype=torch.int32).cpu().contiguous()
    s = sampled.detach().to("cpu", torch.int32).contiguous()
    pb, gb = _bits(p_sel), _bits(gumbel)
    lib().oracle_image_commit_m(_p(ids), B, L, _p(pm), pm.numel(), _p(s), _p(pb), _p(gb), C.c_float(remask_temp),
                                int(mask_len_sched), mask_id, text_vocab)
    return ids
