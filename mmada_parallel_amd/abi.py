"""ctypes binding of libmmada_mi355x.so (include/mmada_mi355x.h).

This is the only place Python touches the native library.  There is NO fallback: if the shared object is missing
or a call fails, an exception is raised (the product path never routes through the CPU oracle or torch ops).
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

c_void_p, c_int, c_float, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class MmadaCfg(C.Structure):
    """struct mmada_cfg (include/mmada_mi355x.h)."""

    _fields_ = [
        ("d_model", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
        ("head_dim", C.c_int32), ("mlp_hidden", C.c_int32), ("vocab", C.c_int32), ("max_seq", C.c_int32),
        ("rms_eps", C.c_float), ("rope_theta", C.c_float),
        ("tp_rank", C.c_int32), ("tp_size", C.c_int32),
        ("mask_token_id", C.c_int32), ("text_vocab_size", C.c_int32), ("codebook_size", C.c_int32),
        ("reserved", C.c_int32),
    ]


# name -> (restype, argtypes); must list every symbol the header declares (checked by tests/test_abi.py)
SIGNATURES = {
    "mmada_create": (c_int, [C.POINTER(MmadaCfg), C.POINTER(C.c_float), C.POINTER(c_void_p)]),
    "mmada_destroy": (c_int, [c_void_p]),
    "mmada_clone_shared": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "mmada_last_error": (C.c_char_p, []),
    "mmada_abi_version": (c_int, []),
    "mmada_bind_globals": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "mmada_bind_layer": (c_int, [c_void_p, c_int] + [c_void_p] * 9 + [c_void_p]),
    "mmada_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "mmada_set_workspace": (c_int, [c_void_p, c_void_p, c_size_t]),
    "mmada_set_consumed_rows": (c_int, [c_void_p, c_int, c_int]),
    "mmada_forward_body": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mmada_head_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mmada_cache_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "mmada_cache_bind": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "mmada_forward_cached": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mmada_cache_head_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mmada_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mmada_embed": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mmada_attn_partial": (c_int, [c_void_p, c_int, c_void_p]),
    "mmada_mlp_partial": (c_int, [c_void_p, c_int, c_void_p]),
    "mmada_stream_ptr": (c_void_p, [c_void_p]),
    "mmada_stream_bytes": (c_size_t, [c_void_p]),
    "mmada_read_stream": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mmada_debug_buffer": (c_int, [c_void_p, c_int, C.POINTER(c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mmada_text_select": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p]),
    "mmada_text_select_random": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                         c_int, c_void_p, c_void_p, c_void_p]),
    "mmada_image_probs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "mmada_image_commit": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                   c_float, c_void_p, c_int, c_int, c_void_p]),
    "mmada_text_select_cfg": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                      c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mmada_image_probs_m": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "mmada_image_commit_m": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                     c_float, c_void_p, c_int, c_void_p]),
    "mmada_image_commit_g": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                     c_float, c_void_p, c_int, c_void_p]),
    "mmada_lfq_gather": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mmada_profile_begin": (c_int, [c_void_p, c_int]),
    "mmada_profile_end": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "mmada_set_option": (c_int, [C.c_char_p, c_int]),
    "mmada_gemm_swiglu_bt": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mmada_gemm_plan": (c_int, [c_int, c_int, c_int]),
    "mmada_attention_plan": (c_int, [c_int, c_int, c_int]),
    "mmada_probe_f2bf": (c_int, [c_void_p, c_void_p, C.c_int64, c_void_p]),
    "mmada_mfma_probe_bytes": (c_size_t, []),
    "mmada_mfma_probe": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mmada_gemm_bt": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mmada_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "mmada_vq_create": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "mmada_vq_create_encoder": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "mmada_vq_get_code": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "mmada_vq_create_vqmodel": (c_int, [c_void_p, c_int, C.POINTER(c_void_p)]),
    "mmada_vq_nearest_code": (c_int, [c_void_p, c_void_p, C.c_int64, c_void_p, c_void_p]),
    "mmada_vq_destroy": (None, [c_void_p]),
    "mmada_vq_bind": (c_int, [c_void_p, C.c_char_p, c_void_p, C.c_int64, c_void_p]),
    "mmada_vq_num_unbound": (c_int, [c_void_p]),
    "mmada_vq_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "mmada_vq_decode_code": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mmada_vq_conv2d": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "mmada_vq_group_norm": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "mmada_vq_group_norm_scratch_bytes": (c_size_t, [c_int]),
    "mmada_comm_export_bytes": (c_int, []),
    "mmada_comm_create": (c_int, [c_void_p, c_int, c_void_p]),
    "mmada_comm_connect_ipc": (c_int, [c_void_p, c_void_p]),
    "mmada_comm_connect_local": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "mmada_comm_unique_id": (c_int, [c_void_p, C.c_char_p]),
    "mmada_comm_connect_rccl": (c_int, [c_void_p, c_void_p, C.c_char_p]),
    "mmada_probe_cu_mask": (c_int, [c_void_p, c_int, c_void_p, c_int]),
    "mmada_comm_set_mode": (c_int, [c_void_p, c_int]),
    "mmada_comm_set_partition": (c_int, [c_void_p, c_int]),
    "mmada_comm_partition": (c_int, [c_void_p]),
    "mmada_comm_streams": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mmada_comm_rccl_nranks": (c_int, [c_void_p]),
    "mmada_comm_set_timeout": (c_int, [c_void_p, C.c_double]),
    "mmada_comm_status": (c_int, [c_void_p, C.POINTER(c_int), C.POINTER(c_int), C.POINTER(c_int), c_void_p]),
    "mmada_comm_part_ptr": (c_void_p, [c_void_p]),
    "mmada_comm_exchange": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mmada_text_select_tp": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mmada_comm_destroy": (c_int, [c_void_p]),
    "mmada_graph_begin": (c_int, [c_void_p]),
    "mmada_graph_end": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "mmada_graph_abort": (c_int, [c_void_p]),
    "mmada_graph_launch": (c_int, [c_void_p, c_void_p]),
    "mmada_graph_num_nodes": (c_int, [c_void_p]),
    "mmada_graph_destroy": (c_int, [c_void_p]),
    "mmada_sdpa": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
}

_lib = None


class MmadaError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the shared library (once).  Raises if it has not been built — no silent fallback."""
    global _lib
    if _lib is None:
        path = os.environ.get("MMADA_MI355X_LIB", _build.LIB_PATH)
        if not os.path.exists(path):
            raise MmadaError(
                f"{path} not found: build it with `python -m mmada_parallel_amd.build` (hipcc, gfx950). "
                "There is no CPU / PyTorch fallback for the hot path.")
        cdll = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = cdll
    return _lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().mmada_last_error()
        raise MmadaError(f"{what or 'mmada call'} failed: {msg.decode() if msg else status}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def stream_ptr() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
