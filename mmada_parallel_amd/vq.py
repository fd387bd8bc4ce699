"""Host mirror of the reference's MAGVITv2 VQ model for the token -> pixel direction.

Same call surface as MMaDA-Parallel-M (models/modeling_magvitv2.py:408-433, used at inference.py:56-59,127-130):

    vq_model = MAGVITv2.from_pretrained(path).to(device); vq_model.requires_grad_(False); vq_model.eval()
    images = vq_model.decode_code(output_image_ids)          # [B, 3, H, W] fp32, unclamped

The arithmetic runs in libmmada_mi355x.so (csrc/vq_decoder.hip) through the C-ABI of include/mmada_mi355x.h; there
is no PyTorch fallback — without the HIP library or a GPU the constructor raises.  The encoder direction
(`encode` / `get_code`) is not part of the sampler's path and is not built.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from typing import Dict, Optional

import torch

from . import abi

DEFAULT_CFG = dict(ch=128, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=[4, 4, 3, 4, 3], z_channels=13, out_ch=3)


class VqCfg(C.Structure):
    """struct mmada_vq_cfg (include/mmada_mi355x.h)."""

    _fields_ = [("ch", C.c_int32), ("n_levels", C.c_int32), ("ch_mult", C.c_int32 * 8),
                ("num_res_blocks", C.c_int32 * 8), ("z_channels", C.c_int32), ("out_ch", C.c_int32)]


class MAGVITv2:
    def __init__(self, state_dict: Dict[str, torch.Tensor], config: Optional[dict] = None,
                 device: Optional[torch.device] = None):
        cfg = dict(DEFAULT_CFG)
        cfg.update(config or {})
        if len(cfg["ch_mult"]) != len(cfg["num_res_blocks"]) or not 1 <= len(cfg["ch_mult"]) <= 8:
            raise ValueError("ch_mult / num_res_blocks must have the same length (1..8)")
        if not torch.cuda.is_available():
            raise RuntimeError("MAGVITv2 (MI355X) needs a GPU: there is no CPU fallback")
        self.config = cfg
        self.device = torch.device(device if device is not None else "cuda:0")
        self._lib = abi.lib()
        c = VqCfg()
        c.ch, c.n_levels, c.z_channels, c.out_ch = cfg["ch"], len(cfg["ch_mult"]), cfg["z_channels"], cfg["out_ch"]
        for i, (m, n) in enumerate(zip(cfg["ch_mult"], cfg["num_res_blocks"])):
            c.ch_mult[i], c.num_res_blocks[i] = m, n
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            abi.check(self._lib.mmada_vq_create(C.byref(c), C.byref(self._handle)), "mmada_vq_create")
            st = abi.stream_ptr()
            # MAGVITv2 checkpoints hold encoder.*, decoder.* and quantize.* (buffers): only decoder.* is consumed;
            # a bare decoder state dict (no prefix) is accepted as well.
            keys = list(state_dict.keys())
            prefixed = any(k.startswith("decoder.") for k in keys)
            for k in keys:
                if prefixed and not k.startswith("decoder."):
                    continue
                t = state_dict[k].to(device=self.device, dtype=torch.float32).contiguous()
                abi.check(self._lib.mmada_vq_bind(self._handle, k.encode(), t.data_ptr(), t.numel(), st), f"bind {k}")
            torch.cuda.current_stream().synchronize()  # the staged tensors `t` may be freed now
        missing = self._lib.mmada_vq_num_unbound(self._handle)
        if missing:
            raise KeyError(f"{missing} decoder tensors missing from the state dict")
        self._ws = None

    @classmethod
    def from_state_dict(cls, state_dict, config=None, **kw):
        return cls(state_dict, config, **kw)

    @classmethod
    def from_pretrained(cls, path: str, **kw):
        """Directory with (optional) config.json and *.safetensors or pytorch_model.bin holding the reference's keys."""
        config = None
        cj = os.path.join(path, "config.json")
        if os.path.exists(cj):
            with open(cj) as f:
                raw = json.load(f)
            config = {k: raw[k] for k in DEFAULT_CFG if k in raw}
        sd: Dict[str, torch.Tensor] = {}
        st_files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
        if st_files:
            from safetensors.torch import load_file

            for fn in st_files:
                sd.update(load_file(os.path.join(path, fn)))
        elif os.path.exists(os.path.join(path, "pytorch_model.bin")):
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no *.safetensors / pytorch_model.bin under {path}")
        return cls(sd, config, **kw)

    # reference call-surface no-ops (inference.py:57-59)
    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def requires_grad_(self, _flag: bool = False):
        return self

    @property
    def scale(self) -> int:
        return 2 ** (len(self.config["ch_mult"]) - 1)

    @torch.no_grad()
    def decode_code(self, codebook_indices: torch.Tensor, shape=None) -> torch.Tensor:
        """[B, N] int64 codebook ids -> [B, out_ch, h*scale, w*scale] fp32 (modeling_magvitv2.py:429-433)."""
        idx = codebook_indices.to(device=self.device, dtype=torch.long).contiguous()
        if idx.dim() != 2:
            raise ValueError("codebook_indices must be [B, N]")
        B, n = idx.shape
        if shape is None:
            h = w = int(math.sqrt(n))  # :209-210
        else:
            h, w = shape
        if h * w != n:
            raise ValueError(f"{n} tokens do not form a {h}x{w} grid")
        with torch.cuda.device(self.device):
            need = self._lib.mmada_vq_workspace_bytes(self._handle, B, h, w)
            if self._ws is None or self._ws.numel() < need + 256:
                self._ws = None
                self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            base = (self._ws.data_ptr() + 255) // 256 * 256
            out = torch.empty((B, self.config["out_ch"], h * self.scale, w * self.scale), dtype=torch.float32,
                              device=self.device)
            abi.check(self._lib.mmada_vq_decode_code(self._handle, idx.data_ptr(), B, h, w, base, need, out.data_ptr(),
                                                     abi.stream_ptr()), "mmada_vq_decode_code")
        return out

    def encode(self, *_a, **_k):
        raise NotImplementedError("the encoder direction is outside the sampler hot path (SURVEY.md §8f)")

    get_code = encode

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            self._lib.mmada_vq_destroy(h)
            self._handle = None


def to_uint8_image(images: torch.Tensor) -> torch.Tensor:
    """inference.py:129-130: clamp((x + 1) / 2, 0, 1) * 255 -> [B, H, W, C] uint8 (truncating cast)."""
    x = torch.clamp((images + 1.0) / 2.0, min=0.0, max=1.0) * 255.0
    return x.permute(0, 2, 3, 1).to(torch.uint8)
