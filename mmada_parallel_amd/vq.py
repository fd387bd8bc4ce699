"""Host mirror of the reference's MAGVITv2 VQ model for the token -> pixel direction.

Same call surface as MMaDA-Parallel-M (models/modeling_magvitv2.py:408-433, used at inference.py:56-59,127-130):

    vq_model = MAGVITv2.from_pretrained(path).to(device); vq_model.requires_grad_(False); vq_model.eval()
    images = vq_model.decode_code(output_image_ids)          # [B, 3, H, W] fp32, unclamped

    image_tokens = vq_model.get_code(image)                  # [B, N] int64 (inference.py:79)

The arithmetic runs in libmmada_mi355x.so (csrc/vq_decoder.hip) through the C-ABI of include/mmada_mi355x.h; there
is no PyTorch fallback — without the HIP library or a GPU the constructor raises.
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
DEFAULT_ENC_CFG = dict(ch=128, ch_mult=[1, 2, 2, 4, 4], num_res_blocks=[4, 3, 4, 3, 4], z_channels=13, in_ch=3)


class VqCfg(C.Structure):
    """struct mmada_vq_cfg (include/mmada_mi355x.h)."""

    _fields_ = [("ch", C.c_int32), ("n_levels", C.c_int32), ("ch_mult", C.c_int32 * 8),
                ("num_res_blocks", C.c_int32 * 8), ("z_channels", C.c_int32), ("out_ch", C.c_int32)]


class MAGVITv2:
    """`state_dict` may hold `decoder.*` and/or `encoder.*` tensors (a MAGVITv2 checkpoint; `quantize.*` buffers are
    ignored), or the bare keys of one of the two networks.  Whichever network is present is built."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], config: Optional[dict] = None,
                 device: Optional[torch.device] = None, encoder_config: Optional[dict] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("MAGVITv2 (MI355X) needs a GPU: there is no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:0")
        self._lib = abi.lib()
        self._handle = self._enc = None
        self._ws = None
        keys = list(state_dict.keys())
        prefixed = any(k.startswith(("decoder.", "encoder.")) for k in keys)
        dec_keys = [k for k in keys if k.startswith("decoder.")] if prefixed else \
            (keys if any(k.startswith("up.") for k in keys) else [])
        enc_keys = [k for k in keys if k.startswith("encoder.")] if prefixed else \
            (keys if any(k.startswith("down.") for k in keys) else [])
        if not dec_keys and not enc_keys:
            raise KeyError("state dict holds neither decoder.* nor encoder.* tensors")
        self.config = dict(DEFAULT_CFG)
        self.config.update(config or {})
        self.encoder_config = dict(DEFAULT_ENC_CFG)
        self.encoder_config.update(encoder_config or {})
        if dec_keys:
            self._handle = self._build(self.config, self.config["out_ch"], False, state_dict, dec_keys)
        if enc_keys:
            self._enc = self._build(self.encoder_config, self.encoder_config["in_ch"], True, state_dict, enc_keys)

    def _build(self, cfg, io_ch, encoder, state_dict, keys):
        if len(cfg["ch_mult"]) != len(cfg["num_res_blocks"]) or not 1 <= len(cfg["ch_mult"]) <= 8:
            raise ValueError("ch_mult / num_res_blocks must have the same length (1..8)")
        c = VqCfg()
        c.ch, c.n_levels, c.z_channels, c.out_ch = cfg["ch"], len(cfg["ch_mult"]), cfg["z_channels"], io_ch
        for i, (m, n) in enumerate(zip(cfg["ch_mult"], cfg["num_res_blocks"])):
            c.ch_mult[i], c.num_res_blocks[i] = m, n
        h = C.c_void_p()
        create = self._lib.mmada_vq_create_encoder if encoder else self._lib.mmada_vq_create
        with torch.cuda.device(self.device):
            abi.check(create(C.byref(c), C.byref(h)), "mmada_vq_create")
            st = abi.stream_ptr()
            for k in keys:
                t = state_dict[k].to(device=self.device, dtype=torch.float32).contiguous()
                abi.check(self._lib.mmada_vq_bind(h, k.encode(), t.data_ptr(), t.numel(), st), f"bind {k}")
            torch.cuda.current_stream().synchronize()  # the staged tensors `t` may be freed now
        missing = self._lib.mmada_vq_num_unbound(h)
        if missing:
            self._lib.mmada_vq_destroy(h)
            raise KeyError(f"{missing} {'encoder' if encoder else 'decoder'} tensors missing from the state dict")
        return h

    def _workspace(self, handle, B, h, w):
        need = self._lib.mmada_vq_workspace_bytes(handle, B, h, w)
        if self._ws is None or self._ws.numel() < need + 256:
            self._ws = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        return (self._ws.data_ptr() + 255) // 256 * 256, need

    @classmethod
    def from_state_dict(cls, state_dict, config=None, **kw):
        if config is not None and "in_ch" in config:  # an encoder configuration
            return cls(state_dict, None, encoder_config=config, **kw)
        return cls(state_dict, config, **kw)

    @classmethod
    def from_pretrained(cls, path: str, **kw):
        """Directory with (optional) config.json and *.safetensors or pytorch_model.bin holding the reference's keys."""
        # The reference's MAGVITv2.__init__ takes no arguments (modeling_magvitv2.py:409-417), so its config.json
        # carries no architecture: the class defaults apply unless "decoder" / "encoder" sub-dicts are present.
        config = enc_config = None
        cj = os.path.join(path, "config.json")
        if os.path.exists(cj):
            with open(cj) as f:
                raw = json.load(f)
            if isinstance(raw.get("decoder"), dict):
                config = {k: raw["decoder"][k] for k in DEFAULT_CFG if k in raw["decoder"]}
            if isinstance(raw.get("encoder"), dict):
                enc_config = {k: raw["encoder"][k] for k in DEFAULT_ENC_CFG if k in raw["encoder"]}
        kw.setdefault("encoder_config", enc_config)
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
        if self._handle is None:
            raise RuntimeError("this MAGVITv2 was built without decoder weights")
        with torch.cuda.device(self.device):
            base, need = self._workspace(self._handle, B, h, w)
            out = torch.empty((B, self.config["out_ch"], h * self.scale, w * self.scale), dtype=torch.float32,
                              device=self.device)
            abi.check(self._lib.mmada_vq_decode_code(self._handle, idx.data_ptr(), B, h, w, base, need, out.data_ptr(),
                                                     abi.stream_ptr()), "mmada_vq_decode_code")
        return out

    @torch.no_grad()
    def get_code(self, pixel_values: torch.Tensor, return_z: bool = False):
        """[B, 3, H, W] fp32 in [-1, 1] -> [B, (H/f)*(W/f)] int64 codebook ids (modeling_magvitv2.py:422-427).
        return_z also returns the pre-quantisation encoder output [B, z_channels, H/f, W/f] (parity tap)."""
        if self._enc is None:
            raise RuntimeError("this MAGVITv2 was built without encoder weights")
        x = pixel_values.to(device=self.device, dtype=torch.float32).contiguous()
        if x.dim() != 4 or x.shape[1] != self.encoder_config["in_ch"]:
            raise ValueError("pixel_values must be [B, in_ch, H, W]")
        B, _, H, W = x.shape
        f = 2 ** (len(self.encoder_config["ch_mult"]) - 1)
        zc = self.encoder_config["z_channels"]
        with torch.cuda.device(self.device):
            base, need = self._workspace(self._enc, B, max(1, H // f), max(1, W // f))
            idx = torch.empty((B, (H // f) * (W // f)), dtype=torch.long, device=self.device)
            z = torch.empty((B, (H // f) * (W // f), zc), dtype=torch.float32, device=self.device) if return_z else None
            abi.check(self._lib.mmada_vq_get_code(self._enc, x.data_ptr(), B, H, W, base, need, idx.data_ptr(), abi.ptr(z),
                                                  abi.stream_ptr()), "mmada_vq_get_code")
        if return_z:
            return idx, z.view(B, H // f, W // f, zc).permute(0, 3, 1, 2).contiguous()
        return idx

    @torch.no_grad()
    def encode(self, pixel_values: torch.Tensor, return_loss: bool = False):
        """(quantized_states [B, z_channels, h, w] of +-1, codebook_indices [B, N]) — modeling_magvitv2.py:415-420."""
        idx = self.get_code(pixel_values)
        B, _, H, W = pixel_values.shape
        f = 2 ** (len(self.encoder_config["ch_mult"]) - 1)
        zc = self.encoder_config["z_channels"]
        shifts = torch.arange(zc - 1, -1, -1, device=idx.device)
        zq = (((idx.unsqueeze(-1) >> shifts) & 1).float() * 2 - 1).view(B, H // f, W // f, zc).permute(0, 3, 1, 2)
        return zq.contiguous(), idx

    def __del__(self):
        for name in ("_handle", "_enc"):
            h = getattr(self, name, None)
            if h is not None and h.value:
                self._lib.mmada_vq_destroy(h)
                setattr(self, name, None)


def to_uint8_image(images: torch.Tensor) -> torch.Tensor:
    """inference.py:129-130: clamp((x + 1) / 2, 0, 1) * 255 -> [B, H, W, C] uint8 (truncating cast)."""
    x = torch.clamp((images + 1.0) / 2.0, min=0.0, max=1.0) * 255.0
    return x.permute(0, 2, 3, 1).to(torch.uint8)
