"""Host mirror of `diffusers.VQModel` as MMaDA-Parallel-A uses it (SURVEY.md §8f rank 1, A variant).

The reference loads the image tokenizer with `VQModel.from_pretrained(vae_ckpt, subfolder="vqvae").to(device)`
(inference.py:94-96) and touches exactly this surface (utils/image_utils.py:13-75,159-173):

    scale   = 2 ** (len(vqvae.config.block_out_channels) - 1)
    latents = vqvae.encode(x).latents                                   # x: [B, 3, H, W] in [0, 1]
    ids     = vqvae.quantize(latents)[2][2]                             # nearest codebook row per latent position
    recon   = vqvae.decode(ids_bhw, force_not_quantize=True, shape=(B, h, w, vqvae.config.latent_channels)).sample
    vqvae.device

`diffusers==0.34.0` is a third-party requirement that is NOT vendored in the reference tree, so the architecture is restated
from its published source (autoencoders/vq_model.py, autoencoders/vae.py) and parity is UNPINNED: the tests compare against
oracle/vqmodel_oracle.py, which carries the same caveat.  The arithmetic runs in libmmada_mi355x.so (csrc/vq_decoder.hip,
mmada_vq_create_vqmodel); there is no PyTorch fallback — without the HIP library or a GPU the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import abi

# diffusers VQModel.__init__ defaults (autoencoders/vq_model.py)
DEFAULT_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=[64], layers_per_block=1, act_fn="silu",
                      latent_channels=3, sample_size=32, num_vq_embeddings=256, norm_num_groups=32, vq_embed_dim=None,
                      scaling_factor=0.18215, norm_type="group", mid_block_add_attention=True, lookup_from_codebook=False,
                      force_upcast=False)


class VqModelCfg(C.Structure):
    """struct mmada_vqmodel_cfg (include/mmada_mi355x.h)."""

    _fields_ = [("n_levels", C.c_int32), ("block_out_channels", C.c_int32 * 8), ("layers_per_block", C.c_int32),
                ("latent_channels", C.c_int32), ("vq_embed_dim", C.c_int32), ("num_vq_embeddings", C.c_int32),
                ("image_channels", C.c_int32), ("mid_block_add_attention", C.c_int32), ("norm_num_groups", C.c_int32)]


class VQModel:
    """Drop-in for the reference's use of `diffusers.VQModel` (inference only, fp32 like the reference)."""

    def __init__(self, config: dict, state_dict: Dict[str, torch.Tensor], device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("VQModel (MI355X) needs a GPU: there is no CPU fallback")
        cfg = dict(DEFAULT_CONFIG)
        cfg.update({k: v for k, v in config.items() if not k.startswith("_")})
        if cfg["act_fn"] not in ("silu", "swish") or cfg["norm_type"] != "group":
            raise NotImplementedError("only act_fn='silu' and norm_type='group' are on the MI355X path")
        if cfg["in_channels"] != cfg["out_channels"]:
            raise NotImplementedError("in_channels != out_channels")
        cfg["vq_embed_dim"] = cfg["vq_embed_dim"] if cfg["vq_embed_dim"] is not None else cfg["latent_channels"]
        self.config = SimpleNamespace(**cfg)
        self.device = torch.device(device if device is not None else "cuda:0")
        self.dtype = torch.float32
        self._lib = abi.lib()
        self._ws = None
        c = VqModelCfg()
        c.n_levels = len(cfg["block_out_channels"])
        if not 1 <= c.n_levels <= 8:
            raise ValueError("1..8 block_out_channels")
        for i, v in enumerate(cfg["block_out_channels"]):
            c.block_out_channels[i] = v
        c.layers_per_block, c.latent_channels, c.vq_embed_dim = cfg["layers_per_block"], cfg["latent_channels"], cfg["vq_embed_dim"]
        c.num_vq_embeddings, c.image_channels = cfg["num_vq_embeddings"], cfg["in_channels"]
        c.mid_block_add_attention, c.norm_num_groups = int(bool(cfg["mid_block_add_attention"])), cfg["norm_num_groups"]
        self._dec = self._build(c, False, state_dict)
        self._enc = self._build(c, True, state_dict) if any(k.startswith("encoder.") for k in state_dict.keys()) else None

    def _build(self, c: VqModelCfg, encoder: bool, sd) -> C.c_void_p:
        h = C.c_void_p()
        own = ("encoder.", "quant_conv.", "quantize.embedding.") if encoder else ("decoder.", "post_quant_conv.", "quantize.embedding.")
        with torch.cuda.device(self.device):
            abi.check(self._lib.mmada_vq_create_vqmodel(C.byref(c), int(encoder), C.byref(h)), "mmada_vq_create_vqmodel")
            st = abi.stream_ptr()
            for k in sd.keys():
                if not k.startswith(own):
                    continue
                t = sd[k].to(device=self.device, dtype=torch.float32).contiguous()
                abi.check(self._lib.mmada_vq_bind(h, k.encode(), t.data_ptr(), t.numel(), st), f"bind {k}")
            torch.cuda.current_stream().synchronize()  # the staged tensors may be freed now
        missing = self._lib.mmada_vq_num_unbound(h)
        if missing:
            self._lib.mmada_vq_destroy(h)
            raise KeyError(f"{missing} {'encoder' if encoder else 'decoder'} tensors of the VQModel are missing from the state dict")
        return h

    # ---- loading ---------------------------------------------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, config: dict, state_dict, **kw):
        return cls(config, state_dict, **kw)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, **kw):
        """diffusers layout: <path>/<subfolder>/config.json + diffusion_pytorch_model.safetensors (or .bin)."""
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, "config.json")) as f:
            config = json.load(f)
        sd: Dict[str, torch.Tensor] = {}
        files = sorted(f for f in os.listdir(root) if f.endswith(".safetensors"))
        if files:
            from safetensors.torch import load_file

            for fn in files:
                sd.update(load_file(os.path.join(root, fn)))
        else:
            bins = sorted(f for f in os.listdir(root) if f.endswith(".bin"))
            if not bins:
                raise FileNotFoundError(f"no *.safetensors / *.bin under {root}")
            for fn in bins:
                sd.update(torch.load(os.path.join(root, fn), map_location="cpu"))
        return cls(config, sd, **kw)

    def to(self, device=None, *_a, **_k):
        """`.to(device)` of the reference (inference.py:95): a no-op for the device the weights already live on."""
        if device is None or isinstance(device, torch.dtype):
            return self
        d = torch.device(device)
        if d.type != "cuda":
            raise NotImplementedError("the MI355X VQModel has no CPU path")
        if d.index is not None and d.index != (self.device.index or 0):
            raise NotImplementedError("the weights live on the device the model was built on; pass device= at construction")
        return self

    def eval(self):
        return self

    def requires_grad_(self, _flag=False):
        return self

    # ---- the three calls of the reference ----------------------------------------------------------------------------------
    def _workspace(self, handle, B, hz, wz):
        need = self._lib.mmada_vq_workspace_bytes(handle, B, hz, wz)
        if self._ws is None or self._ws.numel() < need + 256:
            self._ws = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        return (self._ws.data_ptr() + 255) // 256 * 256, need

    @property
    def scale(self) -> int:
        return 2 ** (len(self.config.block_out_channels) - 1)

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """VQModel.encode: Encoder + quant_conv -> latents [B, vq_embed_dim, H/f, W/f] (no quantisation)."""
        if self._enc is None:
            raise RuntimeError("this VQModel was built without encoder.* weights")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B, Cin, H, W = x.shape
        f = self.scale
        if Cin != self.config.in_channels or H % f or W % f:
            raise ValueError(f"encode: expected [B, {self.config.in_channels}, H, W] with H, W multiples of {f}")
        hz, wz, D = H // f, W // f, self.config.vq_embed_dim
        ws, nb = self._workspace(self._enc, B, hz, wz)
        idx = torch.empty((B, hz * wz), dtype=torch.int64, device=self.device)
        z = torch.empty((B, hz * wz, D), dtype=torch.float32, device=self.device)
        abi.check(self._lib.mmada_vq_get_code(self._enc, x.data_ptr(), B, H, W, ws, nb, idx.data_ptr(), z.data_ptr(),
                                              abi.stream_ptr()), "mmada_vq_get_code")
        latents = z.view(B, hz, wz, D).permute(0, 3, 1, 2).contiguous()
        # quantize(latents) on this very (unmodified) tensor reuses the indices computed alongside
        self._last = (latents, idx, latents._version)
        return SimpleNamespace(latents=latents) if return_dict else (latents,)

    def quantize(self, latents: torch.Tensor):
        """VectorQuantizer.forward: (z_q, loss, (perplexity, min_encodings, min_encoding_indices)); the reference reads only
        [2][2] (utils/image_utils.py:168).  Indices are flat [B*h*w] like diffusers' (sane_index_shape=False)."""
        last = getattr(self, "_last", None)
        B, D, hz, wz = latents.shape
        if last is not None and last[0] is latents and latents._version == last[2]:
            idx = last[1].reshape(-1)
        else:
            z = latents.to(device=self.device, dtype=torch.float32).permute(0, 2, 3, 1).contiguous()
            idx = torch.empty(B * hz * wz, dtype=torch.int64, device=self.device)
            abi.check(self._lib.mmada_vq_nearest_code(self._dec, z.data_ptr(), B * hz * wz, idx.data_ptr(), abi.stream_ptr()),
                      "mmada_vq_nearest_code")
        return None, None, (None, None, idx)

    def decode(self, h: torch.Tensor, force_not_quantize: bool = False, return_dict: bool = True, shape=None):
        """VQModel.decode.  Integer `h` [B, hz, wz] with force_not_quantize=True is the reference's call (lookup_from_codebook:
        codebook rows -> post_quant_conv -> Decoder); float latents are quantised first, as diffusers does without the flag."""
        if h.dtype.is_floating_point:
            if force_not_quantize:
                raise NotImplementedError("decoding un-quantised float latents is not on the MI355X path")
            B, _, hz, wz = h.shape
            idx = self.quantize(h)[2][2].view(B, hz, wz)
        else:
            if not (force_not_quantize and self.config.lookup_from_codebook):
                raise ValueError("integer codes need force_not_quantize=True and a lookup_from_codebook config, as in the reference")
            idx = h.to(device=self.device, dtype=torch.int64)
            if shape is not None:
                idx = idx.reshape(shape[0], shape[1], shape[2])
            B, hz, wz = idx.shape
        idx = idx.contiguous()
        f = self.scale
        ws, nb = self._workspace(self._dec, B, hz, wz)
        out = torch.empty((B, self.config.out_channels, hz * f, wz * f), dtype=torch.float32, device=self.device)
        abi.check(self._lib.mmada_vq_decode_code(self._dec, idx.data_ptr(), B, hz, wz, ws, nb, out.data_ptr(), abi.stream_ptr()),
                  "mmada_vq_decode_code")
        return SimpleNamespace(sample=out, commit_loss=None) if return_dict else (out,)

    def __del__(self):
        try:
            for h in (getattr(self, "_dec", None), getattr(self, "_enc", None)):
                if h:
                    self._lib.mmada_vq_destroy(h)
            self._dec = self._enc = None
        except Exception:
            pass
