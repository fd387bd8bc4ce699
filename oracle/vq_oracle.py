"""CPU ORACLE (test infrastructure — never imported by the product path).

Plain-PyTorch fp32 restatement of the MAGVITv2 token <-> pixel paths of MMaDA-Parallel-M:
    MAGVITv2.get_code                    /root/reference/MMaDA-Parallel-M/models/modeling_magvitv2.py:422-427
    VQGANEncoder.forward                 modeling_magvitv2.py:143-171 (module tree :62-141), Downsample common_modules.py:83-90
    MAGVITv2.decode_code                 /root/reference/MMaDA-Parallel-M/models/modeling_magvitv2.py:429-433
    LFQuantizer.get_codebook_entry       modeling_magvitv2.py:208-221  (embedding table built at :187-195)
    VQGANDecoder.forward                 modeling_magvitv2.py:369-406  (module tree :277-367)
    ResnetBlock.forward                  models/common_modules.py:337-357
    AttnBlock.forward                    models/common_modules.py:187-211
    Upsample.forward                     models/common_modules.py:36-40
    Normalize / nonlinearity             models/common_modules.py:16-24
The reference runs the VQ model in fp32 (inference.py:56-59: `.to(device)` without a dtype).
Parity is PINNED: tests/test_oracle_golden.py checks this file against tests/golden/vq_decode*.npz, produced by
importing and running the reference's own VQGANDecoder / LFQuantizer (oracle/gen_golden.py: gen_vq_decode).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F


def lfq_codebook_entry(indices: torch.Tensor, codebook_dim: int = 13, shape: Optional[Sequence[int]] = None) -> torch.Tensor:
    # modeling_magvitv2.py:187-195 (table: bit c of the index, MSB first, mapped to +-1) and :208-221
    b, n = indices.shape
    h, w = (int(math.sqrt(n)),) * 2 if shape is None else shape
    shifts = torch.arange(codebook_dim - 1, -1, -1, dtype=torch.long)
    z = ((indices.reshape(-1, 1) >> shifts) & 1).float() * 2 - 1
    return z.view(b, h, w, codebook_dim).permute(0, 3, 1, 2).contiguous()


def swish(x: torch.Tensor) -> torch.Tensor:  # common_modules.py:16-18
    return x * torch.sigmoid(x)


def group_norm(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str) -> torch.Tensor:  # common_modules.py:21-24
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def conv(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str, padding: int) -> torch.Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=1, padding=padding)


def resnet_block(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str) -> torch.Tensor:
    # common_modules.py:337-357 with temb=None, dropout p=0
    h = conv(swish(group_norm(x, sd, p + ".norm1")), sd, p + ".conv1", 1)
    h = conv(swish(group_norm(h, sd, p + ".norm2")), sd, p + ".conv2", 1)
    if (p + ".nin_shortcut.weight") in sd:  # in_channels != out_channels, use_conv_shortcut=False
        x = conv(x, sd, p + ".nin_shortcut", 0)
    return x + h


def attn_block(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str) -> torch.Tensor:
    # common_modules.py:187-211: single head over the h*w positions, scale c^-0.5
    h_ = group_norm(x, sd, p + ".norm")
    q, k, v = conv(h_, sd, p + ".q", 0), conv(h_, sd, p + ".k", 0), conv(h_, sd, p + ".v", 0)
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + conv(h_, sd, p + ".proj_out", 0)


@torch.no_grad()
def decoder_forward(sd: Dict[str, torch.Tensor], cfg: dict, z: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """VQGANDecoder.forward (modeling_magvitv2.py:369-406).  `sd` uses the decoder's own state-dict keys
    (no "decoder." prefix).  cfg: ch_mult, num_res_blocks (per level, low index = highest resolution)."""
    n_levels = len(cfg["ch_mult"])
    h = conv(z, sd, "post_quant_conv", 0)
    h = conv(h, sd, "conv_in", 1)
    h = resnet_block(h, sd, "mid.block_1")
    h = attn_block(h, sd, "mid.attn_1")
    h = resnet_block(h, sd, "mid.block_2")
    if taps is not None:
        taps["mid"] = h
    for lvl in reversed(range(n_levels)):
        for b in range(cfg["num_res_blocks"][lvl]):
            h = resnet_block(h, sd, f"up.{lvl}.block.{b}")
            # attn_resolutions=[5] never matches a level resolution (modeling_magvitv2.py:281,337-338): no attention here
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")  # common_modules.py:37
            h = conv(h, sd, f"up.{lvl}.upsample.conv", 1)
        if taps is not None:
            taps[f"up{lvl}"] = h
    h = swish(group_norm(h, sd, "norm_out"))
    return conv(h, sd, "conv_out", 1)


@torch.no_grad()
def decode_code(sd: Dict[str, torch.Tensor], cfg: dict, indices: torch.Tensor, shape=None) -> torch.Tensor:
    """MAGVITv2.decode_code (modeling_magvitv2.py:429-433): [B, N] int64 -> [B, 3, 16*h, 16*w] fp32."""
    return decoder_forward(sd, cfg, lfq_codebook_entry(indices, cfg.get("z_channels", 13), shape))


@torch.no_grad()
def encoder_forward(sd: Dict[str, torch.Tensor], cfg: dict, x: torch.Tensor) -> torch.Tensor:
    """VQGANEncoder.forward (modeling_magvitv2.py:143-171): [B, 3, H, W] -> [B, z_channels, H/f, W/f]."""
    n_levels = len(cfg["ch_mult"])
    h = conv(x, sd, "conv_in", 1)
    for lvl in range(n_levels):
        for b in range(cfg["num_res_blocks"][lvl]):
            h = resnet_block(h, sd, f"down.{lvl}.block.{b}")
        if lvl != n_levels - 1:  # Downsample.forward, common_modules.py:83-90: pad right/bottom, 3x3 stride 2
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            p = f"down.{lvl}.downsample.conv"
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], stride=2, padding=0)
    h = resnet_block(h, sd, "mid.block_1")
    h = attn_block(h, sd, "mid.attn_1")
    h = resnet_block(h, sd, "mid.block_2")
    h = swish(group_norm(h, sd, "norm_out"))
    h = conv(h, sd, "conv_out", 1)
    return conv(h, sd, "quant_conv", 0)


@torch.no_grad()
def get_code(sd: Dict[str, torch.Tensor], cfg: dict, pixel_values: torch.Tensor, return_z: bool = False):
    """MAGVITv2.get_code (modeling_magvitv2.py:422-427): sign quantisation (:241-243: z_q = +1 where z > 0 else -1) and
    get_indices (:201-206: sum of 2^(dim-1-c) over the channels with z_q > 0) -> [B, N] int64."""
    z = encoder_forward(sd, cfg, pixel_values)
    dim = z.shape[1]
    power = 2 ** torch.arange(dim - 1, -1, -1)
    idx = (power.reshape(1, -1, 1, 1) * (z > 0).long()).sum(1).reshape(z.shape[0], -1)
    return (idx, z) if return_z else idx


def to_uint8_image(x: torch.Tensor) -> torch.Tensor:
    """inference.py:129-130: clamp((x+1)/2, 0, 1) * 255 -> HWC uint8 (truncating cast)."""
    x = torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0) * 255.0
    return x.permute(0, 2, 3, 1).to(torch.uint8)
