"""CPU ORACLE (test infrastructure — never imported by the product path).  **PARITY UNPINNED.**

Plain-PyTorch fp32 restatement of `diffusers.VQModel` as MMaDA-Parallel-A drives it
(/root/reference/MMaDA-Parallel-A/utils/image_utils.py:13-75 decode_vq_to_image, :159-173 encode_img_with_breaks,
inference.py:94-96).  The class lives in the third-party requirement diffusers==0.34.0 (requirements.txt pin), which is NOT
vendored under /root/reference and is not installed in this image, so no golden vectors can be produced from it: this
file restates the PUBLISHED architecture of that release and is pinned to nothing —

    VQModel.encode / decode / lookup_from_codebook       diffusers/models/autoencoders/vq_model.py
    Encoder, Decoder, VectorQuantizer                    diffusers/models/autoencoders/vae.py
    DownEncoderBlock2D, UpDecoderBlock2D, UNetMidBlock2D diffusers/models/unets/unet_2d_blocks.py
    ResnetBlock2D, Downsample2D(padding=0), Upsample2D   diffusers/models/resnet.py, downsampling.py, upsampling.py
    Attention (one head, residual, GroupNorm) + SDPA     diffusers/models/attention_processor.py

(GroupNorm eps 1e-6, SiLU, nearest 2x upsampling followed by a 3x3 convolution, F.pad(x, (0,1,0,1)) before the stride-2
convolution, argmin of torch.cdist for the codebook index).  State-dict keys are the checkpoint's own.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def _gn(x, sd, p, groups):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _conv(x, sd, p, padding, stride=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def resnet(x, sd, p, groups):
    h = _conv(F.silu(_gn(x, sd, p + ".norm1", groups)), sd, p + ".conv1", 1)
    h = _conv(F.silu(_gn(h, sd, p + ".norm2", groups)), sd, p + ".conv2", 1)
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(x, sd, p + ".conv_shortcut", 0)
    return x + h  # output_scale_factor = 1


def mid_attention(x, sd, p, groups):
    B, Cc, H, W = x.shape
    h = _gn(x.view(B, Cc, H * W), sd, p + ".group_norm", groups).transpose(1, 2)          # [B, HW, C]
    q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    a = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]          # one head of width C
    a = F.linear(a, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return x + a.transpose(1, 2).reshape(B, Cc, H, W)


def mid_block(x, sd, p, cfg):
    g = cfg["norm_num_groups"]
    x = resnet(x, sd, p + ".resnets.0", g)
    if cfg.get("mid_block_add_attention", True):
        x = mid_attention(x, sd, p + ".attentions.0", g)
    return resnet(x, sd, p + ".resnets.1", g)


@torch.no_grad()
def encode(sd: Dict[str, torch.Tensor], cfg: dict, x: torch.Tensor) -> torch.Tensor:
    """VQModel.encode(x).latents: [B, 3, H, W] -> [B, vq_embed_dim, H/f, W/f]."""
    g, L = cfg["norm_num_groups"], len(cfg["block_out_channels"])
    h = _conv(x, sd, "encoder.conv_in", 1)
    for i in range(L):
        for j in range(cfg["layers_per_block"]):
            h = resnet(h, sd, f"encoder.down_blocks.{i}.resnets.{j}", g)
        if i != L - 1:
            h = _conv(F.pad(h, (0, 1, 0, 1), mode="constant", value=0), sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", 0, stride=2)
    h = mid_block(h, sd, "encoder.mid_block", cfg)
    h = _conv(F.silu(_gn(h, sd, "encoder.conv_norm_out", g)), sd, "encoder.conv_out", 1)
    return _conv(h, sd, "quant_conv", 0)


@torch.no_grad()
def nearest_code(sd: Dict[str, torch.Tensor], latents: torch.Tensor) -> torch.Tensor:
    """VectorQuantizer.forward's min_encoding_indices (flat [B*h*w])."""
    z = latents.permute(0, 2, 3, 1).contiguous().view(-1, latents.shape[1])
    return torch.argmin(torch.cdist(z, sd["quantize.embedding.weight"]), dim=1)


@torch.no_grad()
def decode_codes(sd: Dict[str, torch.Tensor], cfg: dict, indices: torch.Tensor) -> torch.Tensor:
    """VQModel.decode(indices [B, h, w], force_not_quantize=True, shape=(B, h, w, C)).sample with lookup_from_codebook."""
    g, L = cfg["norm_num_groups"], len(cfg["block_out_channels"])
    B, hz, wz = indices.shape
    zq = F.embedding(indices.reshape(-1), sd["quantize.embedding.weight"]).view(B, hz, wz, -1).permute(0, 3, 1, 2).contiguous()
    h = _conv(zq, sd, "post_quant_conv", 0)
    h = _conv(h, sd, "decoder.conv_in", 1)
    h = mid_block(h, sd, "decoder.mid_block", cfg)
    for i in range(L):
        for j in range(cfg["layers_per_block"] + 1):
            h = resnet(h, sd, f"decoder.up_blocks.{i}.resnets.{j}", g)
        if i != L - 1:
            h = _conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", 1)
    return _conv(F.silu(_gn(h, sd, "decoder.conv_norm_out", g)), sd, "decoder.conv_out", 1)
