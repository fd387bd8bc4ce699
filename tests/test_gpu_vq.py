"""GPU parity tests of the MAGVITv2 token -> pixel decode (csrc/vq_decoder.hip) through the C-ABI.

The reference runs this network in fp32 (MMaDA-Parallel-M/inference.py:56-59), so does the HIP path; the comparison
is against the fp32 CPU oracle (oracle/vq_oracle.py, pinned to the reference by tests/golden/vq_decode.npz) and
against that fixture itself.  Tolerances: one convolution / GroupNorm 1e-5 of the output range (fp32 sums in a
different order); the whole 5-level decoder (37 convolutions, 32 GroupNorms) 2e-5 of the output range (measured 5e-6) and at most
1 level in the final uint8 image.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import GOLDEN
from mmada_parallel_amd import MAGVITv2, abi, synth
from mmada_parallel_amd.vq import to_uint8_image
from oracle import vq_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


CONV_CASES = [
    # B, H, W, Cin, Cout, k, ups, bias, resid          what it covers
    (2, 12, 10, 128, 256, 3, 0, True, True),      # MFMA path, ragged pixel tile (240 rows), bias + residual
    (1, 9, 7, 256, 128, 3, 1, True, False),       # upsample folded into the A-tile addressing
    (2, 8, 8, 256, 128, 1, 0, True, False),       # nin_shortcut (1x1)
    (1, 16, 16, 512, 512, 3, 0, False, False),    # four N tiles, no bias (attention GEMM use)
    (1, 5, 6, 96, 200, 3, 0, True, True),         # Cout not a multiple of the 128 tile
    (2, 8, 8, 13, 512, 3, 0, True, False),        # conv_in: direct kernel, 16 output channels per thread
    (2, 8, 8, 13, 13, 1, 0, True, False),         # post_quant_conv
    (1, 20, 24, 128, 3, 3, 0, True, False),       # conv_out geometry (NHWC output here)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_torch_fp32(case):
    B, H, W, Ci, Co, k, ups, has_bias, has_resid = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) * (Ci * k * k) ** -0.5
    b = torch.randn(Co, generator=g) if has_bias else None
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(xin, w, b, padding=k // 2)
    r = torch.randn_like(ref) if has_resid else None
    if r is not None:
        ref = ref + r
    xd, wd = nhwc(x).to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV)
    bd = b.to(DEV) if b is not None else None
    rd = nhwc(r).to(DEV) if r is not None else None
    out = torch.empty((B, H << ups, W << ups, Co), dtype=torch.float32, device=DEV)
    abi.check(abi.lib().mmada_vq_conv2d(xd.data_ptr(), wd.data_ptr(), abi.ptr(bd), abi.ptr(rd), out.data_ptr(), B, H, W,
                                        Ci, Co, k, ups, abi.stream_ptr()), "conv2d")
    got = out.cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item() + 1e-6, f"max err {err:.3e} of range {ref.abs().max().item():.3f}"


@pytest.mark.parametrize("H,W", [(16, 16), (10, 12), (9, 7)])
def test_downsample_conv_matches_torch(H, W):
    # Downsample.forward (common_modules.py:83-90): F.pad(x, (0,1,0,1)) then Conv2d(3, stride 2, padding 0)
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randn(2, 128, H, W, generator=g)
    w = torch.randn(128, 128, 3, 3, generator=g) * 0.03
    b = torch.randn(128, generator=g)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2, padding=0)
    Ho, Wo = ref.shape[2:]
    xd, wd, bd = nhwc(x).to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV), b.to(DEV)
    out = torch.empty((2, Ho, Wo, 128), dtype=torch.float32, device=DEV)
    abi.check(abi.lib().mmada_vq_conv2d(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), 0, out.data_ptr(), 2, H, W, 128, 128,
                                        3, -1, abi.stream_ptr()), "conv2d")
    assert (out.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_conv2d_residual_may_alias_output():
    # ResnetBlock without nin_shortcut: x = x + conv2(h) is done in place on x
    g = torch.Generator().manual_seed(3)
    h = torch.randn(1, 128, 10, 10, generator=g)
    x = torch.randn(1, 128, 10, 10, generator=g)
    w = torch.randn(128, 128, 3, 3, generator=g) * 0.03
    ref = x + F.conv2d(h, w, None, padding=1)
    hd, wd, xd = nhwc(h).to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV), nhwc(x).to(DEV)
    abi.check(abi.lib().mmada_vq_conv2d(hd.data_ptr(), wd.data_ptr(), 0, xd.data_ptr(), xd.data_ptr(), 1, 10, 10, 128,
                                        128, 3, 0, abi.stream_ptr()), "conv2d")
    assert (xd.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("C_,HW,swish", [(128, 100, 1), (256, 4096, 1), (512, 1024, 0), (128, 70001, 1), (1024, 64, 1)])
def test_group_norm_swish_matches_torch(C_, HW, swish):
    B = 2
    g = torch.Generator().manual_seed(C_ + HW)
    x = torch.randn(B, C_, HW, generator=g) * 3.0 + 0.7  # non-zero mean: E[x^2] - E[x]^2 must not cancel in fp32
    gamma, beta = torch.randn(C_, generator=g), torch.randn(C_, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    if swish:
        ref = ref * torch.sigmoid(ref)
    xd = x.permute(0, 2, 1).contiguous().to(DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    out = torch.empty_like(xd)
    lib = abi.lib()
    scratch = torch.empty(lib.mmada_vq_group_norm_scratch_bytes(B), dtype=torch.uint8, device=DEV)
    abi.check(lib.mmada_vq_group_norm(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), out.data_ptr(), scratch.data_ptr(),
                                      B, HW, C_, swish, abi.stream_ptr()), "group_norm")
    got = out.cpu().permute(0, 2, 1)
    assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("name", ["tiny", "full"])
def test_decode_code_matches_reference_fixture_and_oracle(name):
    z = np.load(os.path.join(GOLDEN, "vq_decode.npz"))
    cfg = synth.VQ_CFG_TINY if name == "tiny" else synth.VQ_CFG_M
    sd = synth.synthetic_vq_state_dict(cfg, int(z[name + "_seed"]))
    idx = torch.from_numpy(z[name + "_idx"])
    vq = MAGVITv2.from_state_dict(sd, cfg, device=DEV)
    img = vq.decode_code(idx.to(DEV)).cpu()
    scale = 2 ** (len(cfg["ch_mult"]) - 1)
    hz = int(idx.shape[1] ** 0.5)
    assert img.shape == (idx.shape[0], 3, hz * scale, hz * scale) and img.dtype == torch.float32
    # (1) the fixture recorded from the reference's own VQGANDecoder
    ref = torch.from_numpy(z[name + "_out"])
    got = img if name == "tiny" else img[:, :, ::4, ::4]
    rng = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    print(f"vq decode [{name}] vs reference fixture: max err {err:.3e} (range {rng:.3f})")
    assert err <= 2e-5 * rng
    # (2) every pixel against the oracle evaluated here, and the uint8 image the reference would save
    full = vq_oracle.decode_code(sd, cfg, idx)
    assert (img - full).abs().max().item() <= 2e-5 * rng
    a, b = to_uint8_image(img).int(), vq_oracle.to_uint8_image(full).int()
    assert (a - b).abs().max().item() <= 1 and (a != b).float().mean().item() < 2e-3


def test_decode_code_is_per_sample_and_deterministic():
    cfg = synth.VQ_CFG_TINY
    sd = synth.synthetic_vq_state_dict(cfg, 11)
    vq = MAGVITv2.from_state_dict({"decoder." + k: v for k, v in sd.items()}, cfg, device=DEV)  # checkpoint-style keys
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, 8192, (3, 8 * 4), generator=g).to(DEV)
    a = vq.decode_code(idx, shape=(8, 4))  # non-square grid
    assert a.shape == (3, 3, 16, 8)
    b = vq.decode_code(idx, shape=(8, 4))
    assert torch.equal(a, b)
    one = vq.decode_code(idx[1:2], shape=(8, 4))
    assert torch.equal(one[0], a[1])
    ref = vq_oracle.decode_code(sd, cfg, idx.cpu(), shape=(8, 4))
    assert (a.cpu() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()


def test_vq_bind_errors_are_loud():
    cfg = synth.VQ_CFG_TINY
    sd = synth.synthetic_vq_state_dict(cfg, 1)
    bad = dict(sd)
    del bad["mid.attn_1.q.weight"]
    with pytest.raises(KeyError):
        MAGVITv2.from_state_dict(bad, cfg, device=DEV)
    bad = dict(sd)
    bad["conv_in.weight"] = torch.zeros(7)
    with pytest.raises(abi.MmadaError):
        MAGVITv2.from_state_dict(bad, cfg, device=DEV)
    bad = dict(sd)
    bad["not_a_decoder_tensor"] = torch.zeros(3)
    with pytest.raises(abi.MmadaError):
        MAGVITv2.from_state_dict(bad, cfg, device=DEV)
    vq = MAGVITv2.from_state_dict(sd, cfg, device=DEV)
    with pytest.raises(abi.MmadaError):
        vq.decode_code(torch.zeros((1, 9), dtype=torch.long))  # 3x3 grid: not a multiple of 32 positions


@pytest.mark.parametrize("name", ["tiny", "full"])
def test_get_code_matches_reference_fixture_and_oracle(name):
    z = np.load(os.path.join(GOLDEN, "vq_encode.npz"))
    cfg = synth.VQ_ENC_CFG_TINY if name == "tiny" else synth.VQ_ENC_CFG_M
    seed = int(z[name + "_seed"])
    sd = synth.synthetic_vq_state_dict(cfg, seed)
    B, res = (2, 16) if name == "tiny" else (1, 512)
    img = synth.synthetic_image(B, res, res, seed=200 + seed)
    vq = MAGVITv2.from_state_dict({"encoder." + k: v for k, v in sd.items()}, cfg, device=DEV)
    idx, zz = vq.get_code(img.to(DEV), return_z=True)
    idx, zz = idx.cpu(), zz.cpu()
    zref = torch.from_numpy(z[name + "_z"])
    err = (zz - zref).abs().max().item()
    print(f"vq encode [{name}] z vs reference fixture: max err {err:.3e} (range {zref.abs().max().item():.3f})")
    assert err <= 2e-5 * zref.abs().max().item()
    # indices are the signs of z: bit-exact wherever no |z| is within the fp32 tolerance of zero
    iref = torch.from_numpy(z[name + "_idx"])
    sure = (zref.abs() > 1e-4).all(1).reshape(B, -1)
    assert sure.float().mean().item() > 0.95
    assert torch.equal(idx[sure], iref[sure])
    # and they are exactly the bit pattern of OUR z everywhere (LFQuantizer.get_indices)
    power = 2 ** torch.arange(cfg["z_channels"] - 1, -1, -1)
    mine = (power.reshape(1, -1, 1, 1) * (zz > 0).long()).sum(1).reshape(B, -1)
    assert torch.equal(idx, mine)
    # oracle evaluated on this machine
    oi, oz = vq_oracle.get_code(sd, cfg, img, return_z=True)
    assert (zz - oz).abs().max().item() <= 2e-5 * oz.abs().max().item()
    zq, idx2 = vq.encode(img.to(DEV))
    assert torch.equal(idx2.cpu(), idx) and torch.equal(zq.cpu(), torch.where(zz > 0, 1.0, -1.0))


def test_encode_decode_roundtrip_runs_on_one_model():
    """A full MAGVITv2 checkpoint layout (encoder.* + decoder.* + quantize.* buffers) builds both directions."""
    enc, dec = synth.VQ_ENC_CFG_TINY, synth.VQ_CFG_TINY
    sd = {"encoder." + k: v for k, v in synth.synthetic_vq_state_dict(enc, 3).items()}
    sd.update({"decoder." + k: v for k, v in synth.synthetic_vq_state_dict(dec, 4).items()})
    sd["quantize.embedding"] = torch.zeros(8192, 13)  # buffer of the reference checkpoint: ignored
    vq = MAGVITv2(sd, dec, device=DEV, encoder_config=enc)
    img = synth.synthetic_image(1, 64, 32, seed=9).to(DEV)
    idx = vq.get_code(img)
    assert idx.shape == (1, 32 * 16) and idx.min().item() >= 0 and idx.max().item() < 8192
    out = vq.decode_code(idx, shape=(32, 16))
    assert out.shape == (1, 3, 64, 32) and torch.isfinite(out).all()
