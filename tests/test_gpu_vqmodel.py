"""GPU tests of the A-variant image tokenizer (mmada_parallel_amd.VQModel over csrc/vq_decoder.hip, C-ABI
mmada_vq_create_vqmodel / mmada_vq_decode_code / mmada_vq_get_code / mmada_vq_nearest_code) and of the pixel <-> token
helpers of utils/image_utils.py.

PARITY UNPINNED: the reference imports `diffusers.VQModel` (third-party, not vendored, not installed), so the comparison
is against oracle/vqmodel_oracle.py — a restatement of the published diffusers 0.34 architecture that nothing can pin
offline — on seeded synthetic checkpoints with the checkpoint's own key names.  fp32 like the reference; tolerance as for
the pinned MAGVITv2 path: 2e-5 of the output range for a whole network."""
import pytest
import torch

from helpers import host_threads
from mmada_parallel_amd import VQModel, synth
from mmada_parallel_amd.utils import image_utils as iu
from oracle import vqmodel_oracle as vo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _unit_image(B, H, W, seed):
    x = synth.synthetic_image(B, H, W, seed=seed)
    return (x - x.amin()) / (x.amax() - x.amin())


def _check_codes(sd, lat_ref, idx_got, tag):
    """Indices must equal the oracle's wherever the nearest and second-nearest codebook rows are separated by more than the
    latent error allows; elsewhere the chosen row must be (within tolerance) as near as the oracle's."""
    z = lat_ref.permute(0, 2, 3, 1).reshape(-1, lat_ref.shape[1])
    d = torch.cdist(z, sd["quantize.embedding.weight"])
    top2 = d.topk(2, dim=1, largest=False)
    idx_ref = top2.indices[:, 0]
    margin = top2.values[:, 1] - top2.values[:, 0]
    same = idx_got.cpu().reshape(-1) == idx_ref
    clear = margin > 1e-4 * d.mean()
    print(f"{tag}: {int(same.sum())}/{same.numel()} indices equal, {int((~clear).sum())} near-ties")
    assert bool(same[clear].all())
    chosen = d.gather(1, idx_got.cpu().reshape(-1, 1))[:, 0]
    assert bool((chosen - top2.values[:, 0] <= 1e-4 * d.mean()).all())


def test_tiny_vqmodel_decode_encode_quantize_vs_oracle():
    cfg = synth.VQMODEL_CFG_TINY
    sd = synth.synthetic_vqmodel_state_dict(cfg, seed=1)
    sd["quantize.embedding.weight"] = sd["quantize.embedding.weight"] * 0.5
    vq = VQModel.from_state_dict(cfg, sd, device=DEV)
    assert 2 ** (len(vq.config.block_out_channels) - 1) == 2 and vq.config.vq_embed_dim == 8
    g = torch.Generator().manual_seed(3)
    codes = torch.randint(0, cfg["num_vq_embeddings"], (2, 16, 16), generator=g)
    got = vq.decode(codes.to(DEV), force_not_quantize=True, shape=(2, 16, 16, cfg["latent_channels"])).sample.cpu()
    ref = vo.decode_codes(sd, cfg, codes)
    rng = ref.abs().max().item()
    print(f"tiny VQModel decode: max err {(got - ref).abs().max().item():.3e} (range {rng:.3f})")
    assert got.shape == (2, 3, 32, 32) and (got - ref).abs().max().item() <= 2e-5 * rng
    x = _unit_image(2, 32, 32, seed=4)
    lat = vq.encode(x.to(DEV)).latents
    lat_ref = vo.encode(sd, cfg, x)
    lrng = lat_ref.abs().max().item()
    print(f"tiny VQModel encode: max err {(lat.cpu() - lat_ref).abs().max().item():.3e} (range {lrng:.3f})")
    assert lat.shape == lat_ref.shape and (lat.cpu() - lat_ref).abs().max().item() <= 2e-5 * lrng
    idx = vq.quantize(lat)[2][2]
    assert idx.shape == (2 * 16 * 16,) and idx.dtype == torch.int64
    _check_codes(sd, lat_ref, idx, "tiny quantize (fused with encode)")
    # quantize() on latents that did not come from encode(): the stand-alone nearest-code entry point, same answer
    idx2 = vq.quantize(lat.clone())[2][2]
    assert torch.equal(idx2, idx)
    lat.mul_(-1.0)                                  # modified in place after encode(): the cached indices must not be reused
    assert torch.equal(vq.quantize(lat)[2][2], vq.quantize(lat.clone())[2][2]) and not torch.equal(vq.quantize(lat)[2][2], idx)
    lat.mul_(-1.0)
    # float latents through decode(): quantised first, like diffusers without force_not_quantize
    rec = vq.decode(lat).sample
    assert torch.equal(rec, vq.decode(idx.view(2, 16, 16), force_not_quantize=True).sample)
    with pytest.raises(ValueError):
        vq.decode(codes.to(DEV))  # integer codes without force_not_quantize
    bad = dict(sd)
    bad.pop("decoder.conv_out.bias")
    with pytest.raises(KeyError):
        VQModel.from_state_dict(cfg, bad, device=DEV)


def test_a_variant_vqmodel_at_512_vs_oracle():
    """The f16 / 8192-code geometry of the reference's token arithmetic: 32 x 32 codes -> 512 x 512 pixels (every pixel against
    the oracle), and a 256 x 256 image -> 16 x 16 latents + codes."""
    host_threads()
    cfg = synth.VQMODEL_CFG_A
    sd = synth.synthetic_vqmodel_state_dict(cfg, seed=2)
    vq = VQModel.from_state_dict(cfg, sd, device=DEV)
    g = torch.Generator().manual_seed(5)
    codes = torch.randint(0, 8192, (1, 32, 32), generator=g)
    got = vq.decode(codes.to(DEV), force_not_quantize=True, shape=(1, 32, 32, 64)).sample.cpu()
    ref = vo.decode_codes(sd, cfg, codes)
    rng = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    print(f"A VQModel decode 32x32 -> 512x512: max err {err:.3e} (range {rng:.3f})")
    assert got.shape == (1, 3, 512, 512) and err <= 2e-5 * rng
    a = (got.clip(0, 1) * 255).round()
    b = (ref.clip(0, 1) * 255).round()
    assert (a - b).abs().max().item() <= 1 and (a != b).float().mean().item() < 2e-3
    x = _unit_image(1, 256, 256, seed=6)
    lat = vq.encode(x.to(DEV)).latents
    lat_ref = vo.encode(sd, cfg, x)
    lerr, lrng = (lat.cpu() - lat_ref).abs().max().item(), lat_ref.abs().max().item()
    print(f"A VQModel encode 256x256 -> 16x16x64: max err {lerr:.3e} (range {lrng:.3f})")
    assert lat.shape == (1, 64, 16, 16) and lerr <= 2e-5 * lrng
    _check_codes(sd, lat_ref, vq.quantize(lat)[2][2], "A quantize")
    # timing of the benchmark geometry (decode of one 512 x 512 image), reported only
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cd = codes.to(DEV)
    vq.decode(cd, force_not_quantize=True)
    e0.record()
    for _ in range(5):
        vq.decode(cd, force_not_quantize=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"A VQModel decode 512x512: {e0.elapsed_time(e1) / 5:.2f} ms")


def test_pixel_token_helpers_follow_the_reference_conventions():
    from PIL import Image

    cfg = synth.VQMODEL_CFG_TINY
    sd = synth.synthetic_vqmodel_state_dict(cfg, seed=1)
    vq = VQModel.from_state_dict(cfg, sd, device=DEV)
    f = 2
    arr = (_unit_image(1, 33, 65, seed=8)[0].permute(1, 2, 0).numpy() * 255).round().astype("uint8")
    img = Image.fromarray(arr)
    toks = iu.encode_img_with_breaks(img, vq, vae_scale_factor=f)
    lat_h, lat_w = 32 // f, 64 // f            # preprocess resizes down to multiples of the scale (16 x 32 latent cells)
    assert len(toks) == 2 + lat_h * (lat_w + 1) and toks[0] == 126349 and toks[-1] == 126350
    body = toks[1:-1]
    assert all(body[(r + 1) * (lat_w + 1) - 1] == 126084 for r in range(lat_h))
    codes = [t - 126356 for i, t in enumerate(body) if (i + 1) % (lat_w + 1)]
    assert all(0 <= c < cfg["num_vq_embeddings"] for c in codes)
    # the same indices as the model's own encode + quantize on the same preprocessed pixels
    x = iu.pil_to_unit_tensor(img, f)
    assert x.shape == (1, 3, 32, 64) and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    idx = vq.quantize(vq.encode(x.to(DEV)).latents)[2][2].tolist()
    assert codes == idx
    out = iu.decode_vq_to_image(torch.tensor([codes], device=DEV), image_height=32, image_width=64, vqvae=vq)
    assert out.size == (64, 32)
    with pytest.raises(ValueError):
        iu.decode_vq_to_image(torch.tensor([codes[:-1]], device=DEV), image_height=32, image_width=64, vqvae=vq)
    # in / out-painting: the rectangle's latent cells (and only they) become the mask token; the others keep their codes
    ptoks, vis = iu.encode_img_with_paint(img, vq, mask_h_ratio=0.5, mask_w_ratio=0.5, mask_mode="inpainting")
    otoks, _ = iu.encode_img_with_paint(img, vq, mask_h_ratio=0.5, mask_w_ratio=0.5, mask_mode="outpainting")
    assert vis.size == img.size and len(ptoks) == len(otoks) == lat_h * (lat_w + 1)
    n_in = sum(t == 126336 for t in ptoks)
    n_out = sum(t == 126336 for t in otoks)
    assert n_in + n_out == lat_h * lat_w and 0 < n_in < n_out
    cells = [t for i, t in enumerate(ptoks) if (i + 1) % (lat_w + 1)]
    assert all(t == 126336 or t - 126356 == c for t, c in zip(cells, codes))


def test_stepwise_preview_decodes_and_marks_the_masked_cells():
    """app.py:310-339: every sampled code decoded, a translucent gray square over each re-masked latent cell."""
    import numpy as np

    cfg = synth.VQMODEL_CFG_TINY
    vq = VQModel.from_state_dict(cfg, synth.synthetic_vqmodel_state_dict(cfg, seed=1), device=DEV)
    g = torch.Generator().manual_seed(9)
    codes = torch.randint(0, cfg["num_vq_embeddings"], (1, 16 * 32), generator=g).to(DEV)
    plain = np.asarray(iu.decode_step_preview(codes, [], vq, 32, 64)).astype(int)
    marked = np.asarray(iu.decode_step_preview(codes, [0, 33], vq, 32, 64)).astype(int)     # cells (0,0) and (1,1), 2 x 2 pixels each
    assert plain.shape == marked.shape == (32, 64, 3)
    diff = (plain != marked).any(-1)
    assert diff[:3, :3].any() and diff[2:5, 2:5].any() and not diff[8:, :].any() and not diff[:, 8:].any()
