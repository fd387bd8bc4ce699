"""Mirror of the reference's utils/image_utils.py for the TI2TI path (MMaDA-Parallel-A).

List helpers: `calculate_vq_params` (:95-111), `add_break_line` (:149-157).
Pixel <-> token functions around the image tokenizer: `decode_vq_to_image` (:13-75), `encode_img_with_breaks` (:159-173),
`encode_img_with_paint` (:175-284) — same signatures and token conventions (newline 126084, <boi>/<eoi> 126349/126350, VQ
offset 126356, mask 126336).  `vqvae` is `mmada_parallel_amd.VQModel` (the MI355X mirror of `diffusers.VQModel`); the two
`diffusers.image_processor.VaeImageProcessor` calls the reference makes (preprocess / postprocess with do_normalize=False)
are restated with PIL + numpy below, since diffusers is neither vendored in the reference nor installed here."""
from typing import List, Optional, Tuple

MASK_TOKEN_ID, NEWLINE_TOKEN_ID, BOI_TOKEN_ID, EOI_TOKEN_ID, VQ_OFFSET = 126336, 126084, 126349, 126350, 126356


def calculate_vq_params(image_height: int, image_width: int, vae_scale: int = 16):
    """-> (seq_len, newline_every, token_grid_height, token_grid_width)."""
    token_grid_height = image_height // vae_scale
    token_grid_width = image_width // vae_scale
    return token_grid_height * token_grid_width, token_grid_width, token_grid_height, token_grid_width


def add_break_line(sequence: List[int], H: int, W: int, new_number: int = 0) -> List[int]:
    """Append `new_number` after every row of W tokens of an H x W grid."""
    result: List[int] = []
    for i in range(H):
        result.extend(sequence[i * W:(i + 1) * W] + [new_number])
    return result


# ---- VaeImageProcessor(vae_scale_factor=f, do_normalize=False) as the reference uses it --------------------------------------
def pil_to_unit_tensor(img, multiple: int):
    """preprocess(): resize down to multiples of `multiple` (lanczos, diffusers' default resample), HWC uint8 -> [1, 3, H, W]
    float32 in [0, 1]; no normalisation."""
    import numpy as np
    import torch
    from PIL import Image

    w, h = img.size
    w, h = w - w % multiple, h - h % multiple
    img = img.resize((w, h), resample=Image.LANCZOS)
    arr = np.asarray(img).astype(np.float32) / 255.0
    if arr.ndim == 2:
        arr = arr[..., None]
    return torch.from_numpy(arr[None]).permute(0, 3, 1, 2).contiguous()


def unit_tensor_to_pil(x):
    """postprocess(output_type="pil") without denormalisation: [B, 3, H, W] in [0, 1] -> list of PIL images (round to uint8)."""
    from PIL import Image

    arr = (x.detach().cpu().permute(0, 2, 3, 1).float().numpy() * 255).round().astype("uint8")
    return [Image.fromarray(a.squeeze(-1) if a.shape[-1] == 1 else a) for a in arr]


def _scale_of(vqvae) -> int:
    return 2 ** (len(vqvae.config.block_out_channels) - 1)


def decode_vq_to_image(vq_codes, save_path: Optional[str] = None, vae_ckpt: Optional[str] = None, image_height: int = 512,
                       image_width: int = 512, vqvae=None):
    """VQ codes [B, seq_len] in [0, codebook_size) -> PIL image of the first sequence (reference :13-75)."""
    if vqvae is None:
        from ..vqmodel import VQModel

        vqvae = VQModel.from_pretrained(vae_ckpt, subfolder="vqvae", device=vq_codes.device)
    scale = _scale_of(vqvae)
    gh, gw = image_height // scale, image_width // scale
    if vq_codes.shape[1] != gh * gw:
        raise ValueError(f"VQ codes length mismatch: {vq_codes.shape[1]} != {gh * gw} "
                         f"for image size ({image_height},{image_width}) with scale {scale}")
    grid = vq_codes.view(vq_codes.shape[0], gh, gw).long()
    recon = vqvae.decode(grid, force_not_quantize=True,
                         shape=(vq_codes.shape[0], gh, gw, vqvae.config.latent_channels)).sample.clip(0, 1)
    img = unit_tensor_to_pil(recon)[0]
    if save_path is not None:
        img.save(save_path)
    return img


def _encode_indices(img, vqvae, scale: int):
    x = pil_to_unit_tensor(img.convert("RGB"), scale).to(vqvae.device)
    latents = vqvae.encode(x).latents
    _, _, lat_h, lat_w = latents.shape
    return x, vqvae.quantize(latents)[2][2].reshape(-1), lat_h, lat_w


def encode_img_with_breaks(img, vqvae, vae_scale_factor: int = 16) -> List[int]:
    """PIL image -> [<boi>] + rows of (VQ index + 126356) each followed by the newline token + [<eoi>] (reference :159-173)."""
    _, idx, lat_h, lat_w = _encode_indices(img, vqvae, vae_scale_factor)
    body = add_break_line((idx + VQ_OFFSET).tolist(), lat_h, lat_w, new_number=NEWLINE_TOKEN_ID)
    return [BOI_TOKEN_ID] + body + [EOI_TOKEN_ID]


def encode_img_with_paint(img, vqvae, *, mask_h_ratio: float = 1, mask_w_ratio: float = 0.2, gray_value: int = 127,
                          downsample_mode: str = "area", dilate_latent_k: int = 0, mask_mode: str = "inpainting") -> Tuple[List[int], object]:
    """In/out-painting input (reference :175-284): the ORIGINAL image is tokenised; latent positions under the centred
    rectangle (inpainting) or outside it (outpainting) become the mask token.  Returns (tokens with newlines, gray preview)."""
    import torch
    import torch.nn.functional as F
    from PIL import Image, ImageDraw

    if mask_mode not in ("inpainting", "outpainting"):
        raise AssertionError("mask_mode must be 'inpainting' or 'outpainting'")
    img = img.convert("RGB")
    W, H = img.size
    mh, mw = int(round(H * mask_h_ratio)), int(round(W * mask_w_ratio))
    top, left = (H - mh) // 2, (W - mw) // 2
    gray = (gray_value,) * 3
    if mask_mode == "inpainting":
        vis = img.copy()
        ImageDraw.Draw(vis).rectangle([left, top, left + mw, top + mh], fill=gray)
    else:
        vis = Image.new("RGB", (W, H), gray)
        vis.paste(img.crop((left, top, left + mw, top + mh)), (left, top))
    x, idx, lat_h, lat_w = _encode_indices(img, vqvae, _scale_of(vqvae))
    Hp, Wp = x.shape[-2:]
    inside = torch.zeros((1, 1, Hp, Wp), dtype=torch.float32, device=x.device)
    t, l = int(round(top * Hp / H)), int(round(left * Wp / W))
    inside[:, :, t:t + int(round(mh * Hp / H)), l:l + int(round(mw * Wp / W))] = 1.0
    masked_px = inside if mask_mode == "inpainting" else 1.0 - inside
    mode = downsample_mode if downsample_mode in ("nearest", "area", "bilinear") else "area"
    m = F.interpolate(masked_px, size=(lat_h, lat_w), mode=mode)
    m = (m > 0.5) if mode == "area" else (m >= 0.5)
    if dilate_latent_k > 0:
        m = F.max_pool2d(m.float(), kernel_size=2 * dilate_latent_k + 1, stride=1, padding=dilate_latent_k) > 0.5
    m = m.reshape(-1)
    tokens = torch.where(m, torch.full_like(idx, MASK_TOKEN_ID), idx + VQ_OFFSET)
    return add_break_line(tokens.tolist(), lat_h, lat_w, NEWLINE_TOKEN_ID), vis


def decode_step_preview(sampled_ids, masked_cells, vqvae, image_height: int = 512, image_width: int = 512):
    """The picture the reference's streaming sampler shows after an image step (app.py:310-339): ALL sampled codes of the
    step decoded, with a translucent gray square over every latent cell the step re-masked.
    sampled_ids: [1, N] codes in [0, codebook_size) — the third item `generate_ti2ti_stepwise` yields;
    masked_cells: indices in [0, N) of the cells still masked after the step (e.g. where the yielded ids are MASK)."""
    from PIL import ImageDraw

    img = decode_vq_to_image(sampled_ids[:1], None, None, image_height, image_width, vqvae)
    cells = [int(c) for c in masked_cells]
    if cells:
        scale = _scale_of(vqvae)
        gw = image_width // scale
        ph, pw = image_height // (image_height // scale), image_width // gw
        img = img.copy()
        draw = ImageDraw.Draw(img, "RGBA")
        for c in cells:
            y, x = (c // gw) * ph, (c % gw) * pw
            draw.rectangle([x, y, x + pw, y + ph], fill=(128, 128, 128, 120))
    return img
