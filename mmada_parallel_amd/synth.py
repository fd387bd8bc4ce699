"""Synthetic weights and token jobs at the reference's shapes.

No checkpoint, tokenizer or VQ-VAE is available offline, so parity tests and the benchmark run on seeded random
weights with the reference's state-dict keys (model/modeling_llada.py:1097-1131, 864-893, 564-575) and on a token
sequence assembled exactly like inference.py:129-161.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

MASK, NEW_LINE, BOI, EOI, BOA, EOA = 126336, 126084, 126349, 126350, 126354, 126355
TEXT_VOCAB, CODEBOOK = 126356, 8192

CFG_8B = dict(d_model=4096, n_heads=32, n_kv_heads=32, n_layers=32, mlp_hidden_size=12288, vocab_size=134656,
              embedding_size=134656, rms_norm_eps=1e-5, rope_theta=500000.0, max_sequence_length=4096)
CFG_TINY = dict(d_model=256, n_heads=2, n_kv_heads=2, n_layers=2, mlp_hidden_size=512, vocab_size=134656,
                embedding_size=134656, rms_norm_eps=1e-5, rope_theta=500000.0, max_sequence_length=1024)


def full_config(cfg: dict) -> dict:
    """The keyword set LLaDAConfig needs for this architecture (block llama, silu, rms, rope fp32, no bias)."""
    out = dict(activation_type="silu", block_type="llama", rope=True, rope_full_precision=True,
               layer_norm_type="rms", weight_tying=False, include_bias=False, include_qkv_bias=False,
               attention_dropout=0.0, residual_dropout=0.0, embedding_dropout=0.0, alibi=False,
               attention_layer_norm=False, scale_logits=False, input_emb_norm=False, flash_attention=False,
               multi_query_attention=None, block_group_size=1, init_device="cpu",
               text_vocab_size=TEXT_VOCAB, codebook_size=CODEBOOK, mask_token_id=MASK)
    out.update(cfg)
    return out


def synthetic_state_dict(cfg: dict, seed: int = 0, device: str = "cpu", dtype=torch.bfloat16,
                         logit_std: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """Seeded random weights: linear/embedding N(0, 0.02²), norms 1+N(0, 0.02²), LM head N(0, 1/d)."""
    d, F, V = cfg["d_model"], cfg["mlp_hidden_size"], cfg.get("embedding_size") or cfg["vocab_size"]
    hd = d // cfg["n_heads"]
    kv = (cfg.get("n_kv_heads") or cfg["n_heads"]) * hd
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def rn(shape, std, mean=0.0):
        t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std + mean
        return t.to(dtype)

    sd: Dict[str, torch.Tensor] = {}
    p = "model.transformer."
    sd[p + "wte.weight"] = rn((V, d), 0.02)
    for i in range(cfg["n_layers"]):
        b = f"{p}blocks.{i}."
        sd[b + "attn_norm.weight"] = rn((d,), 0.02, 1.0)
        sd[b + "ff_norm.weight"] = rn((d,), 0.02, 1.0)
        sd[b + "q_proj.weight"] = rn((d, d), 0.02)
        sd[b + "k_proj.weight"] = rn((kv, d), 0.02)
        sd[b + "v_proj.weight"] = rn((kv, d), 0.02)
        sd[b + "attn_out.weight"] = rn((d, d), 0.02)
        sd[b + "ff_proj.weight"] = rn((F, d), 0.02)
        sd[b + "up_proj.weight"] = rn((F, d), 0.02)
        sd[b + "ff_out.weight"] = rn((d, F), 0.02)
    sd[p + "ln_f.weight"] = rn((d,), 0.02, 1.0)
    sd[p + "ff_out.weight"] = rn((V, d), logit_std if logit_std is not None else d ** -0.5)
    return sd


def synthetic_state_dict_peaked(cfg: dict, delta: int, seed: int = 0, device: str = "cpu", dtype=torch.bfloat16, nfreq: int = 16,
                                amp: float = 6.0, c0: float = 1.0, small: float = 0.6, beta: float = 5.0, lns: float = 0.25) -> Dict[str, torch.Tensor]:
    """A PEAKED synthetic checkpoint: seeded random weights with one planted circuit, so that the sampler's decisions have
    margins far above bf16 rounding noise and decision-level parity can be asserted (and can fail).  With the flat weights of
    synthetic_state_dict every post-CFG arg-max is a near-tie among thousands of classes and no implementation — the
    reference on another CPU included — reproduces a free-running trajectory.

    The circuit ("copy the token `delta` positions back"):
      * every embedding carries the same component c0 along a unit vector e0, its random content is orthogonal to e0;
      * block 0's q_proj / k_proj read (almost) only that component, so queries and keys are the same vector at every
        position up to the rotary rotation; the planted phases (`nfreq` highest rotary frequencies, amplitude `amp`) put the
        score maximum at key position = query position - delta: content-independent, sharp attention;
      * block 0's v_proj ignores e0 (values carry token content), so the block writes M·norm(wte[token at i - delta]) into
        position i, M = attn_out · v_proj;
      * the LM head row of token v is the unit direction M·norm(wte[v]) times `beta` and a log-normal factor (so that the
        confidences of different positions differ), plus the usual random head.
    Everything else — the other blocks, every MLP — is the random recipe scaled by `small`: real arithmetic on every GEMM, a
    perturbation of the planted signal.  Masked positions (all the same MASK embedding) thus predict the token delta
    positions to their left: distinct, position-dependent, peaked predictions.  Measured on CPU (tools/dbg/peaked_explore2.py):
    bf16 and fp32 evaluation agree on 100 % of post-CFG image arg-maxima, text arg-maxima and first-commit sets."""
    import math

    d, H = cfg["d_model"], cfg["n_heads"]
    hd = d // H
    kvh = cfg.get("n_kv_heads") or H
    sd = synthetic_state_dict(cfg, seed=seed, device=device, dtype=dtype)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1234)
    e0 = torch.randn(d, generator=g, device=device)
    e0 /= e0.norm()
    p = "model.transformer."
    wte = sd[p + "wte.weight"].float()
    wte = wte - (wte @ e0)[:, None] * e0[None, :] + c0 * e0[None, :]
    sd[p + "wte.weight"] = wte.to(dtype)
    for i in range(cfg["n_layers"]):
        for n in ("v_proj", "attn_out", "ff_proj", "up_proj", "ff_out"):
            if i == 0 and n in ("v_proj", "attn_out"):
                continue
            k = f"{p}blocks.{i}.{n}.weight"
            sd[k] = (sd[k].float() * small).to(dtype)
    inv = 1.0 / (cfg.get("rope_theta", 10000.0) ** (torch.arange(0, hd, 2, dtype=torch.float64) / hd))
    qv, kv = torch.zeros(hd, dtype=torch.float64), torch.zeros(hd, dtype=torch.float64)
    for f in range(nfreq):   # rotary pair (f, f + hd/2), model/modeling_llada.py:402-435
        ang = delta * inv[f].item()
        kv[f] = amp
        qv[f], qv[f + hd // 2] = amp * math.cos(ang), -amp * math.sin(ang)
    qv, kv = qv.float().to(device), kv.float().to(device)
    b0 = p + "blocks.0."
    wq = qv.repeat(H)[:, None] * e0[None, :]
    wk = kv.repeat(kvh)[:, None] * e0[None, :]
    sd[b0 + "q_proj.weight"] = (wq + sd[b0 + "q_proj.weight"].float() * small).to(dtype)
    sd[b0 + "k_proj.weight"] = (wk + sd[b0 + "k_proj.weight"].float() * small).to(dtype)
    wv = sd[b0 + "v_proj.weight"].float()
    sd[b0 + "v_proj.weight"] = ((wv - (wv @ e0)[:, None] * e0[None, :]) * 2.0).to(dtype)
    sd[b0 + "attn_out.weight"] = (sd[b0 + "attn_out.weight"].float() * 2.0).to(dtype)
    # the LM head reads the copied token: row v = beta * lognormal * unit(M · norm(wte[v])) (+ the random head)
    an = sd[b0 + "attn_norm.weight"].float()
    lnx = wte / wte.pow(2).mean(-1, keepdim=True).add(cfg.get("rms_norm_eps", 1e-5)).sqrt() * an[None, :]
    wo, wv2 = sd[b0 + "attn_out.weight"].float(), sd[b0 + "v_proj.weight"].float()
    if kvh != H:   # grouped-query attention: a kv head's values reach every query head of its group
        wv2 = wv2.view(kvh, 1, hd, d).expand(kvh, H // kvh, hd, d).reshape(H * hd, d)
    y = (lnx @ wv2.t()) @ wo.t()
    y = y / y.norm(dim=-1, keepdim=True)
    scale = torch.exp(torch.randn(y.shape[0], generator=g, device=device) * lns)
    head = sd[p + "ff_out.weight"].float()
    sd[p + "ff_out.weight"] = (y * (beta * scale)[:, None] * (4.0 / math.sqrt(d)) + head).to(dtype)
    return sd


CFG_PEAKED = dict(d_model=1024, n_heads=8, n_kv_heads=8, n_layers=4, mlp_hidden_size=2048, vocab_size=134656,
                  embedding_size=134656, rms_norm_eps=1e-5, rope_theta=500000.0, max_sequence_length=4096)


def peaked_delta(job: dict) -> int:
    """The copy distance used with synthetic_state_dict_peaked for a synthetic_job: the masked output-image and text positions
    all copy from inside the conditioning image span (distinct image codes)."""
    return job["image_start"] - 40


# ---- the M variant's free-running job on a peaked checkpoint (BASELINE configs[3] geometry) --------------------------------
M_PEAKED_KW = dict(text_cfg=2.5, image_cfg=4.0, text_steps=24, image_steps=8, image_temperature=1.0, text_temperature=0.0)
M_PEAKED_BETA = 2.0   # LM-head gain of the planted circuit: the text CFG combine multiplies logit margins by up to 2.5 and the
                      # image combine by 5, so the gain is lower than the A recipe's 5 (keeps fp64 text confidences below 1.0:
                      # no exact ties for the reference's torch.topk to order)


def m_peaked_job(seed: int = 7) -> dict:
    """MMaDA-Parallel-M interleave_generate inputs at the bench's configs[3] geometry (MMaDA-Parallel-M/inference.py:79,
    113-127 with stand-in special ids): prompt = <|interleave|> <|soi|> 1024 image codes <|eoi|> 40 text tokens (P = 1067);
    the sampler appends <|soi|> 1024 x MASK <|eoi|> <bos> 255 x MASK: L = 2349.  The unconditional prompt has token 0 in place
    of the image codes and other text (as tools/m_end_to_end.py / bench.py --config 3 build it).  delta = 1066: every masked
    output-image position copies the input-image code at the same raster position."""
    g = torch.Generator().manual_seed(seed)
    N, T, n_text = 1024, 256, 40
    img = torch.randint(0, CODEBOOK, (N,), generator=g) + TEXT_VOCAB
    text = torch.randint(0, 100000, (n_text,), generator=g)
    un_text = torch.randint(0, 100000, (n_text,), generator=g)
    head, eoi = torch.tensor([126340, 126084]), torch.tensor([126085])
    inp = torch.cat([head, img, eoi, text])
    unc = torch.cat([head, torch.zeros_like(img), eoi, un_text])
    P = int(inp.shape[0])
    return dict(input_ids=inp, uncond_input_ids=unc, N=N, T=T, P=P, L=P + 1 + N + 1 + T, delta=P - 1, soi=126084, eoi=126085,
                bos=126080, text_vocab=TEXT_VOCAB, codebook=CODEBOOK, img_start=P + 1, text_start=P + 1 + N + 1)


def host_isa() -> str:
    """Which bf16 GEMM code path this host's CPU gives torch / oneDNN: the reference's CPU forward (and therefore every
    float fixture recorded from it) is bit-reproducible only within one class.  Fixtures that contain float
    GEMM outputs are stored once per class: tests/golden/<name>.<host_isa>.npz."""
    try:
        flags = next(ln for ln in open("/proc/cpuinfo") if ln.startswith("flags")).split()
    except (OSError, StopIteration):
        return "other"
    for key in ("amx_bf16", "avx512_bf16"):
        if key in flags:
            return key
    return "avx512" if "avx512f" in flags else "other"


def add_break_line(sequence: List[int], H: int, W: int, new_number: int = 0) -> List[int]:
    """utils/image_utils.py:149-157."""
    result: List[int] = []
    for i in range(H):
        result.extend(sequence[i * W:(i + 1) * W] + [new_number])
    return result


def calculate_vq_params(image_height: int, image_width: int, vae_scale: int = 16):
    """utils/image_utils.py:95-111."""
    gh, gw = image_height // vae_scale, image_width // vae_scale
    return gh * gw, gw, gh, gw


def synthetic_job(height: int = 512, width: int = 512, text_gen_length: int = 256, prompt_len: int = 64,
                  uncond_prompt_len: int = 24, in_height: Optional[int] = None, in_width: Optional[int] = None,
                  seed: int = 1) -> dict:
    """Token sequence of one TI2TI job assembled like inference.py:117-161 (random prompt / input-image ids).

    512x512 defaults give L = 64 + 1058 + 2 + 1056 + 1 + 256 + 1 = 2438 (BASELINE config 2); 256x256 with a
    512x512 input image gives L = 1654 (config 1)."""
    g = torch.Generator().manual_seed(seed)
    prompt_ids = torch.randint(0, 126000, (prompt_len,), generator=g).tolist()
    unc_ids = prompt_ids[:uncond_prompt_len]
    _, _, ih, iw = calculate_vq_params(in_height or 512, in_width or 512)
    img_vq = (torch.randint(0, CODEBOOK, (ih * iw,), generator=g) + TEXT_VOCAB).tolist()
    input_img = [BOI] + add_break_line(img_vq, ih, iw, NEW_LINE) + [EOI]
    con_prefix = prompt_ids[:-1] + input_img + prompt_ids[-1:]
    uncon_text = unc_ids[:-1] + input_img + unc_ids[-1:]
    uncon_image = prompt_ids
    seq_len, newline_every, gh, gw = calculate_vq_params(height, width)
    img_mask = add_break_line([MASK] * seq_len, gh, gw, NEW_LINE)
    pred = [BOA, BOI] + img_mask + [EOI] + [MASK] * text_gen_length + [EOA]
    image_start = len(con_prefix) + 2
    image_end = image_start + len(img_mask)
    text_start = image_end + 1
    return dict(input_ids=torch.tensor(con_prefix + pred, dtype=torch.long).unsqueeze(0),
                uncon_text=torch.tensor(uncon_text, dtype=torch.long).unsqueeze(0),
                uncon_image=torch.tensor(uncon_image, dtype=torch.long).unsqueeze(0),
                text_start=text_start, text_end=text_start + text_gen_length, image_start=image_start,
                seq_len=seq_len, newline_every=newline_every)


def paint_job(kind: str, codebook_size: int = CODEBOOK, text_vocab: int = TEXT_VOCAB, seed: int = 21, **job_kw):
    """A TI2TI job in painting mode (inference.py:141-146 -> utils/image_utils.py encode_img_with_paint): the OUTPUT image
    span starts partly known — the cells outside a centred rectangle keep codes ("inpainting") or only the cells inside do
    ("outpainting") — so the sampler's unknown count starts below N.  Same layout as synthetic_job otherwise."""
    job = synthetic_job(**job_kw)
    ids = job["input_ids"].clone()
    n, per = job["seq_len"], job["newline_every"]
    rows = n // per
    g = torch.Generator().manual_seed(seed)
    codes = torch.randint(0, codebook_size, (n,), generator=g) + text_vocab
    inside = torch.zeros(rows, per, dtype=torch.bool)
    inside[rows // 4:rows - rows // 4, per // 4:per - per // 4] = True
    known = (~inside if kind == "inpainting" else inside).reshape(-1)
    pos = [i for i in range(job["image_start"], job["image_start"] + n + rows) if int(ids[0, i]) != NEW_LINE]
    for j, p_ in enumerate(pos):
        if bool(known[j]):
            ids[0, p_] = codes[j]
    job["input_ids"] = ids
    return job


# ---- MAGVITv2 decoder of MMaDA-Parallel-M --------------------------------------------------------
# Defaults of VQGANDecoder.__init__ (MMaDA-Parallel-M/models/modeling_magvitv2.py:278-287); level 0 = full resolution.
def dllm_cache_script():
    """The call sequence of the fixture (shared with the tests): a prime call, then compute-mask steps on changed ids.
    Returns [(cat, ids, mask-or-None)]; B = 1 (the reference's rotary q_mask indexes one sequence, :714-716)."""
    job = synthetic_job(height=64, width=64, text_gen_length=16, prompt_len=8, uncond_prompt_len=4, in_height=64,
                        in_width=64, seed=1)  # the tiny job of the forward fixtures
    ids0 = job["input_ids"].clone()
    L = ids0.shape[1]
    ts, te = job["text_start"], job["text_end"]
    g = torch.Generator().manual_seed(11)
    ids1 = ids0.clone()
    ids1[0, ts:ts + 6] = torch.randint(0, 1000, (6,), generator=g)          # six text tokens get unmasked
    m1 = torch.zeros(1, L, dtype=torch.bool)
    m1[0, ts:te] = True                                                    # recompute the text span ...
    m1[0, job["image_start"] + 1:job["image_start"] + 4] = True            # ... and three image tokens
    ids2 = ids1.clone()
    ids2[0, job["image_start"] + 1] = TEXT_VOCAB + 77                # one image token committed
    m2 = torch.zeros(1, L, dtype=torch.bool)
    m2[0, job["image_start"]:job["image_start"] + 9] = True
    m2[0, te - 3:te] = True
    return [("cond", ids0, None), ("cond", ids1, m1), ("cond", ids2, m2), ("other", ids1, None), ("other", ids2, m2)]



VQ_CFG_M = dict(ch=128, ch_mult=[1, 1, 2, 2, 4], num_res_blocks=[4, 4, 3, 4, 3], z_channels=13, out_ch=3)
VQ_CFG_TINY = dict(ch=128, ch_mult=[1, 2], num_res_blocks=[1, 2], z_channels=13, out_ch=3)
# Defaults of VQGANEncoder.__init__ (modeling_magvitv2.py:62-73)
VQ_ENC_CFG_M = dict(ch=128, ch_mult=[1, 2, 2, 4, 4], num_res_blocks=[4, 3, 4, 3, 4], z_channels=13, in_ch=3)
VQ_ENC_CFG_TINY = dict(ch=128, ch_mult=[1, 2], num_res_blocks=[2, 1], z_channels=13, in_ch=3)


def vq_decoder_param_shapes(cfg: dict) -> Dict[str, tuple]:
    """State-dict keys and shapes of the reference VQGANDecoder for `cfg` (module tree modeling_magvitv2.py:305-367,
    ResnetBlock common_modules.py:299-335, AttnBlock :168-185, Upsample :27-34)."""
    ch, mult, nrb, zc = cfg["ch"], cfg["ch_mult"], cfg["num_res_blocks"], cfg["z_channels"]
    out: Dict[str, tuple] = {}

    def conv(p, co, ci, k):
        out[p + ".weight"], out[p + ".bias"] = (co, ci, k, k), (co,)

    def norm(p, c):
        out[p + ".weight"], out[p + ".bias"] = (c,), (c,)

    def res(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", co, ci, 3)
        norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".nin_shortcut", co, ci, 1)

    block_in = ch * mult[-1]
    conv("conv_in", block_in, zc, 3)
    res("mid.block_1", block_in, block_in)
    norm("mid.attn_1.norm", block_in)
    for n in ("q", "k", "v", "proj_out"):
        conv("mid.attn_1." + n, block_in, block_in, 1)
    res("mid.block_2", block_in, block_in)
    for lvl in reversed(range(len(mult))):
        block_out = ch * mult[lvl]
        for b in range(nrb[lvl]):
            res(f"up.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            conv(f"up.{lvl}.upsample.conv", block_in, block_in, 3)
    norm("norm_out", block_in)
    conv("conv_out", cfg["out_ch"], block_in, 3)
    conv("post_quant_conv", zc, zc, 1)
    return out


def vq_encoder_param_shapes(cfg: dict) -> Dict[str, tuple]:
    """State-dict keys and shapes of the reference VQGANEncoder (module tree modeling_magvitv2.py:81-141)."""
    ch, mult, nrb, zc = cfg["ch"], cfg["ch_mult"], cfg["num_res_blocks"], cfg["z_channels"]
    out: Dict[str, tuple] = {}

    def conv(p, co, ci, k):
        out[p + ".weight"], out[p + ".bias"] = (co, ci, k, k), (co,)

    def norm(p, c):
        out[p + ".weight"], out[p + ".bias"] = (c,), (c,)

    def res(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", co, ci, 3)
        norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".nin_shortcut", co, ci, 1)

    conv("conv_in", ch, cfg["in_ch"], 3)
    block_in = ch
    for lvl in range(len(mult)):
        block_out = ch * mult[lvl]
        for b in range(nrb[lvl]):
            res(f"down.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
        if lvl != len(mult) - 1:
            conv(f"down.{lvl}.downsample.conv", block_in, block_in, 3)
    res("mid.block_1", block_in, block_in)
    norm("mid.attn_1.norm", block_in)
    for n in ("q", "k", "v", "proj_out"):
        conv("mid.attn_1." + n, block_in, block_in, 1)
    res("mid.block_2", block_in, block_in)
    norm("norm_out", block_in)
    conv("conv_out", zc, block_in, 3)
    conv("quant_conv", zc, zc, 1)
    return out


def synthetic_image(B: int, H: int, W: int, seed: int = 0) -> torch.Tensor:
    """Seeded smooth-ish image batch in [-1, 1], [B, 3, H, W] fp32 (what image_transform_squash + Normalize(0.5, 0.5)
    hands to get_code, MMaDA-Parallel-M/training/utils.py:208-213)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    low = torch.randn(B, 3, max(2, H // 8), max(2, W // 8), generator=g)
    img = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=False)
    img = img + 0.1 * torch.randn(B, 3, H, W, generator=g)
    return torch.tanh(img)


def synthetic_vq_state_dict(cfg: dict, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 decoder (or, for a cfg with "in_ch", encoder) weights (CPU generator, so every machine gets the same tensors): conv N(0, 1/fan_in),
    conv bias N(0, 0.05²), GroupNorm weight 1+N(0, 0.1²), bias N(0, 0.1²)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    shapes = vq_encoder_param_shapes(cfg) if "in_ch" in cfg else vq_decoder_param_shapes(cfg)
    for name, shape in shapes.items():
        if len(shape) == 4:
            std = (shape[1] * shape[2] * shape[3]) ** -0.5
            sd[name] = torch.randn(shape, generator=g) * std
        elif ".norm" in name or name.startswith("norm_out"):
            sd[name] = torch.randn(shape, generator=g) * 0.1 + (1.0 if name.endswith(".weight") else 0.0)
        else:
            sd[name] = torch.randn(shape, generator=g) * 0.05
    return sd


# ---- diffusers.VQModel (A variant image tokenizer): configs + seeded synthetic checkpoints -----------------------------------
# the f16 / 8192-code layout the reference's token arithmetic assumes (vae_scale 16, VQ offset 126356 + 8192 codes)
VQMODEL_CFG_A = dict(in_channels=3, out_channels=3, block_out_channels=[128, 256, 256, 512, 768], layers_per_block=2,
                     latent_channels=64, num_vq_embeddings=8192, norm_num_groups=32, vq_embed_dim=None,
                     mid_block_add_attention=False, lookup_from_codebook=True)
VQMODEL_CFG_TINY = dict(in_channels=3, out_channels=3, block_out_channels=[128, 256], layers_per_block=1, latent_channels=8,
                        num_vq_embeddings=64, norm_num_groups=32, vq_embed_dim=None, mid_block_add_attention=True,
                        lookup_from_codebook=True)


def vqmodel_param_shapes(cfg: dict) -> Dict[str, tuple]:
    """Checkpoint keys / shapes of a diffusers VQModel with this config (Encoder, Decoder, quant convs, codebook)."""
    bo, lpb, lat = cfg["block_out_channels"], cfg["layers_per_block"], cfg["latent_channels"]
    emb = cfg.get("vq_embed_dim") or lat
    L, shapes = len(bo), {}

    def conv(p, co, ci, k):
        shapes[p + ".weight"], shapes[p + ".bias"] = (co, ci, k, k), (co,)

    def norm(p, c):
        shapes[p + ".weight"], shapes[p + ".bias"] = (c,), (c,)

    def res(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", co, ci, 3); norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    def mid(p, c):
        res(p + ".resnets.0", c, c)
        if cfg.get("mid_block_add_attention", True):
            norm(p + ".attentions.0.group_norm", c)
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                shapes[f"{p}.attentions.0.{n}.weight"], shapes[f"{p}.attentions.0.{n}.bias"] = (c, c), (c,)
        res(p + ".resnets.1", c, c)

    conv("encoder.conv_in", bo[0], cfg["in_channels"], 3)
    cin = bo[0]
    for i in range(L):
        for j in range(lpb):
            res(f"encoder.down_blocks.{i}.resnets.{j}", cin, bo[i]); cin = bo[i]
        if i != L - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cin, cin, 3)
    mid("encoder.mid_block", cin)
    norm("encoder.conv_norm_out", cin); conv("encoder.conv_out", lat, cin, 3)
    conv("quant_conv", emb, lat, 1)
    shapes["quantize.embedding.weight"] = (cfg["num_vq_embeddings"], emb)
    conv("post_quant_conv", lat, emb, 1)
    cin = bo[-1]
    conv("decoder.conv_in", cin, lat, 3)
    mid("decoder.mid_block", cin)
    for i in range(L):
        co = bo[L - 1 - i]
        for j in range(lpb + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", cin, co); cin = co
        if i != L - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", cin, cin, 3)
    norm("decoder.conv_norm_out", cin); conv("decoder.conv_out", cfg["out_channels"], cin, 3)
    return shapes


def synthetic_vqmodel_state_dict(cfg: dict, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights with fan-in scaled convolutions (activations stay O(1) through the depth) and a spread-out
    codebook, so that nearest-code decisions are well separated."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in vqmodel_param_shapes(cfg).items():
        if k == "quantize.embedding.weight":
            sd[k] = torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            sd[k] = torch.randn(shp, generator=g) * 0.02
        elif len(shp) == 1:
            sd[k] = 1.0 + torch.randn(shp, generator=g) * 0.05
        else:
            fan_in = 1
            for v in shp[1:]:
                fan_in *= v
            sd[k] = torch.randn(shp, generator=g) * (1.0 / fan_in) ** 0.5
    return sd
