"""Planted 'copy' checkpoint: block 0's attention copies the token DELTA positions back (content-independent rotary q/k), the LM
head reads the copied token out; everything else random and small.  bf16-vs-fp32 decision agreement on CPU."""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmada_parallel_amd import synth
from oracle import llada_oracle as O

torch.set_num_threads(8)
cfg = dict(d_model=1024, n_heads=8, n_kv_heads=8, n_layers=4, mlp_hidden_size=2048, vocab_size=134656, embedding_size=134656,
           rms_norm_eps=1e-5, rope_theta=500000.0, max_sequence_length=4096)

def make(delta, nfreq=16, amp=6.0, c0=1.0, small=0.3, beta=12.0, seed=0, head_noise=1.0):
    d, H = cfg["d_model"], cfg["n_heads"]; hd = d // H
    sd = synth.synthetic_state_dict(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1234)
    e0 = torch.randn(d, generator=g); e0 /= e0.norm()
    wte = sd["model.transformer.wte.weight"].float()
    wte = wte - (wte @ e0)[:, None] * e0[None, :] + c0 * e0[None, :]          # shared component c0 along e0, content orthogonal to it
    sd["model.transformer.wte.weight"] = wte.to(torch.bfloat16)
    p = "model.transformer.blocks."
    for i in range(cfg["n_layers"]):
        for n in ("v_proj", "attn_out", "ff_proj", "up_proj", "ff_out"):
            if i == 0 and n in ("v_proj", "attn_out"):
                continue
            sd[f"{p}{i}.{n}.weight"] = (sd[f"{p}{i}.{n}.weight"].float() * small).to(torch.bfloat16)
    # block 0: q, k read only the e0 component -> constant vectors; rotary phases put the score peak at i - j = delta
    theta = cfg["rope_theta"]
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float64) / hd))   # [hd/2]
    qv = torch.zeros(hd, dtype=torch.float64); kv = torch.zeros(hd, dtype=torch.float64)
    for f in range(nfreq):
        # rotate_half pairs dim f with f + hd/2: (x1, x2) -> (x1 cos - x2 sin, x2 cos + x1 sin) at angle pos * inv[f]
        kv[f] = amp; kv[f + hd // 2] = 0.0
        ang = delta * inv[f].item()
        qv[f] = amp * math.cos(ang); qv[f + hd // 2] = -amp * math.sin(ang)
    wq = torch.zeros(d, d); wk = torch.zeros(d, d)
    for h in range(H):
        wq[h * hd:(h + 1) * hd] = qv.float()[:, None] * e0[None, :]
        wk[h * hd:(h + 1) * hd] = kv.float()[:, None] * e0[None, :]
    sd[p + "0.q_proj.weight"] = (wq + sd[p + "0.q_proj.weight"].float() * small).to(torch.bfloat16)
    sd[p + "0.k_proj.weight"] = (wk + sd[p + "0.k_proj.weight"].float() * small).to(torch.bfloat16)
    wv = sd[p + "0.v_proj.weight"].float(); wv = wv - (wv @ e0)[:, None] * e0[None, :]   # values carry content only
    sd[p + "0.v_proj.weight"] = (wv * 2.0).to(torch.bfloat16)
    sd[p + "0.attn_out.weight"] = (sd[p + "0.attn_out.weight"].float() * 2.0).to(torch.bfloat16)
    # LM head: row v = the direction token v leaves in the stream when it is copied, times a log-normal scale
    an = sd[p + "0.attn_norm.weight"].float()
    x = wte
    lnx = x / x.pow(2).mean(-1, keepdim=True).add(1e-5).sqrt() * an[None, :]
    M = sd[p + "0.attn_out.weight"].float() @ sd[p + "0.v_proj.weight"].float()       # d x d
    y = lnx @ M.t()                                                                     # V x d
    y = y / y.norm(dim=-1, keepdim=True)
    scale = torch.exp(torch.randn(y.shape[0], generator=g) * 0.35)
    noise = sd["model.transformer.ff_out.weight"].float() * head_noise
    sd["model.transformer.ff_out.weight"] = (y * (beta * scale)[:, None] / math.sqrt(d) * 4 + noise).to(torch.bfloat16)
    return sd

def decisions(sd, job, dtype):
    sdd = {k: v.to(dtype) for k, v in sd.items()}
    ids = job["input_ids"]
    N = job["seq_len"]; nl = job["newline_every"]
    pos = [i for i in range(job["image_start"], job["image_start"] + N + N // nl) if int(ids[0, i]) != synth.NEW_LINE]
    unc = ids.clone(); unc[0, :job["uncon_image"].shape[1]] = job["uncon_image"][0]
    xc = O.forward_hidden(sdd, cfg, ids); xu = O.forward_hidden(sdd, cfg, unc)
    c = O.head(sdd, cfg, xc[:, pos], synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK)[0]
    u = O.head(sdd, cfg, xu[:, pos], synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK)[0]
    t = O.head(sdd, cfg, xc[:, job["text_start"]:job["text_end"]])[0]
    return c, u, t, pos

def report(name, sd, job, delta):
    t0 = time.time()
    cb, ub, tb, pos = decisions(sd, job, torch.bfloat16)
    cf, uf, tf, _ = decisions(sd, job, torch.float32)
    fb = cb.float() + 4.0 * (cb.float() - ub.float()); ff = cf + 4.0 * (cf - uf)
    ia = (fb.argmax(-1) == ff.argmax(-1)).float().mean().item(); ta = (tb.float().argmax(-1) == tf.argmax(-1)).float().mean().item()
    ids = job["input_ids"][0]
    want_img = torch.tensor([int(ids[p - delta]) - synth.TEXT_VOCAB for p in pos])
    copy_ok = (ff.argmax(-1) == want_img).float().mean().item()
    want_txt = torch.tensor([int(ids[p - delta]) for p in range(job["text_start"], job["text_end"])])
    copy_txt = (tf.argmax(-1) == want_txt).float().mean().item()
    pb = torch.softmax(tb.double(), -1).max(-1).values; pf = torch.softmax(tf.double(), -1).max(-1).values
    k = max(1, len(pb) // 8)
    ord_agree = len(set(pb.topk(k).indices.tolist()) & set(pf.topk(k).indices.tolist())) / k
    pbi = torch.softmax(fb, -1).max(-1).values; pfi = torch.softmax(ff, -1).max(-1).values
    top2 = ff.topk(2, -1).values; nz = (fb - ff).abs().max(-1).values
    print(f"{name}: image argmax bf16==fp32 {ia:.3f}, copies the planted token {copy_ok:.3f}, {len(set(ff.argmax(-1).tolist()))} distinct, "
          f"min margin/max-noise {((top2[:,0]-top2[:,1])/nz).min():.1f} | text {ta:.3f}, copies {copy_txt:.3f}, {len(set(tf.argmax(-1).tolist()))} distinct | "
          f"text conf range {pf.min():.3f}..{pf.max():.3f}, first-commit set agree {ord_agree:.2f} | image conf {pfi.min():.3f}..{pfi.max():.3f} | {time.time()-t0:.0f}s", flush=True)

job = synth.synthetic_job(128, 128, text_gen_length=64, prompt_len=32, uncond_prompt_len=12, in_height=256, in_width=256, seed=1)
L = job["input_ids"].shape[1]
delta = job["image_start"] - 40   # output-image and text positions copy from inside the input image span
print("L =", L, "image_start", job["image_start"], "text_start", job["text_start"], "delta", delta)
for kw in [dict(), dict(amp=4.0), dict(amp=8.0, nfreq=24), dict(beta=6.0), dict(beta=24.0), dict(small=0.6), dict(small=1.0)]:
    report(str(kw), make(delta, **kw), job, delta)
