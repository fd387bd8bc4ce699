"""Exploration (CPU): which synthetic weights make the sampler's decisions robust to bf16 rounding?
Compares the oracle's bf16 evaluation with an fp32 evaluation of the SAME bf16-representable weights."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmada_parallel_amd import synth
from oracle import llada_oracle as O

torch.set_num_threads(8)
cfg = dict(d_model=1024, n_heads=8, n_kv_heads=8, n_layers=4, mlp_hidden_size=2048, vocab_size=134656, embedding_size=134656,
           rms_norm_eps=1e-5, rope_theta=500000.0, max_sequence_length=4096)

def make(qk, vo, lns, seed=0, mlp=1.0, lowrank=0.0):
    sd = synth.synthetic_state_dict(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 77)
    for i in range(cfg["n_layers"]):
        b = f"model.transformer.blocks.{i}."
        for n in ("q_proj", "k_proj"):
            sd[b + n + ".weight"] = (sd[b + n + ".weight"].float() * qk).to(torch.bfloat16)
        for n in ("v_proj", "attn_out"):
            sd[b + n + ".weight"] = (sd[b + n + ".weight"].float() * vo).to(torch.bfloat16)
        for n in ("ff_proj", "up_proj", "ff_out"):
            sd[b + n + ".weight"] = (sd[b + n + ".weight"].float() * mlp).to(torch.bfloat16)
    w = sd["model.transformer.ff_out.weight"].float()
    scale = torch.exp(torch.randn(w.shape[0], generator=g) * lns)
    sd["model.transformer.ff_out.weight"] = (w * scale[:, None]).to(torch.bfloat16)
    return sd

def decisions(sd, job, dtype):
    sdd = {k: v.to(dtype) for k, v in sd.items()}
    ids = job["input_ids"]
    L = ids.shape[1]
    N = job["seq_len"]; nl = job["newline_every"]
    pos = [i for i in range(job["image_start"], job["image_start"] + N + N // nl) if int(ids[0, i]) != synth.NEW_LINE]
    unc = ids.clone(); unc[0, :job["uncon_image"].shape[1]] = job["uncon_image"][0]
    xc = O.forward_hidden(sdd, cfg, ids)
    xu = O.forward_hidden(sdd, cfg, unc)
    c = O.head(sdd, cfg, xc[:, pos], synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK)[0]
    u = O.head(sdd, cfg, xu[:, pos], synth.TEXT_VOCAB, synth.TEXT_VOCAB + synth.CODEBOOK)[0]
    t = O.head(sdd, cfg, xc[:, job["text_start"]:job["text_end"]])[0]
    return c, u, t

def report(name, sd, job):
    t0 = time.time()
    cb, ub, tb = decisions(sd, job, torch.bfloat16)
    cf, uf, tf = decisions(sd, job, torch.float32)
    # reference arithmetic of the combine: bf16 ops (parallel_generator.py:282-295) -- here just the decision level
    fb = (cb.float() + 4.0 * (cb.float() - ub.float()))
    ff = (cf + 4.0 * (cf - uf))
    ia = (fb.argmax(-1) == ff.argmax(-1)).float().mean().item()
    ta = (tb.float().argmax(-1) == tf.argmax(-1)).float().mean().item()
    # margins in units of the bf16-vs-fp32 noise
    noise = (fb - ff).std().item()
    top2 = ff.topk(2, -1).values
    marg = ((top2[:, 0] - top2[:, 1]) / max(noise, 1e-9))
    tnoise = (tb.float() - tf).std().item()
    ttop2 = tf.topk(2, -1).values
    tmarg = (ttop2[:, 0] - ttop2[:, 1]) / max(tnoise, 1e-9)
    # text confidence ordering: the positions that would be committed first (top-8 by max prob)
    pb = torch.softmax(tb.double(), -1).max(-1).values; pf = torch.softmax(tf.double(), -1).max(-1).values
    k = 8
    ord_agree = len(set(pb.topk(k).indices.tolist()) & set(pf.topk(k).indices.tolist())) / k
    uniq_img = len(set(ff.argmax(-1).tolist())); uniq_txt = len(set(tf.argmax(-1).tolist()))
    print(f"{name}: image argmax agree {ia:.3f} (margin/noise p5 {marg.quantile(0.05):.1f} median {marg.median():.1f}; {uniq_img} distinct codes) | "
          f"text argmax agree {ta:.3f} (margin/noise p5 {tmarg.quantile(0.05):.1f} median {tmarg.median():.1f}; {uniq_txt} distinct) | "
          f"top-{k} confident positions agree {ord_agree:.2f} | {time.time() - t0:.0f}s", flush=True)

job = synth.synthetic_job(128, 128, text_gen_length=64, prompt_len=32, uncond_prompt_len=12, in_height=256, in_width=256, seed=1)
print("L =", job["input_ids"].shape[1])
for (qk, vo, lns, mlp) in [(1, 1, 0, 1), (3, 1, 0, 1), (3, 1, 1.0, 1), (3, 1, 1.5, 1), (4, 2, 1.5, 1), (3, 1, 2.0, 1), (6, 1, 1.5, 1)]:
    report(f"qk x{qk} vo x{vo} lognormal {lns} mlp x{mlp}", make(qk, vo, lns, mlp=mlp), job)
