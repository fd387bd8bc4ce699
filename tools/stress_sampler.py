#!/usr/bin/env python
"""Randomised bit-exactness sweep of the sampler kernels against the C oracle (oracle/sampler_oracle.c): many seeds,
shapes, logit scales (incl. large magnitudes and coarse grids that create exact ties), CFG scales and k / mask_len
values.  A validation tool (uses the oracle, so it is test infrastructure); any mismatch prints the failing case and
exits non-zero.    python tools/stress_sampler.py [iterations]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmada_parallel_amd import abi, synth  # noqa: E402
from oracle import sampler_oracle as so  # noqa: E402

DEV = "cuda:0"


def bits(t):
    return t.detach().to("cpu", torch.bfloat16).contiguous().view(torch.int16)


def make_handle():
    lib = abi.lib()
    c = abi.MmadaCfg(d_model=256, n_layers=1, n_heads=2, n_kv_heads=2, head_dim=128, mlp_hidden=512, vocab=134656,
                     max_seq=1024, rms_eps=1e-5, rope_theta=500000.0, tp_rank=0, tp_size=1, mask_token_id=synth.MASK,
                     text_vocab_size=synth.TEXT_VOCAB, codebook_size=synth.CODEBOOK, reserved=0)
    h = C.c_void_p()
    abi.check(lib.mmada_create(C.byref(c), None, C.byref(h)), "create")
    return lib, h


def rand_logits(g, shape, it):
    scale = [0.5, 2.0, 8.0, 30.0][it % 4]
    x = torch.randn(shape, generator=g) * scale
    q = [None, 1, 4, None, 16][it % 5]
    if q:
        x = (x * q).round() / q
    return x.to(torch.bfloat16)


def main(iters=60):
    lib, h = make_handle()
    st = abi.stream_ptr()
    bad = 0
    for it in range(iters):
        g = torch.Generator().manual_seed(1000 + it)
        # ---- text select (plain and CFG-combined) ----
        B = 1 + it % 3
        T = int(torch.randint(1, 300, (1,), generator=g))
        V = [1000, 2560, 8200, 134656][it % 4] if T < 80 else [1000, 2560][it % 2]
        L, ts = T + 13, 5
        ld = (V + 7) // 8 * 8
        lg = torch.zeros(B, T, ld, dtype=torch.bfloat16)
        lg[..., :V] = rand_logits(g, (B, T, V), it)
        un = torch.zeros(B, T, ld, dtype=torch.bfloat16)
        un[..., :V] = (lg[..., :V].float() + torch.randn(B, T, V, generator=g)).to(torch.bfloat16)
        ids = torch.randint(0, 1000, (B, L), generator=g)
        ids[:, ts:ts + T] = synth.MASK
        for b in range(B):
            ids[b, ts + torch.randperm(T, generator=g)[: T // 4]] = 3 + b
        k = [int(torch.randint(0, max(1, T - T // 4) + 1, (1,), generator=g)) for _ in range(B)]
        k_dev = torch.tensor(k, dtype=torch.int32, device=DEV)
        scratch = torch.empty(B * T * 16, dtype=torch.uint8, device=DEV)
        lgd, und = lg.to(DEV), un.to(DEV)
        ids_dev = ids.to(DEV)
        abi.check(lib.mmada_text_select(h, lgd.data_ptr(), None, B, T, V, ld, ids_dev.data_ptr(), L, ts, k_dev.data_ptr(),
                                        scratch.data_ptr(), st), "text_select")
        ref, _, _ = so.text_select(lg[..., :V].contiguous(), None, ids, ts, k)
        if not torch.equal(ids_dev.cpu(), ref):
            bad += 1
            print(f"MISMATCH text_select it={it} B={B} T={T} V={V} k={k}")
        cfg = [0.7, 1.5, 2.5, 0.0][it % 4]
        ids_dev = ids.to(DEV)
        abi.check(lib.mmada_text_select_cfg(h, lgd.data_ptr(), und.data_ptr(), cfg, None, B, T, V, ld, ids_dev.data_ptr(), L,
                                            ts, k_dev.data_ptr(), scratch.data_ptr(), st), "text_select_cfg")
        ref, _, _ = so.text_select_cfg(lg[..., :V].contiguous(), un[..., :V].contiguous(), cfg, ids, ts, k)
        if not torch.equal(ids_dev.cpu(), ref):
            bad += 1
            print(f"MISMATCH text_select_cfg it={it} B={B} T={T} V={V} cfg={cfg} k={k}")
        # ---- image probs (A dual CFG, M) + commit ----
        N = int(torch.randint(1, 300, (1,), generator=g))
        CB = [512, 8192, 1024][it % 3]
        c = rand_logits(g, (B, N, CB), it + 1)
        ut = (c.float() + torch.randn(B, N, CB, generator=g)).to(torch.bfloat16)
        ui = (c.float() + torch.randn(B, N, CB, generator=g) * 0.3).to(torch.bfloat16)
        cs, ci = [(0.0, 4.0), (2.5, 4.0), (3.0, 0.0), (0.0, 0.0), (1.3, 7.0)][it % 5]
        cd, utd, uid = c.to(DEV), ut.to(DEV), ui.to(DEV)
        probs = torch.empty(B, N, CB, dtype=torch.bfloat16, device=DEV)
        am = torch.empty(B, N, dtype=torch.int32, device=DEV)
        pm = torch.empty(B, N, dtype=torch.bfloat16, device=DEV)
        abi.check(lib.mmada_image_probs(h, cd.data_ptr(), utd.data_ptr(), uid.data_ptr(), B, N, CB, cs, ci, probs.data_ptr(),
                                        am.data_ptr(), pm.data_ptr(), st), "image_probs")
        am_r, pm_r, pr_r = so.image_probs(c, ut, ui, cs, ci, want_probs=True)
        if not (torch.equal(am.cpu(), am_r) and torch.equal(bits(pm), bits(pm_r)) and torch.equal(bits(probs), bits(pr_r))):
            bad += 1
            print(f"MISMATCH image_probs it={it} B={B} N={N} CB={CB} cs={cs} ci={ci}")
        gm = [3.5, 0.3, 2.0][it % 3]
        abi.check(lib.mmada_image_probs_m(h, cd.data_ptr(), utd.data_ptr(), B, N, CB, gm, probs.data_ptr(), am.data_ptr(),
                                          pm.data_ptr(), st), "image_probs_m")
        am_r, pm_r, pr_r = so.image_probs_m(c, ut, gm)
        if not (torch.equal(am.cpu(), am_r) and torch.equal(bits(pm), bits(pm_r)) and torch.equal(bits(probs), bits(pr_r))):
            bad += 1
            print(f"MISMATCH image_probs_m it={it} B={B} N={N} CB={CB} g={gm}")
        Lc = N + 30
        pos = (torch.sort(torch.randperm(Lc - 4, generator=g)[:N]).values + 2).to(torch.int32)
        idc = torch.randint(0, 1000, (B, Lc), generator=g)
        for b in range(B):
            idc[b, pos.long()] = synth.MASK
            kn = pos.long()[torch.randperm(N, generator=g)[: int(torch.randint(0, N + 1, (1,), generator=g))]]
            idc[b, kn] = synth.TEXT_VOCAB + torch.randint(0, synth.CODEBOOK, (kn.numel(),), generator=g)
        sampled = torch.randint(0, synth.CODEBOOK, (B, N), generator=g, dtype=torch.int32)
        p = pm.cpu() if it % 2 else (torch.randint(0, 60, (B, N), generator=g).float() / 4096).to(torch.bfloat16)
        noise = torch.randn(B, N, generator=g).to(torch.bfloat16)
        temp = [0.0, 0.4, 1.0][it % 3]
        mlen = int(torch.randint(-1, N + 3, (1,), generator=g))
        pos_d, s_d, p_d, n_d = pos.to(DEV), sampled.to(DEV), p.to(DEV), noise.to(DEV)
        ml = torch.tensor([mlen], dtype=torch.int32, device=DEV)
        idd = idc.to(DEV)
        abi.check(lib.mmada_image_commit(h, idd.data_ptr(), B, Lc, pos_d.data_ptr(), N, s_d.data_ptr(), p_d.data_ptr(),
                                         n_d.data_ptr(), temp, ml.data_ptr(), synth.TEXT_VOCAB, synth.CODEBOOK, st), "commit")
        if not torch.equal(idd.cpu(), so.image_commit(idc, pos, sampled, p, noise, temp, mlen)):
            bad += 1
            print(f"MISMATCH image_commit it={it} B={B} N={N} temp={temp} mlen={mlen}")
        idd = idc.to(DEV)
        abi.check(lib.mmada_image_commit_m(h, idd.data_ptr(), B, Lc, pos_d.data_ptr(), N, s_d.data_ptr(), p_d.data_ptr(),
                                           n_d.data_ptr(), temp, ml.data_ptr(), synth.TEXT_VOCAB, st), "commit_m")
        if not torch.equal(idd.cpu(), so.image_commit_m(idc, pos, sampled, p, noise, temp, mlen, synth.MASK, synth.TEXT_VOCAB)):
            bad += 1
            print(f"MISMATCH image_commit_m it={it} B={B} N={N} temp={temp} mlen={mlen}")
    print(f"stress_sampler: {iters} iterations x 6 kernels, {bad} mismatches")
    lib.mmada_destroy(h)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 60))
