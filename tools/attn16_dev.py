#!/usr/bin/env python
"""Development harness of csrc/attention.hip: correctness against fp32 SDPA on edge shapes, then interleaved timing of the
attention forms at the 8B shapes (kernel times come from rocprofv3 --kernel-trace --stats around this script; the hipEvent
figure printed here is the whole mmada_sdpa call incl. its three layout kernels).

    python tools/attn16_dev.py --check            # shapes incl. partial groups / tiles, GQA, forced rescale
    python tools/attn16_dev.py --time --batch 1   # forms "1,0" = late waves (default), plain order
(Round 6 used it with extra kernel variants — half-step pipeline, shadowed exponentials — that were measured and removed:
profiles/HISTORY.md.)
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmada_parallel_amd import abi, synth  # noqa: E402


def make_handle(lib, max_seq=4096):
    cfg = synth.CFG_8B
    c = abi.MmadaCfg(d_model=cfg["d_model"], n_layers=1, n_heads=32, n_kv_heads=32, head_dim=128, mlp_hidden=12288,
                     vocab=134656, max_seq=max_seq, rms_eps=1e-5, rope_theta=500000.0, tp_rank=0, tp_size=1,
                     mask_token_id=126336, text_vocab_size=126356, codebook_size=8192, reserved=0)
    inv = (C.c_float * 64)(*(1.0 / (500000.0 ** (torch.arange(0, 128, 2, dtype=torch.float) / 128))).tolist())
    h = C.c_void_p()
    abi.check(lib.mmada_create(C.byref(c), inv, C.byref(h)), "create")
    return h


def sdpa(lib, h, q, k, v, st):
    B, H, L, _ = q.shape
    out = torch.empty(B, L, H * 128, dtype=torch.bfloat16, device="cuda")
    abi.check(lib.mmada_sdpa(h, q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, k.shape[1], L, st), "sdpa")
    return out.view(B, L, H, 128).permute(0, 2, 1, 3)


def check(lib, h, st):
    g = torch.Generator(device="cuda").manual_seed(0)
    worst = 0.0
    for (B, H, Hkv, L, spike) in [(1, 4, 4, 100, 0), (2, 8, 8, 333, 0), (1, 8, 2, 64, 0), (1, 8, 8, 65, 0), (1, 2, 2, 17, 0), (3, 8, 4, 1000, 1),
                                  (1, 32, 32, 2438, 0), (2, 32, 32, 2438, 1), (1, 4, 4, 1, 0), (1, 16, 16, 1654, 0)]:
        nb = lib.mmada_workspace_bytes(h, B, max(L, 64))
        ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
        abi.check(lib.mmada_set_workspace(h, (ws.data_ptr() + 255) // 256 * 256, nb), "ws")
        q, k, v = (torch.randn(B, hh, L, 128, device="cuda", generator=g).to(torch.bfloat16) for hh in (H, Hkv, Hkv))
        if spike:  # force the rescale path late in the key range: one key with a huge score for some queries
            k[:, :, L * 3 // 4] *= 6.0
            q[:, :, ::7] *= 3.0
        rep = H // Hkv
        ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float().repeat_interleave(rep, 1), v.float().repeat_interleave(rep, 1))
        res = {}
        for form in (0, 1):
            abi.check(lib.mmada_set_option(b"attention_form", form), "opt")
            res[form] = sdpa(lib, h, q, k, v, st).float()
        torch.cuda.synchronize()
        e_new = (res[1] - ref).abs().max().item()
        same = torch.equal(res[0], res[1])
        nan = bool(torch.isnan(res[1]).any())
        worst = max(worst, e_new)
        bits = res[1].to(torch.bfloat16).view(torch.int16).to(torch.int64)
        digest = int((bits * torch.arange(1, bits.numel() + 1, device=bits.device).view(bits.shape) % 1000003).sum().item())
        print(f"B={B} H={H}/{Hkv} L={L} spike={spike}: max|err| vs fp32 {e_new:.3e}  form 0 == form 1 bits {same}  nan {nan}  digest {digest}", flush=True)
    print("worst new", worst)


def timing(lib, h, st, args):
    B, H, L = args.batch, 32, args.L
    nb = lib.mmada_workspace_bytes(h, B, L)
    ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    abi.check(lib.mmada_set_workspace(h, (ws.data_ptr() + 255) // 256 * 256, nb), "ws")
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(B, H, L, 128, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3))
    out = torch.empty(B, L, H * 128, dtype=torch.bfloat16, device="cuda")
    flops = 4.0 * B * H * L * L * 128

    def run(n):
        for _ in range(n):
            abi.check(lib.mmada_sdpa(h, q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, H, L, st), "sdpa")

    forms = []
    for f in args.forms.split(","):
        a, _, b = f.partition(":")
        forms.append((int(a), int(b or 0)))
    abi.check(lib.mmada_set_option(b"attention_form", forms[0][0]), "opt")
    run(args.warm)
    torch.cuda.synchronize()
    ms = {f: [] for f in forms}
    for _ in range(args.rounds):
        for f in forms:
            abi.check(lib.mmada_set_option(b"attention_form", f[0]), "opt")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(args.iters)
            e1.record()
            torch.cuda.synchronize()
            ms[f].append(e0.elapsed_time(e1) / args.iters)
    for f in forms:
        t = sorted(ms[f])[len(ms[f]) // 2]
        print(f"B={B} L={L} form {f[0]} variant {f[1]}: median {t * 1e3:.1f} us per mmada_sdpa call (incl. 3 layout kernels) "
              f"= {flops / t / 1e9:.0f} TF lower bound", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--L", type=int, default=2438)
    ap.add_argument("--iters", type=int, default=25)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--warm", type=int, default=2000)
    ap.add_argument("--forms", default="1,0")
    args = ap.parse_args()
    lib = abi.lib()
    h = make_handle(lib)
    st = torch.cuda.current_stream().cuda_stream
    if args.check:
        check(lib, h, st)
    if args.time:
        timing(lib, h, st, args)
    lib.mmada_set_option(b"attention_form", -1)


if __name__ == "__main__":
    main()
