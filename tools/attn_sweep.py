#!/usr/bin/env python
"""Attention kernel (csrc/attention.hip) at the 8B shapes, measured the way a loop-order comparison has to be measured.

    cd /tmp && rocprofv3 --kernel-trace --stats -d OUT -o a -- python $REPO/tools/attn_sweep.py --batch 1
The C-ABI entry (mmada_sdpa) runs three layout kernels before the attention kernel, so the kernel time comes from
rocprofv3; the hipEvent figure printed here is the whole call.  Random N(0,1) q/k/v (never bench on zeros).

Protocol lesson of round 2 (profiles/r02_attn_variants_warm_b*.txt): twelve loop orders of the kernel (s_setprio around
the MFMA clusters, P exponentiated in 2 / 4 chunks with the P·V MFMAs issued per chunk, sched_group_barrier pinning, a
software-pipelined loop with the next tile's S MFMAs beside the previous tile's exponentials) looked 6-15 % apart when
each was timed for 40 launches one after the other in a fresh process — and within +-2 % of each other (127.9-132.5 us at
B = 1, 208.5-213.5 us at B = 2, production 128.5 / 209.8) once the GPU had been loaded for a second first and the variants
were interleaved in rounds: the clock governor, not the loop order, had made the difference.  None was kept.
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmada_parallel_amd import abi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--L", type=int, default=2438)
    ap.add_argument("--iters", type=int, default=25)
    ap.add_argument("--rounds", type=int, default=14)
    ap.add_argument("--warm", type=int, default=3000)
    ap.add_argument("--data", default="randn", help="randn (the measurement), or zeros: the same instruction stream with no operand toggling — "
                    "a launch that gets faster on zeros was held back by the power governor, not by its schedule (DIAGNOSTIC)")
    ap.add_argument("--forms", default="0,1", help="attention forms to interleave; forms > 1 exist in the -DMMADA_TUNE build only "
                    "(11: no soft-max, 12: no MFMA, 13: no tile barrier — DIAGNOSTIC, wrong results; 14: static priority)")
    args = ap.parse_args()
    forms = [int(f) for f in args.forms.split(",")]
    if max(forms) > 2:   # diagnostic variants live in the -DMMADA_TUNE build (prebuilt copies travel with the snapshot)
        from tools.build_tune import PRODUCT_TUNE_LIB, build_product_tune
        os.environ["MMADA_MI355X_LIB"] = PRODUCT_TUNE_LIB if os.environ.get("MMADA_TUNE_PREBUILT") == "1" and os.path.exists(PRODUCT_TUNE_LIB) else build_product_tune()
    lib = abi.lib()
    cfg = synth.CFG_8B
    c = abi.MmadaCfg(d_model=cfg["d_model"], n_layers=1, n_heads=32, n_kv_heads=32, head_dim=128, mlp_hidden=12288,
                     vocab=134656, max_seq=4096, rms_eps=1e-5, rope_theta=500000.0, tp_rank=0, tp_size=1,
                     mask_token_id=126336, text_vocab_size=126356, codebook_size=8192, reserved=0)
    inv = (C.c_float * 64)(*(1.0 / (500000.0 ** (torch.arange(0, 128, 2, dtype=torch.float) / 128))).tolist())
    h = C.c_void_p()
    abi.check(lib.mmada_create(C.byref(c), inv, C.byref(h)), "create")
    B, H, L = args.batch, 32, args.L
    nb = lib.mmada_workspace_bytes(h, B, L)
    ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    abi.check(lib.mmada_set_workspace(h, (ws.data_ptr() + 255) // 256 * 256, nb), "ws")
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(B, H, L, 128, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3))
    if args.data == "zeros":
        q, k, v = (torch.zeros_like(t) for t in (q, k, v))
    st = torch.cuda.current_stream().cuda_stream
    flops = 4.0 * B * H * L * L * 128
    out = torch.empty(B, L, H * 128, dtype=torch.bfloat16, device="cuda")

    def run(n):
        for _ in range(n):
            abi.check(lib.mmada_sdpa(h, q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, H, L, st), "sdpa")

    run(args.warm)  # ~1 s of load: the clock governor settles
    torch.cuda.synchronize()
    ms = {f: [] for f in forms}
    for _ in range(args.rounds):   # the two kernel forms interleaved in rounds (same clock, same heat)
        for form in forms:         # 0: round-2 issue order; 1: software-pipelined matrix blocks (the default)
            abi.check(lib.mmada_set_option(b"attention_form", form), "set_option")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(args.iters)
            e1.record()
            torch.cuda.synchronize()
            ms[form].append(e0.elapsed_time(e1) / args.iters)
    lib.mmada_set_option(b"attention_form", -1)
    names = {0: "round-2 issue order", 1: "pipelined matrix blocks", 11: "DIAG no soft-max", 12: "DIAG no MFMA", 13: "DIAG no barrier",
             14: "static priority, second workgroup"}
    for form in forms:
        t = sorted(ms[form])[len(ms[form]) // 2]
        print(f"B={B} L={L} form {form} ({names.get(form, '?')}): median {t * 1e3:.1f} us per "
              f"mmada_sdpa call (incl. 3 layout kernels) = {flops / t / 1e9:.0f} TF lower bound over {args.rounds} rounds of {args.iters}")
    ref = torch.nn.functional.scaled_dot_product_attention(q[:, :2].float(), k[:, :2].float(), v[:, :2].float())
    got = out.view(B, L, H, 128)[:, :, :2].permute(0, 2, 1, 3).float()
    print(f"vs fp32 SDPA (2 heads): max |err| {(got - ref).abs().max().item():.3e}")


if __name__ == "__main__":
    main()
