#!/usr/bin/env python
"""What a CU partition costs and what it buys, measured on ONE GPU (round 5; csrc/tp_comm.hip: mmada_comm_set_partition).

The 8-phase GEMM's 320x256 tile fills a CU (2 waves x 256 VGPRs per SIMD, 144+ KiB LDS): no exchange wave can be resident
beside it, so without a partition "the exchange runs under the next GEMM" means the high-priority exchange stream gets a CU
whenever a GEMM workgroup retires.  With a partition the exchange stream owns n CUs and the GEMMs run on the other 256 - n.

Measured here, on the kernels of one tensor-parallel rank at TP = 8 weak scaling (M = 8 x 2440 / 2 rows per chunk):
  (a) the per-rank GEMMs on the masked compute stream vs an unmasked stream: the deterministic cost of the partition;
  (b) the exchange's kernels (hand-off, owner reduce + RMSNorm, gather; a one-rank group: local operands) alone on the exchange
      stream, and WHILE the compute stream runs GEMMs back to back — with the partition and with the plain high-priority stream:
      how long an exchange takes under load and how much the GEMMs slow down.
The remote-load latency of xGMI is not in these numbers (one device); what they show is whether the two streams really run
side by side.   python tools/tp_overlap_probe.py [--cus 16,32]
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmada_parallel_amd import LLaDAForMultiModalGeneration, abi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cus", default="16,32")
    ap.add_argument("--rows", type=int, default=9760, help="stream rows of one chunk of a TP = 8, batch-8 forward")
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    dev = "cuda:0"
    lib = abi.lib()
    cfg = dict(synth.CFG_8B, n_layers=1)
    sd = synth.synthetic_state_dict(cfg, seed=3, device="cpu")
    model = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=dev, max_batch=1)
    h = model._handle
    B, L = 1, 1220          # the owner share of one chunk at TP = 8: 9760 / 8 rows (a one-rank group owns every row it has)
    M = B * 1224
    abi.check(lib.mmada_set_option(b"tp_allow_single_rank", 1), "set_option")
    abi.check(lib.mmada_comm_create(h, M, None), "comm_create")
    arr = (C.c_void_p * 1)(h.value)
    abi.check(lib.mmada_comm_connect_local(h, arr), "connect_local")
    model._comm_in_library, model.tp_collective, model._comm_rows = True, "pull", M
    ids = torch.zeros((B, L), dtype=torch.long, device=dev)
    model._ensure_ws(B, L)
    st0 = abi.stream_ptr()
    abi.check(lib.mmada_embed(h, ids.data_ptr(), B, L, st0), "embed")
    w = torch.ones(4096, dtype=torch.bfloat16, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    # per-rank shapes at TP = 8: gate/up N = 3072, K = 4096 ; down N = 4096, K = 1536 ; rows of one chunk
    R = args.rows
    A = torch.randn(R, 4096, device=dev, generator=g).to(torch.bfloat16)
    Wg = (torch.randn(3072, 4096, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    Cg = torch.empty(R, 3072, dtype=torch.bfloat16, device=dev)
    flops = 2.0 * R * 3072 * 4096
    out = {"rows_per_chunk": R, "gemm": "gate/up per rank at TP=8 (N=3072, K=4096), plain store epilogue", "variants": {}}

    def streams():
        ex, cm = C.c_void_p(), C.c_void_p()
        abi.check(lib.mmada_comm_streams(h, C.byref(ex), C.byref(cm)), "comm_streams")
        return ex.value, cm.value

    def run_gemms(stream_ptr, n):
        for _ in range(n):
            abi.check(lib.mmada_gemm_bt(A.data_ptr(), Wg.data_ptr(), Cg.data_ptr(), R, 3072, 4096, stream_ptr), "gemm")

    def run_exchanges(stream_ptr, n):
        for _ in range(n):
            abi.check(lib.mmada_comm_exchange(h, w.data_ptr(), stream_ptr), "exchange")

    def timed(fn_list):
        """fn_list: [(stream_ptr, callable)] enqueued back to back; per-stream elapsed ms."""
        evs = []
        torch.cuda.synchronize()
        for sp, fn in fn_list:
            s = torch.cuda.ExternalStream(sp) if sp else torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            evs.append((e0, e1, s, sp, fn))
        for e0, e1, s, sp, fn in evs:
            fn(sp if sp else abi.stream_ptr())
        for e0, e1, s, sp, fn in evs:
            e1.record(s)
        torch.cuda.synchronize()
        return [e0.elapsed_time(e1) for e0, e1, *_ in evs]

    n = args.reps
    run_gemms(st0, 200)   # clocks
    torch.cuda.synchronize()
    base = timed([(None, lambda sp: run_gemms(sp, n))])[0] / n
    out["gemm_unmasked_us"] = base * 1e3
    out["gemm_unmasked_tflops"] = flops / (base * 1e-3) / 1e12
    # what a CU-masked QUEUE costs by itself: the same GEMM on a plain second stream, on a stream masked to ALL 256 CUs, and
    # on streams masked to 248 / 240 / 224 CUs (the same number removed from every XCD: bit i = CU i / 8 of XCD i % 8)
    hip = C.CDLL("libamdhip64.so")
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (ncu + 31) // 32
    out["gemm_on_other_queues"] = {}

    def masked_stream(first_bit):
        m = (C.c_uint32 * words)()
        for b in range(first_bit, ncu):
            m[b // 32] |= 1 << (b % 32)
        st = C.c_void_p()
        assert hip.hipExtStreamCreateWithCUMask(C.byref(st), words, m) == 0
        return st
    plain = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(plain), 1) == 0   # hipStreamNonBlocking
    for name, st in [("plain non-blocking stream", plain)] + [(f"mask of {ncu - fb} CUs", masked_stream(fb)) for fb in (0, 8, 16, 32)]:
        run_gemms(st.value, 5)
        t = timed([(st.value, lambda sp: run_gemms(sp, n))])[0] / n
        out["gemm_on_other_queues"][name] = {"us": t * 1e3, "tflops": flops / (t * 1e-3) / 1e12}
        print(json.dumps({name: out["gemm_on_other_queues"][name]}), flush=True)
        hip.hipStreamDestroy(st)
    for cus in [0] + [int(c) for c in args.cus.split(",")]:
        abi.check(lib.mmada_comm_set_partition(h, cus), "set_partition")
        ex, cm = streams()
        v = {}
        cm_ptr = cm if cus else None            # no partition: compute on the caller's (unmasked) stream
        ga = timed([(cm_ptr, lambda sp: run_gemms(sp, n))])[0] / n
        xa = timed([(ex, lambda sp: run_exchanges(sp, n))])[0] / n
        both = timed([(cm_ptr, lambda sp: run_gemms(sp, 3 * n)), (ex, lambda sp: run_exchanges(sp, n))])
        v["gemm_alone_us"] = ga * 1e3
        v["gemm_alone_tflops"] = flops / (ga * 1e-3) / 1e12
        v["exchange_alone_us"] = xa * 1e3
        v["gemm_stream_ms_3n_gemms_with_n_exchanges_beside"] = both[0]
        v["gemm_stream_ms_3n_gemms_alone"] = ga * 3 * n
        v["exchange_stream_us_per_exchange_beside_gemms"] = both[1] / n * 1e3
        v["gemm_slowdown_beside_exchanges"] = both[0] / (ga * 3 * n)
        out["variants"]["no partition (high-priority exchange stream)" if cus == 0 else f"{cus} exchange CUs / {256 - cus} compute CUs"] = v
        print(json.dumps({("none" if cus == 0 else cus): v}), flush=True)
    abi.check(lib.mmada_comm_set_partition(h, 0), "set_partition")
    lib.mmada_set_option(b"tp_allow_single_rank", 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
