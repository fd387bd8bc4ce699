"""GPU parity tests of the individual HIP kernels, called through the C-ABI (libmmada_mi355x.so).

Integer / index results (argmax, selections, written ids) must be BIT-EXACT against the CPU oracle.
Floating-point kernels (GEMM, RMSNorm, attention) are compared with an fp32 evaluation of the same op on the same
bf16 inputs; the tolerance is stated at each assert (bf16 has 8 significant bits: 1 ulp = 2^-8 relative).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import os

from helpers import ROOT, bits
from mmada_parallel_amd import abi, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def st():
    return abi.stream_ptr()


@pytest.fixture(scope="module")
def handle():
    """A tiny handle: the sampler / sdpa entry points only need cfg (mask id, vocab offsets) and a workspace."""
    lib = abi.lib()
    c = abi.MmadaCfg(d_model=256, n_layers=1, n_heads=2, n_kv_heads=2, head_dim=128, mlp_hidden=512, vocab=134656,
                     max_seq=1024, rms_eps=1e-5, rope_theta=500000.0, tp_rank=0, tp_size=1, mask_token_id=synth.MASK,
                     text_vocab_size=synth.TEXT_VOCAB, codebook_size=synth.CODEBOOK, reserved=0)
    h = C.c_void_p()
    abi.check(lib.mmada_create(C.byref(c), None, C.byref(h)), "create")
    ws = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    base = (ws.data_ptr() + 255) // 256 * 256
    abi.check(lib.mmada_set_workspace(h, base, ws.numel() - 256), "set_workspace")
    yield h
    lib.mmada_destroy(h)
    del ws


# ---------------------------------------------------------------------------------------------------------------- GEMM
def test_gemm_every_row_tile_height():
    """Each BM configuration of the production kernel (forced through MMADA_GEMM_BM in a fresh process)."""
    import subprocess
    import sys

    code = (
        "import torch, sys; sys.path.insert(0, %r)\n"
        "from mmada_parallel_amd import abi\n"
        "torch.manual_seed(0)\n"
        "for (M, N, K) in [(517, 768, 192), (2438, 4096, 1024)]:\n"
        "    A = torch.randn(M, K, device='cuda').to(torch.bfloat16); W = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)\n"
        "    C = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device='cuda')\n"
        "    abi.check(abi.lib().mmada_gemm_bt(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, abi.stream_ptr()))\n"
        "    ref = A.float() @ W.float().t(); err = (C.float() - ref).abs()\n"
        "    assert (err <= 2.0 ** -8 * ref.abs() + 1e-3 * ref.abs().max()).all(), err.max().item()\n"
        "print('ok')\n" % ROOT)
    for bm in (128, 160, 192, 224, 256):
        env = dict(os.environ, MMADA_GEMM_BM=str(bm))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "ok" in out.stdout, f"BM={bm}: {out.stderr[-800:]}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 384, 256), (77, 200, 128), (1000, 8192, 256),
                                   (2438, 4096, 4096), (2440, 12288, 4096), (4876, 4096, 12288)])
def test_gemm_bt(M, N, K):
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)  # asymmetric operands: catches transposed C
    Cout = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    abi.check(abi.lib().mmada_gemm_bt(A.data_ptr(), W.data_ptr(), Cout.data_ptr(), M, N, K, st()), "gemm")
    ref = A.float() @ W.float().t()
    err = (Cout.float() - ref).abs()
    scale = ref.abs().max().item()
    # fp32 accumulation, one bf16 rounding of the result: |err| <= 2^-9 |ref| + accumulation-order noise
    assert torch.isfinite(Cout.float()).all()
    assert (err <= 2.0 ** -8 * ref.abs() + 1e-3 * scale).all(), f"max err {err.max().item()} scale {scale}"


# ------------------------------------------------------------------------------------------------------------- RMSNorm
@pytest.mark.parametrize("rows,d", [(5, 256), (70, 4096), (2440, 4096)])
def test_rmsnorm(rows, d):
    from oracle import llada_oracle

    torch.manual_seed(rows)
    x = (torch.randn(rows, d) * 3).to(torch.bfloat16)
    w = (1 + 0.02 * torch.randn(d)).to(torch.bfloat16)
    out = torch.empty(rows, d, dtype=torch.bfloat16, device=DEV)
    xd, wd = x.to(DEV), w.to(DEV)
    abi.check(abi.lib().mmada_rmsnorm(xd.data_ptr(), wd.data_ptr(), out.data_ptr(), rows, d, 1e-5, st()), "rmsnorm")
    ref = llada_oracle.rms_norm(x, w, 1e-5)
    diff = (out.cpu().float() - ref.float()).abs()
    # same rounding sequence as the reference; the fp32 sum order may move the normalised value across one bf16
    # boundary (1 ulp = 2^-8..2^-7 relative) before the second rounding of w*y
    assert (diff <= 2.0 ** -6 * ref.float().abs() + 1e-6).all()
    assert (bits(out) != bits(ref)).float().mean() < 5e-3


def test_f2bf_matches_torch_on_every_bit_pattern():
    """f2bf (csrc/common.h: the hardware conversion v_cvt_pk_bf16_f32 since round 4) against torch's fp32 -> bf16 cast on
    ALL 2^32 fp32 bit patterns: equal bits for every finite value (denormals included) and both infinities; a NaN stays a
    NaN (payloads may differ: torch canonicalises)."""
    lib = abi.lib()
    chunk = 1 << 27
    for c in range(32):
        bits = torch.arange(c * chunk, (c + 1) * chunk, dtype=torch.int64, device=DEV).to(torch.int32)  # wraps to the signed pattern
        x = bits.view(torch.float32)
        out = torch.empty(chunk, dtype=torch.int16, device=DEV)
        abi.check(lib.mmada_probe_f2bf(x.data_ptr(), out.data_ptr(), chunk, st()), "probe_f2bf")
        ref = x.to(torch.bfloat16).view(torch.int16)
        nan = torch.isnan(x)
        assert torch.equal(out[~nan], ref[~nan]), f"chunk {c}: {int((out[~nan] != ref[~nan]).sum())} finite values differ"
        if bool(nan.any()):
            assert bool(torch.isnan(out[nan].view(torch.bfloat16)).all()), f"chunk {c}: a NaN became a number"
        del bits, x, out, ref, nan


def test_swiglu_table_is_bit_identical():
    """The 8-phase SwiGLU epilogue reads SiLU of the bf16-rounded gate value from a table (gemm_epilogue.h: SiluLut, filled on the
    device by the function it replaces; untabulated values take the evaluating path).  With A = identity the pre-activations
    are the weight entries themselves, so every finite bf16 bit pattern goes through the epilogue as a gate value (tabulated ones
    in whole waves, the rest — zeros, denormals, tiny, huge — mixed in and in waves of their own; Inf / NaN in a third run), against
    two up values.  Table on / table off / the 16-wave kernel (always evaluates) must agree bit for bit, every tile configuration, and the
    finite results must be torch's silu(gate) * up rounded the way the reference rounds (bf16 after the Linear, after SiLU, after
    the product)."""
    lib = abi.lib()
    K = M = 256
    A = torch.eye(M, K, dtype=torch.bfloat16, device=DEV)
    pat = torch.arange(65536, dtype=torch.int32, device=DEV).to(torch.int16).view(torch.bfloat16)     # every bf16 value
    perm = torch.randperm(65536, generator=torch.Generator().manual_seed(5)).to(DEV)
    # 0 x Inf = NaN: a non-finite weight poisons its whole column of A·W^T, so the sweeps use the 65 280 finite patterns (the
    # other 256 slots hold 1.0) and a third run keeps Inf / NaN in (bit identity of the paths only)
    finite = torch.where(torch.isfinite(pat.float()), pat, torch.ones_like(pat))
    # the table's own domain, |x| in [2^-14, 2^5): 4 864 patterns, tiled and shuffled — every wave of this run takes the table path
    # (in the full sweeps above almost every wave holds an untabulated value and evaluates)
    key = torch.arange(65536, dtype=torch.int32, device=DEV) % (2 * 19 * 128)
    tab = (((key & 1) << 15) | ((key >> 1) + (113 << 7))).to(torch.int16).view(torch.bfloat16)
    for gate_vals, up_val, against_torch in ((finite, 1.0, True), (finite[perm], -0.37109375, True), (pat[perm], 1.0, False),
                                             (tab[perm], 1.0, True), (tab, 2.5, True)):
        # pre-activation P[m][n] = W[n][m]; columns come in groups of 32 = [16 gate | 16 up]; 256 groups x 16 x 256 rows = 65 536 gates
        N = 2 * 256
        gate = gate_vals.view(256, 256)                      # gate[m][j]: row m, gate column j
        W = torch.empty(N, K, dtype=torch.bfloat16, device=DEV)
        Wv = W.view(N // 32, 2, 16, K)                        # [group][gate | up][16][K]
        Wv[:, 0] = gate.t().contiguous().view(16, 16, K)      # gate column j = 16 * group + i  ->  W row (group, 0, i)
        Wv[:, 1] = up_val
        outs = {}
        try:
            for name, cfg, lut in (("table", 0, 1), ("evaluated", 0, 0), ("table 256x256", 1, 1), ("table 160x256", 2, 1),
                                   ("table 320x128", 3, 1), ("16-wave", 1256, 1)):
                abi.check(lib.mmada_set_option(b"gemm_config", cfg), "set_option")
                abi.check(lib.mmada_set_option(b"gemm_silu_lut", lut), "set_option")
                C = torch.full((M, N // 2), float("nan"), dtype=torch.bfloat16, device=DEV)
                abi.check(lib.mmada_gemm_swiglu_bt(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, st()), "swiglu")
                torch.cuda.synchronize()
                outs[name] = C
        finally:
            lib.mmada_set_option(b"gemm_config", -1)
            lib.mmada_set_option(b"gemm_silu_lut", 1)
        base = outs["evaluated"].view(torch.int16)
        for name, C in outs.items():
            assert torch.equal(C.view(torch.int16), base), f"{name}: {int((C.view(torch.int16) != base).sum())} of 65536 outputs differ"
        if not against_torch:
            continue
        g32 = gate.float()
        ref = (torch.nn.functional.silu(g32).to(torch.bfloat16).float() * up_val).to(torch.bfloat16)
        fin = torch.isfinite(ref.float()) & torch.isfinite(g32) & (ref.float().abs() > 1e-30)   # sanity check on normal results only
        got = outs["table"]
        # SiLU evaluated in fp32 by two libms may round a half-way case differently: allow one bf16 ulp on a handful, none elsewhere
        d = (got.float() - ref.float()).abs()[fin]
        ulp = (ref.float().abs()[fin] * 2.0 ** -7).clamp_min(1e-40)
        assert bool((d <= ulp).all()) and float((d > 0).float().mean()) < 2e-3, (float(d.max()), float((d > 0).float().mean()))


# ------------------------------------------------------------------------------------------------- GEMM configurations
@pytest.mark.parametrize("M,N,K", [(2440, 1536, 512), (648, 1024, 1024), (8, 256, 256), (4880, 512, 4096), (328, 8200, 256),
                                   (624, 512, 256), (632, 512, 256), (5000, 256, 256)])
def test_gemm_configurations_are_bit_identical(M, N, K):
    """Every tile configuration of both GEMM kernels (8-phase: 320x256, 256x256, 160x256, 320x128 with swapped MFMA operand
    roles and 16-byte epilogue accesses, both read schedules; 16-wave: BM 128..320) accumulates a K-tile at a time in the same order with the same
    MFMA: the planner's choice never changes a bit of the result.  Ragged last tiles in M and N included; M = 2440, 4880, 648,
    328, 624 take the short row tiles of the 320-row configuration (pitch 304), M = 632 and 5000 do not fit them."""
    lib = abi.lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    outs = {}
    try:
        for code in (-1, 0, 1, 2, 3, 1128, 1160, 1192, 1224, 1256, 1320):
            abi.check(lib.mmada_set_option(b"gemm_config", code), "set_option")
            Cc = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            abi.check(lib.mmada_gemm_bt(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, st()), "gemm")
            torch.cuda.synchronize()
            outs[code] = Cc
        # the 320-row configuration once more with every row tile at full height (short row tiles off: gemm8.hip)
        abi.check(lib.mmada_set_option(b"gemm_config", 0), "set_option")
        abi.check(lib.mmada_set_option(b"gemm_short_tiles", 0), "set_option")
        Cc = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        abi.check(lib.mmada_gemm_bt(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, st()), "gemm")
        torch.cuda.synchronize()
        outs["0, full-height row tiles"] = Cc
        abi.check(lib.mmada_set_option(b"gemm_short_tiles", 1), "set_option")
        # other tile ORDERS (groups of GM row tiles x GN column tiles per XCD-round: the FETCH_SIZE sweep of DESIGN.md): the map
        # from workgroup id to tile stays a bijection — every output written exactly once, same bits
        for cfg in (0, 2):
            for order in (408, 216, 301, 1602):
                abi.check(lib.mmada_set_option(b"gemm_config", cfg), "set_option")
                abi.check(lib.mmada_set_option(b"gemm_tile_order", order), "set_option")
                Cc = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
                abi.check(lib.mmada_gemm_bt(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, st()), "gemm")
                torch.cuda.synchronize()
                outs[f"{cfg}, tile order {order}"] = Cc
    finally:
        lib.mmada_set_option(b"gemm_config", -1)
        lib.mmada_set_option(b"gemm_short_tiles", 1)
        lib.mmada_set_option(b"gemm_tile_order", 0)
    ref = (A.float() @ W.float().t())
    assert ((outs[-1].float() - ref).abs() <= 2.0 ** -7 * ref.abs() + 1e-2).all()
    for code, Cc in outs.items():
        assert torch.equal(Cc, outs[-1]), f"configuration {code}: {int((Cc != outs[-1]).sum())} elements differ"


# ----------------------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,H,Hkv,L", [(1, 2, 2, 70), (2, 2, 1, 333), (1, 8, 8, 1000), (1, 2, 2, 64), (1, 1, 1, 1), (1, 8, 8, 2438),
                                       (2, 8, 4, 2438), (1, 2, 2, 129), (1, 2, 2, 192)])
def test_attention_forms_are_bit_identical(handle, B, H, Hkv, L):
    """Form 1 of the attention kernel (waves 4-7 of a workgroup accumulate P·V one key tile late, so that a wave's soft-max sits
    beside its SIMD partner's matrix phase) does per query row exactly the arithmetic of form 0 (every wave in the plain order),
    in the same order: identical bits, for odd and even key-tile counts, partial groups, grouped heads and a forced rescale.
    The shapes also cover 1-3 query groups per wave and workgroup counts per head from 1 to 8 (attention_chunks)."""
    torch.manual_seed(1000 + L)
    q = torch.randn(B, H, L, 128).to(torch.bfloat16).to(DEV)
    k = torch.randn(B, Hkv, L, 128).to(torch.bfloat16).to(DEV)
    v = torch.randn(B, Hkv, L, 128).to(torch.bfloat16).to(DEV)
    k[0, 0, L // 2] *= 6.0   # a spike: forces the deferred-rescale branch in some rows (guide rule 26)
    outs = []
    lib = abi.lib()
    try:
        for form in (0, 1):
            abi.check(lib.mmada_set_option(b"attention_form", form), "set_option")
            out = torch.full((B, L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
            abi.check(lib.mmada_sdpa(handle, q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, Hkv, L, st()), "sdpa")
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        lib.mmada_set_option(b"attention_form", -1)
    assert torch.isfinite(outs[1].float()).all()
    assert torch.equal(outs[0], outs[1]), f"{int((outs[0] != outs[1]).sum())} elements differ"


@pytest.mark.parametrize("H,L", [(8, 1000), (32, 333), (16, 2438)])
def test_attention_is_invariant_to_its_launch_plan(handle, H, L):
    """A query row's arithmetic does not depend on which wave, workgroup or round carries its 16-row group (csrc/attention.hip):
    the launch plan (mmada_attention_plan: workgroups per head, hence groups per wave) changes with the number of (batch, head)
    pairs, so the same sequence inside batches of 1, 2 and 3 runs under different plans — and must give the same bits."""
    torch.manual_seed(77 + L)
    q = torch.randn(3, H, L, 128).to(torch.bfloat16).to(DEV)
    k = torch.randn(3, H, L, 128).to(torch.bfloat16).to(DEV)
    v = torch.randn(3, H, L, 128).to(torch.bfloat16).to(DEV)
    k[:, 0, L // 3] *= 5.0   # some rows take the deferred-rescale branch
    lib = abi.lib()
    groups = (L + 15) // 16
    plans = {B: lib.mmada_attention_plan(B * H, groups, L) for B in (1, 2, 3)}
    outs = {}
    for B in (1, 2, 3):
        out = torch.full((B, L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
        qb, kb, vb = (t[:B].contiguous() for t in (q, k, v))
        abi.check(lib.mmada_sdpa(handle, qb.data_ptr(), kb.data_ptr(), vb.data_ptr(), out.data_ptr(), B, H, H, L, st()), "sdpa")
        torch.cuda.synchronize()
        outs[B] = out
    assert torch.isfinite(outs[3].float()).all()
    assert torch.equal(outs[1][0], outs[3][0]) and torch.equal(outs[2], outs[3][:2]), plans
    if (H, L) == (8, 1000):
        assert len(set(plans.values())) > 1, plans   # the case really exercises different plans


@pytest.mark.parametrize("B,H,Hkv,L", [(1, 2, 2, 70), (2, 2, 1, 333), (1, 4, 4, 1000), (1, 2, 2, 64), (1, 1, 1, 1),
                                       (1, 2, 2, 2438)])
def test_sdpa(handle, B, H, Hkv, L):
    torch.manual_seed(L)
    q = torch.randn(B, H, L, 128).to(torch.bfloat16)
    k = torch.randn(B, Hkv, L, 128).to(torch.bfloat16)
    v = torch.randn(B, Hkv, L, 128).to(torch.bfloat16)
    out = torch.full((B, L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    abi.check(abi.lib().mmada_sdpa(handle, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), out.data_ptr(), B, H, Hkv, L, st()),
              "sdpa")
    kk, vv = k.float(), v.float()
    if Hkv != H:
        kk, vv = kk.repeat_interleave(H // Hkv, 1), vv.repeat_interleave(H // Hkv, 1)
    ref = F.scaled_dot_product_attention(q.float(), kk, vv).transpose(1, 2).reshape(B, L, H * 128)
    got = out.cpu().float()
    assert torch.isfinite(got).all()
    # P is rounded to bf16 before PV (as torch's bf16 flash kernel does): abs error ~ 2^-8 * |v| / sqrt(n_eff)
    assert (got - ref).abs().max().item() < 2e-2, (got - ref).abs().max().item()
    assert (got - ref).abs().mean().item() < 2e-3


# -------------------------------------------------------------------------------------------------------- text sampler
def _text_inputs(B, T, V, L, ts, seed, quantise=None):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, T, V, generator=g) * 2
    if quantise:  # coarse grid -> many exact ties, exercises lowest-index argmax
        logits = (logits * quantise).round() / quantise
    logits = logits.to(torch.bfloat16)
    ids = torch.randint(0, 1000, (B, L), generator=g)
    ids[:, ts:ts + T] = synth.MASK
    for b in range(B):  # a few already-unmasked positions
        ids[b, ts + torch.randperm(T, generator=g)[: T // 5]] = 7 + b
    return logits, ids


@pytest.mark.parametrize("B,T,V,k,quant", [(1, 16, 2560, [3], None), (2, 64, 134656, [5, 0], None),
                                           (2, 256, 134656, [2, 7], 2), (1, 40, 4100, [32], 1), (3, 33, 1000, [1, 2, 3], None)])
def test_text_select_bit_exact(handle, B, T, V, k, quant):
    from oracle import sampler_oracle as so

    L, ts = T + 20, 9
    ld = (V + 7) // 8 * 8
    logits, ids = _text_inputs(B, T, V, L, ts, seed=T + V, quantise=quant)
    lpad = torch.zeros(B, T, ld, dtype=torch.bfloat16)
    lpad[..., :V] = logits
    ld_dev, ids_dev = lpad.to(DEV), ids.to(DEV)
    k_dev = torch.tensor(k, dtype=torch.int32, device=DEV)
    scratch = torch.empty(B * T * 16, dtype=torch.uint8, device=DEV)
    abi.check(abi.lib().mmada_text_select(handle, ld_dev.data_ptr(), None, B, T, V, ld, ids_dev.data_ptr(), L, ts,
                                          k_dev.data_ptr(), scratch.data_ptr(), st()), "text_select")
    ref, conf_ref, x0_ref = so.text_select(logits, None, ids, ts, k)
    assert torch.equal(ids_dev.cpu(), ref)
    conf = scratch[: B * T * 8].view(torch.float64).cpu().view(B, T)
    x0 = scratch[B * T * 8: B * T * 12].view(torch.int32).cpu().view(B, T)
    assert torch.equal(x0, x0_ref)
    m = torch.isfinite(conf_ref)
    assert torch.equal(torch.isfinite(conf), m)
    assert torch.allclose(conf[m], conf_ref[m], rtol=1e-12, atol=0)


def test_text_select_with_noisy_argmax(handle):
    from oracle import sampler_oracle as so

    B, T, V, L, ts = 2, 32, 8192, 60, 5
    logits, ids = _text_inputs(B, T, V, L, ts, seed=11)
    noisy = (logits.float() + torch.randn(B, T, V, generator=torch.Generator().manual_seed(1))).to(torch.bfloat16)
    k = [4, 6]
    ids_dev, lg, nz = ids.to(DEV), logits.to(DEV), noisy.to(DEV)
    k_dev = torch.tensor(k, dtype=torch.int32, device=DEV)
    scratch = torch.empty(B * T * 16, dtype=torch.uint8, device=DEV)
    abi.check(abi.lib().mmada_text_select(handle, lg.data_ptr(), nz.data_ptr(), B, T, V, V, ids_dev.data_ptr(), L, ts,
                                          k_dev.data_ptr(), scratch.data_ptr(), st()), "text_select")
    ref, _, _ = so.text_select(logits, noisy, ids, ts, k)
    assert torch.equal(ids_dev.cpu(), ref)


# ------------------------------------------------------------------------------------------------------- image sampler
@pytest.mark.parametrize("B,N,CB,cs,ci,quant", [(1, 64, 8192, 0.0, 4.0, None), (2, 100, 8192, 2.5, 4.0, None),
                                                (1, 1024, 8192, 2.3, 0.0, None), (1, 50, 512, 0.0, 0.0, 4),
                                                (3, 16, 8192, 3.0, 4.0, 2)])
def test_image_probs_bit_exact(handle, B, N, CB, cs, ci, quant):
    from oracle import sampler_oracle as so

    g = torch.Generator().manual_seed(N + CB)
    c = torch.randn(B, N, CB, generator=g) * 1.5
    if quant:
        c = (c * quant).round() / quant
    c = c.to(torch.bfloat16)
    ut = (c.float() + torch.randn(B, N, CB, generator=g) * 0.5).to(torch.bfloat16)
    ui = (c.float() + torch.randn(B, N, CB, generator=g) * 0.5).to(torch.bfloat16)
    cd, utd, uid = c.to(DEV), ut.to(DEV), ui.to(DEV)
    probs = torch.empty(B, N, CB, dtype=torch.bfloat16, device=DEV)
    am = torch.empty(B, N, dtype=torch.int32, device=DEV)
    pm = torch.empty(B, N, dtype=torch.bfloat16, device=DEV)
    abi.check(abi.lib().mmada_image_probs(handle, cd.data_ptr(), utd.data_ptr(), uid.data_ptr(), B, N, CB, cs, ci,
                                          probs.data_ptr(), am.data_ptr(), pm.data_ptr(), st()), "image_probs")
    am_r, pm_r, pr_r = so.image_probs(c, ut, ui, cs, ci, want_probs=True)
    assert torch.equal(am.cpu(), am_r)
    assert torch.equal(bits(pm), bits(pm_r))
    assert torch.equal(bits(probs), bits(pr_r))


@pytest.mark.parametrize("case", ["fresh", "half_known", "all_known", "one_unknown", "noise"])
def test_image_commit_bit_exact(handle, case):
    from oracle import sampler_oracle as so

    g = torch.Generator().manual_seed({"fresh": 1, "half_known": 2, "all_known": 3, "one_unknown": 4, "noise": 5}[case])
    B, N, L = 2, 256, 400
    pos = torch.sort(torch.randperm(L - 10, generator=g)[:N]).values + 5
    ids = torch.randint(0, 1000, (B, L), generator=g)
    known = {"fresh": 0, "half_known": N // 2, "all_known": N, "one_unknown": N - 1, "noise": N // 3}[case]
    for b in range(B):
        perm = torch.randperm(N, generator=g)
        ids[b, pos] = synth.MASK
        kn = pos[perm[:known]]
        ids[b, kn] = synth.TEXT_VOCAB + torch.randint(0, synth.CODEBOOK, (known,), generator=g)
    sampled = torch.randint(0, synth.CODEBOOK, (B, N), generator=g, dtype=torch.int32)
    # bf16 probabilities from a coarse set -> massive ties in the log-confidence (stable order matters)
    p = (torch.randint(1, 40, (B, N), generator=g).float() / 4096).to(torch.bfloat16)
    noise = torch.randn(B, N, generator=g).to(torch.bfloat16) if case == "noise" else torch.zeros(B, N, dtype=torch.bfloat16)
    temp = 0.37 if case == "noise" else 0.0
    pos_d, samp_d, p_d, noise_d = pos.to(torch.int32).to(DEV), sampled.to(DEV), p.to(DEV), noise.to(DEV)  # keep alive
    for mlen in (-1, 0, 1, 17, N // 2, N + 5):
        ids_dev = ids.to(DEV)
        ml = torch.tensor([mlen], dtype=torch.int32, device=DEV)
        abi.check(abi.lib().mmada_image_commit(handle, ids_dev.data_ptr(), B, L, pos_d.data_ptr(), N, samp_d.data_ptr(),
                                               p_d.data_ptr(), noise_d.data_ptr(), temp, ml.data_ptr(), synth.TEXT_VOCAB,
                                               synth.CODEBOOK, st()), "image_commit")
        ref = so.image_commit(ids, pos.to(torch.int32), sampled, p, noise, temp, mlen)
        assert torch.equal(ids_dev.cpu(), ref), f"{case} mlen={mlen}"


def test_log_conf_all_bf16_probabilities(handle, golden_dir):
    """image_commit's bf16 log-confidence for EVERY non-negative bf16 probability, checked through its observable
    effect: ranking N tokens whose probabilities enumerate all bit patterns must give the oracle's ids."""
    from oracle import sampler_oracle as so

    vals = torch.arange(0, 0x7f80, dtype=torch.int32)
    chunks = vals.split(4096)
    for ci, ch in enumerate(chunks):
        N = ch.numel()
        L = N + 4
        pos = torch.arange(2, 2 + N, dtype=torch.int32)
        ids = torch.full((1, L), synth.MASK, dtype=torch.long)
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(ci))
        p = ch.to(torch.int16).view(torch.bfloat16)[perm].view(1, N)
        sampled = torch.zeros(1, N, dtype=torch.int32)
        noise = torch.zeros(1, N, dtype=torch.bfloat16)
        pos_d, samp_d, p_d, noise_d = pos.to(DEV), sampled.to(DEV), p.to(DEV), noise.to(DEV)  # keep alive
        for mlen in (N // 3, N - 2):
            ids_dev = ids.to(DEV)
            ml = torch.tensor([mlen], dtype=torch.int32, device=DEV)
            abi.check(abi.lib().mmada_image_commit(handle, ids_dev.data_ptr(), 1, L, pos_d.data_ptr(), N,
                                                   samp_d.data_ptr(), p_d.data_ptr(), noise_d.data_ptr(), 0.0,
                                                   ml.data_ptr(), synth.TEXT_VOCAB, synth.CODEBOOK, st()), "image_commit")
            assert torch.equal(ids_dev.cpu(), so.image_commit(ids, pos, sampled, p, noise, 0.0, mlen))


def test_lfq_gather(handle):
    from oracle import sampler_oracle as so

    idx = torch.randint(0, 2 ** 13, (2, 1024))
    idx_d = idx.to(DEV)
    out = torch.empty(2, 13, 1024, dtype=torch.float32, device=DEV)
    abi.check(abi.lib().mmada_lfq_gather(handle, idx_d.data_ptr(), 2, 1024, 13, 1, out.data_ptr(), st()), "lfq")
    assert torch.equal(out.cpu(), so.lfq_gather(idx, 13))
    outb = torch.empty(2, 13, 1024, dtype=torch.bfloat16, device=DEV)
    abi.check(abi.lib().mmada_lfq_gather(handle, idx_d.data_ptr(), 2, 1024, 13, 0, outb.data_ptr(), st()), "lfq")
    assert torch.equal(outb.cpu().float(), so.lfq_gather(idx, 13))


# ---------------------------------------------------------------------------------------------------- M-variant kernels
@pytest.mark.parametrize("B,T,V,k,cfg,with_x0", [(1, 16, 2560, [3], 1.5, False), (2, 32, 134656, [4, 9], 0.7, False),
                                                 (1, 24, 4100, [24], 0.0, False), (2, 16, 2560, [5, 2], 2.3, True)])
def test_text_select_cfg_bit_exact(handle, B, T, V, k, cfg, with_x0):
    from oracle import sampler_oracle as so

    L, ts = T + 12, 7
    ld = (V + 7) // 8 * 8
    cond, ids = _text_inputs(B, T, V, L, ts, seed=3 * T + V)
    unc = (cond.float() + torch.randn(B, T, V, generator=torch.Generator().manual_seed(V)) * 0.7).to(torch.bfloat16)
    x0_in = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(1), dtype=torch.int32) if with_x0 else None
    cp, up = torch.zeros(B, T, ld, dtype=torch.bfloat16), torch.zeros(B, T, ld, dtype=torch.bfloat16)
    cp[..., :V], up[..., :V] = cond, unc
    c_d, u_d, ids_d = cp.to(DEV), up.to(DEV), ids.to(DEV)
    x0_d = x0_in.to(DEV) if with_x0 else None
    k_d = torch.tensor(k, dtype=torch.int32, device=DEV)
    scratch = torch.empty(B * T * 16, dtype=torch.uint8, device=DEV)
    abi.check(abi.lib().mmada_text_select_cfg(handle, c_d.data_ptr(), u_d.data_ptr(), cfg, abi.ptr(x0_d), B, T, V, ld,
                                              ids_d.data_ptr(), L, ts, k_d.data_ptr(), scratch.data_ptr(), st()), "text_cfg")
    ref, conf_ref, x0_ref = so.text_select_cfg(cond, unc, cfg, ids, ts, k, x0_in=x0_in)
    assert torch.equal(ids_d.cpu(), ref)
    conf = scratch[: B * T * 8].view(torch.float64).cpu().view(B, T)
    m = torch.isfinite(conf_ref)
    assert torch.equal(torch.isfinite(conf), m) and torch.allclose(conf[m], conf_ref[m], rtol=1e-12, atol=0)


@pytest.mark.parametrize("B,N,CB,cfg", [(1, 64, 8192, 3.5), (2, 40, 512, 2.0), (1, 16, 8192, 0.3)])
def test_image_probs_m_bit_exact(handle, B, N, CB, cfg):
    from oracle import sampler_oracle as so

    g = torch.Generator().manual_seed(N * CB)
    c = (torch.randn(B, N, CB, generator=g) * 1.5).to(torch.bfloat16)
    u = (c.float() + torch.randn(B, N, CB, generator=g) * 0.5).to(torch.bfloat16)
    c_d, u_d = c.to(DEV), u.to(DEV)
    probs = torch.empty(B, N, CB, dtype=torch.bfloat16, device=DEV)
    am = torch.empty(B, N, dtype=torch.int32, device=DEV)
    pm = torch.empty(B, N, dtype=torch.bfloat16, device=DEV)
    abi.check(abi.lib().mmada_image_probs_m(handle, c_d.data_ptr(), u_d.data_ptr(), B, N, CB, cfg, probs.data_ptr(),
                                            am.data_ptr(), pm.data_ptr(), st()), "image_probs_m")
    am_r, pm_r, pr_r = so.image_probs_m(c, u, cfg)
    assert torch.equal(am.cpu(), am_r) and torch.equal(bits(pm), bits(pm_r)) and torch.equal(bits(probs), bits(pr_r))


@pytest.mark.parametrize("known", [0, 100, 255])
def test_image_commit_m_bit_exact(handle, known):
    from oracle import sampler_oracle as so

    g = torch.Generator().manual_seed(known + 9)
    B, N, L, tv = 2, 256, 300, 2048
    pos = torch.arange(20, 20 + N, dtype=torch.int32)
    ids = torch.randint(0, 1000, (B, L), generator=g)
    for b in range(B):
        ids[b, 20:20 + N] = synth.MASK
        kn = torch.randperm(N, generator=g)[:known] + 20
        ids[b, kn] = tv + torch.randint(0, 512, (known,), generator=g)
    sampled = torch.randint(0, 512, (B, N), generator=g, dtype=torch.int32)
    p = (torch.randint(0, 30, (B, N), generator=g).float() / 2048).to(torch.bfloat16)     # includes p == 0, heavy ties
    gum = (-torch.log(-torch.log(torch.rand(B, N, generator=g).clamp(min=1e-20)))).to(torch.bfloat16)
    pos_d, s_d, p_d, g_d = pos.to(DEV), sampled.to(DEV), p.to(DEV), gum.to(DEV)
    for temp in (0.0, 0.61):
        for mlen in (-1, 1, 40, N + 3):
            ids_d = ids.to(DEV)
            ml = torch.tensor([mlen], dtype=torch.int32, device=DEV)
            abi.check(abi.lib().mmada_image_commit_m(handle, ids_d.data_ptr(), B, L, pos_d.data_ptr(), N, s_d.data_ptr(),
                                                     p_d.data_ptr(), g_d.data_ptr(), temp, ml.data_ptr(), tv, st()),
                      "image_commit_m")
            ref = so.image_commit_m(ids, pos, sampled, p, gum, temp, mlen, text_vocab=tv)
            assert torch.equal(ids_d.cpu(), ref), f"known={known} temp={temp} mlen={mlen}"
