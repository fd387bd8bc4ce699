import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmada_parallel_amd import abi, synth
lib = abi.lib()
cfg = synth.CFG_8B
c = abi.MmadaCfg(d_model=cfg["d_model"], n_layers=1, n_heads=32, n_kv_heads=32, head_dim=128, mlp_hidden=12288, vocab=134656, max_seq=4096, rms_eps=1e-5, rope_theta=500000.0, tp_rank=0, tp_size=1, mask_token_id=126336, text_vocab_size=126356, codebook_size=8192, reserved=0)
inv = (C.c_float * 64)(*(1.0 / (500000.0 ** (torch.arange(0, 128, 2, dtype=torch.float) / 128))).tolist())
h = C.c_void_p(); abi.check(lib.mmada_create(C.byref(c), inv, C.byref(h)), "create")
nb = lib.mmada_workspace_bytes(h, 2, 2438); ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
abi.check(lib.mmada_set_workspace(h, (ws.data_ptr() + 255) // 256 * 256, nb), "ws")
st = torch.cuda.current_stream().cuda_stream
for (B, H, Hkv, L) in [(2, 8, 4, 2438), (1, 32, 32, 2438), (2, 32, 32, 2438), (1, 8, 8, 1000)]:
    torch.manual_seed(1000 + L)
    q = torch.randn(B, H, L, 128).to(torch.bfloat16).cuda(); k = torch.randn(B, Hkv, L, 128).to(torch.bfloat16).cuda(); v = torch.randn(B, Hkv, L, 128).to(torch.bfloat16).cuda()
    k[0, 0, L // 2] *= 6.0
    outs = []
    for form in (1, 2, 2):
        lib.mmada_set_option(b"attention_form", form)
        out = torch.full((B, L, H * 128), float("nan"), dtype=torch.bfloat16, device="cuda")
        abi.check(lib.mmada_sdpa(h, q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, Hkv, L, st), "sdpa")
        torch.cuda.synchronize(); outs.append(out)
    d = (outs[0] != outs[1]) | (outs[0].isnan() != outs[1].isnan())
    d = d.view(B, L, H, 128)
    print(f"B={B} H={H} Hkv={Hkv} L={L}: differing elements {int(d.sum())}; run-to-run differing {int((outs[1]!=outs[2]).sum())}")
    if d.any():
        rows = d.any(-1)  # B, L, H
        for b in range(B):
            for hh in range(H):
                r = rows[b, :, hh].nonzero()[:, 0]
                if len(r):
                    qbs = sorted(set((r // 32).tolist()))
                    e = (outs[0].float() - outs[1].float()).view(B, L, H, 128)[b, :, hh].abs().max().item()
                    cols = d[b, :, hh].any(0).nonzero()[:, 0].tolist()
                    print(f"  b={b} h={hh}: {len(r)} rows, q-blocks {qbs[:12]}{'...' if len(qbs) > 12 else ''} max|diff| {e:.3e} cols {cols[:8]}..{cols[-1]} ({len(cols)})")
lib.mmada_set_option(b"attention_form", -1)
