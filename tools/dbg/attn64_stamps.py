"""Phase cycle stamps of attention64 (tuning build, VAR bit 7): mean shader cycles per key tile and phase."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MMADA_MI355X_LIB"] = os.path.join(ROOT, "tools", "libmmada_mi355x_tune.so")
import numpy as np
import torch
from mmada_parallel_amd import abi, synth
lib = abi.lib()
raw = C.CDLL(os.environ["MMADA_MI355X_LIB"])
cfg = synth.CFG_8B
c = abi.MmadaCfg(d_model=cfg["d_model"], n_layers=1, n_heads=32, n_kv_heads=32, head_dim=128, mlp_hidden=12288, vocab=134656, max_seq=4096, rms_eps=1e-5, rope_theta=500000.0, tp_rank=0, tp_size=1, mask_token_id=126336, text_vocab_size=126356, codebook_size=8192, reserved=0)
inv = (C.c_float * 64)(*(1.0 / (500000.0 ** (torch.arange(0, 128, 2, dtype=torch.float) / 128))).tolist())
h = C.c_void_p(); abi.check(lib.mmada_create(C.byref(c), inv, C.byref(h)), "create")
nb = lib.mmada_workspace_bytes(h, 2, 2438); ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
abi.check(lib.mmada_set_workspace(h, (ws.data_ptr() + 255) // 256 * 256, nb), "ws")
st = torch.cuda.current_stream().cuda_stream
L = 2438
names = ["A", "B-max", "decide", "B-exp", "-", "C", "wait vm/lgkm", "barrier"]
for B in (1, 2):
    q, k, v = (torch.randn(B, 32, L, 128, device="cuda").to(torch.bfloat16) for _ in range(3))
    out = torch.empty(B, L, 32 * 128, dtype=torch.bfloat16, device="cuda")
    for form in (148,) + tuple(int(x) for x in sys.argv[1:]):
        lib.mmada_set_option(b"attention_form", form)
        for _ in range(30):
            abi.check(lib.mmada_sdpa(h, q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, 32, 32, L, st), "sdpa")
        torch.cuda.synchronize()
        buf = np.zeros(1024 * 4 * 8, dtype=np.uint64)
        assert raw.mmada_tune_attn64_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
        s = buf.reshape(1024, 4, 8).astype(np.float64)
        nfull = 256 if B == 1 else 512
        ntile = 39
        for nm, sel in (("full passes", s[:min(nfull, 1024)]), ("half passes", s[nfull:nfull + 128 * B])):
            if sel.size == 0:
                continue
            per = sel.mean(axis=(0, 1)) / ntile
            print(f"B={B} form {form} {nm}: cycles per tile by phase: " + ", ".join(f"{n} {x:.0f}" for n, x in zip(names, per) if n != "-") + f" | total {per.sum():.0f}")
lib.mmada_set_option(b"attention_form", -1)
