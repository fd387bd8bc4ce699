import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24"); os.environ.setdefault("MMADA_TP_TIMEOUT_S", "10")
from helpers import tp_group, tp_each
from mmada_parallel_amd import synth, LLaDAForMultiModalGeneration
DEV = "cuda:0"
nl = int(os.environ.get("NL", "2"))
cfg = dict(synth.CFG_8B, n_layers=nl)
sd = synth.synthetic_state_dict(cfg, seed=3, device=DEV)
job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
ids = job["input_ids"].to(DEV); L = ids.shape[1]
m1 = LLaDAForMultiModalGeneration.from_state_dict(synth.full_config(cfg), sd, device=DEV, max_batch=1)
m1.forward_body(ids); ref = m1.hidden_state()
trow = torch.arange(job["text_start"], job["text_end"], dtype=torch.int32, device=DEV)
for tp in (2, 4, 2, 4, 8):
    for rep in range(2):
        ranks, streams = tp_group(cfg, sd, tp, (L + 7) // 8 * 8)
        tp_each(ranks, streams, lambda m: m.forward_body(ids))
        hid = tp_each(ranks, streams, lambda m: m.hidden_state())
        tl = tp_each(ranks, streams, lambda m: m.head_rows(trow, 0, 4096))
        for r in range(1, tp):
            dh = (hid[0] != hid[r]); dl = (tl[0] != tl[r])
            rows = dh.any(-1)[0].nonzero().flatten()
            print(f"tp={tp} rep={rep} rank{r} vs rank0: hidden diff elems {int(dh.sum())} rows {rows[:8].tolist()}..{rows[-3:].tolist() if rows.numel() else []} n={rows.numel()}; logits diff {int(dl.sum())}")
        e = (hid[0].float() - ref.float()).abs()
        print(f"   vs TP=1: mean rel {(e.mean() / ref.float().abs().mean()).item():.3e}; status {[m.comm_status()['error'] for m in ranks]}")
        del ranks, streams
