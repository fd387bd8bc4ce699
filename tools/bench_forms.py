"""A/B of attention forms (and of the GEMM's short row tiles: "1:0" = form 1, short tiles off) inside the headline bench
(short runs): python tools/bench_forms.py 1 0 1:0 1 0"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for f in sys.argv[1:]:
    form, _, short = f.partition(":")
    env = dict(os.environ, MMADA_ATTN_FORM=form, MMADA_GEMM_SHORT_TILES=short or "1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-probe"],
                       capture_output=True, text=True, env=env, timeout=600)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        k = d["config"]["kernels"]
        print("form", f, "img/s", round(d["value"], 5), "ms/step", round(d["ms_per_step"], 1), "attn TF", round(k["flash_attention"]["tflops"]),
              "attn ms", round(k["flash_attention"]["avg_ms"], 4), "gate/up TF", round(k["gate_up_swiglu_gemm"]["tflops"]), "qkv", round(k["qkv_rope_gemm"]["tflops"]), "down", round(k["down_gemm"]["tflops"]),
              "vq enc/dec ms", d["config"]["vq_encode_ms"], d["config"]["vq_decode_ms"], d["config"]["rocm_smi_during_run"], flush=True)
    except Exception as e:
        print("form", f, "failed:", e, p.stderr[-500:])
