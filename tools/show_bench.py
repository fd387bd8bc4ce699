import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],4), round(d["config"]["job_mfma_frac"],4), {k:(round(v["avg_ms"],3), round(v["tflops"])) for k,v in d["config"]["kernels"].items()})
    except Exception as e: print(f, "ERR", e)
