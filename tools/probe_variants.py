"""MFMA-only probe variants (mmada_set_option("probe_variant", v)) on random bf16 operands, each ~1.5 s after a common warm-up,
interleaved in two rounds: TFLOP/s, sclk and socket power.  v&1: 32x32x16 instead of 16x16x32; v&2: 4 waves/CU instead of 8."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mmada_parallel_amd import abi  # noqa: E402

lib = abi.lib()
dev = torch.device("cuda", 0)
names = {0: "16x16x32, 8 waves/CU", 1: "32x32x16, 8 waves/CU", 2: "16x16x32, 4 waves/CU", 3: "32x32x16, 4 waves/CU"}
out = []
bench.attainable_probe(lib, dev, 0)  # warm
for rnd in range(2):
    for v in (0, 1, 2, 3):
        abi.check(lib.mmada_set_option(b"probe_variant", v), "set_option")
        r = bench.attainable_probe(lib, dev, 0)
        out.append({"round": rnd, "variant": names[v], "tflops": round(r["tflops"], 1), "smi": r["rocm_smi"]})
        print(json.dumps(out[-1]), flush=True)
abi.check(lib.mmada_set_option(b"probe_variant", 0), "set_option")
