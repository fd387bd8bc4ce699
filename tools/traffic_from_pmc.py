#!/usr/bin/env python
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE / TCC hit-miss PMC summaries (tools/pmc_summary.py output).

    python tools/traffic_from_pmc.py gpurun_out/r03/pmc_f.txt gpurun_out/r03/pmc_w.txt gpurun_out/r03/pmc_t.txt > profiles/traffic.json

Dominant kernel = the gate/up + SiLU*mul GEMM (EPI_SWIGLU): gemm8_kernel<2, Gemm8<320, 256, ...>>, launched at M = 2440
(768 workgroups of 512 threads: grid 393216) and M = 4880 (1536 workgroups: grid 786432).  gfx950 correction of the guide's
HBM section: FETCH_SIZE (KiB) counts 128-byte requests at 64 bytes, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Algorithmic bytes per launch: A [M, 4096] + W [24576, 4096] read once + C [M, 12288] written, bf16."""
import json
import re
import sys


def parse(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", line)
            if m and cur:
                out[cur][m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return out


def find(tab, grid):
    for name, vals in tab.items():
        if "gemm8_kernel<2" in name and f"grid={grid}" in name:
            return vals
    raise SystemExit(f"gate/up kernel with grid {grid} not found")


def main(f_path, w_path, t_path=None):
    f, w = parse(f_path), parse(w_path)
    t = parse(t_path) if t_path else None
    res = {"_comment": "HBM-side bytes per launch of the dominant kernel (gate/up GEMM + SiLU*mul, gemm8_kernel<EPI_SWIGLU, 320x256>), from "
                       "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB) of `bench.py --no-cpu-baseline --no-probe --text-steps 8 "
                       "--timesteps 4 --warmup 0` (tools/profile_round3.sh). gfx950 correction per MI355X_MICROARCH.md HBM section: "
                       "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. FETCH_SIZE counts L2->fabric requests INCLUDING Infinity-Cache hits.",
           "kernel": "gate_up_swiglu_gemm", "per_launch_bytes": {}, "algorithmic_bytes": {}, "launch_mix": {"M2438": 128, "M4876": 64},
           "l2_hit_rate": {}}
    for key, M, grid in (("M2438", 2440, 768 * 512), ("M4876", 4880, 1536 * 512)):
        fv, wv = find(f, grid)["FETCH_SIZE"][1], find(w, grid)["WRITE_SIZE"][1]
        res["per_launch_bytes"][key] = (2.0 * fv + wv) * 1024.0
        res["algorithmic_bytes"][key] = (M * 4096 + 24576 * 4096 + M * 12288) * 2
        if t:
            tv = find(t, grid)
            res["l2_hit_rate"][key] = tv["TCC_HIT_sum"][1] / (tv["TCC_HIT_sum"][1] + tv["TCC_MISS_sum"][1])
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:4])
