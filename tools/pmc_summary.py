#!/usr/bin/env python
"""Average rocprofv3 --pmc counters per kernel from a *_counter_collection.csv (csv output format)."""
import csv
import re
import sys
from collections import defaultdict


def main(path, pat="gemm|attn"):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if not re.search(pat, name):
                continue
            name = re.sub(r"\(anonymous namespace\)::|void ", "", name)[:70] + f" grid={r['Grid_Size']}"
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[name]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name, cs in acc.items():
        print(name)
        for c, v in sorted(cs.items()):
            print(f"    {c:28s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")


if __name__ == "__main__":
    main(*sys.argv[1:])
