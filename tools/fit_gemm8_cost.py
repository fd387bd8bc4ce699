#!/usr/bin/env python
"""Fit the 8-phase kernel's cost model (csrc/gemm.hip: plan) to a tools/gemm_sweep.py table (variants 300-303) and show which
configuration the model would pick per point against the measured best.   python tools/fit_gemm8_cost.py profiles/r04_gemm8_sweep_final.txt"""
import sys
import numpy as np

SH = {"qkv": (12288, 4096), "o": (4096, 4096), "gateup": (24576, 4096), "down": (4096, 12288),
      "qkv8": (1536, 4096), "o8": (4096, 512), "gu8": (3072, 4096), "dn8": (4096, 1536),
      "qkv2": (6144, 4096), "o2": (4096, 2048), "gu2": (12288, 4096), "dn2": (4096, 6144),
      "qkv4": (3072, 4096), "o4": (4096, 1024), "gu4": (6144, 4096), "dn4": (4096, 3072)}
BM = [320, 256, 160, 320]
BN = [256, 256, 256, 128]
rows = []
cols = None
for ln in open(sys.argv[1]):
    p = ln.split()
    if not p:
        continue
    if p[0] == "shape":
        cols = [int(x[1:]) for x in p[2:] if x.startswith("v")]
        continue
    if p[0] in SH and cols and not ln.startswith("#"):
        M = int(p[1])
        for v, tf in zip(cols, p[2:]):
            if 300 <= v <= 303:
                N, K = SH[p[0]]
                rows.append((p[0], M, N, K, v - 300, 2.0 * M * N * K / (float(tf.rstrip("!")) * 1e12) * 1e6))
X, y = [], []
for name, M, N, K, c, us in rows:
    tiles = -(-M // BM[c]) * -(-N // BN[c])
    rounds = -(-tiles // 256)
    area = BM[c] * BN[c] / 256.0
    X.append([rounds * area * (K // 64), rounds, rounds * area])
    y.append(us)
X, y = np.array(X), np.array(y)
# relative least squares
w = 1.0 / y
coef, *_ = np.linalg.lstsq(X * w[:, None], y * w, rcond=None)
print("A8 = %.5f us, D0 = %.2f us, D1 = %.4f us   (relative rms %.1f %%)" % (coef[0], coef[1], coef[2],
      100 * np.sqrt(np.mean(((X @ coef - y) / y) ** 2))))


def show(A8, D0, D1, H8):
    pts = {}
    for (name, M, N, K, c, us) in rows:
        pts.setdefault((name, M), {})[c] = us
    bad = 0
    for (name, M), d in pts.items():
        N, K = SH[name]
        cost = {}
        for c in d:
            tiles = -(-M // BM[c]) * -(-N // BN[c])
            area = BM[c] * BN[c] / 256.0
            cost[c] = -(-tiles // 256) * (A8 * area * (K // 64) * H8[c] + D0 + D1 * area)
        pick = min(cost, key=cost.get)
        best = min(d, key=d.get)
        loss = d[pick] / d[best] - 1
        bad += loss > 0.02
        print(f"  {name:7s} M={M:6d} pick {pick} ({d[pick]:7.1f} us, model {cost[pick]:7.1f})  best {best} ({d[best]:7.1f} us)  loss {100 * loss:4.1f} %")
    return bad


print("current constants:")
show(0.00483, 6.0, 0.03, [1.0, 1.0, 1.08, 1.08])
if len(sys.argv) > 2:
    A8, D0, D1 = (float(x) for x in sys.argv[2:5])
    H = [float(x) for x in sys.argv[5:9]] if len(sys.argv) >= 9 else [1.0, 1.0, 1.08, 1.08]
    print("proposed:", A8, D0, D1, H)
    show(A8, D0, D1, H)
