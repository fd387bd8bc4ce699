#!/usr/bin/env python
"""Map the bits of a stream CU mask (hipExtStreamCreateWithCUMask) to hardware units on this device: for groups of mask bits,
launch a grid on a stream masked to them and collect where its workgroups ran (XCC_ID, SE, SH, CU from the hardware registers).
mmada_comm_set_partition (csrc/tp_comm.hip) needs the exchange stream's CUs spread evenly over the eight XCDs.

    python tools/cu_mask_probe.py            # prints bit -> (xcd, se, cu) for every bit, and a summary per 8 / 16 / 32-bit group
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mmada_parallel_amd import abi  # noqa: E402


def where(lib, bits, ncu, n_blocks=4096):
    words = (ncu + 31) // 32
    mask = (C.c_uint32 * words)()
    for b in bits:
        mask[b // 32] |= 1 << (b % 32)
    out = (C.c_uint32 * n_blocks)()
    abi.check(lib.mmada_probe_cu_mask(mask, words, out, n_blocks), "probe_cu_mask")
    seen = {}
    for v in out:
        xcc, hw = v & 15, v >> 8
        key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
        seen[key] = seen.get(key, 0) + 1
    return seen


def main():
    torch.cuda.init()
    lib = abi.lib()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    print(f"{ncu} CUs")
    per_bit = {}
    for b in range(ncu):
        s = where(lib, [b], ncu, 512)
        per_bit[b] = sorted(s)
    # An XCD whose share of the mask is all zero is NOT restricted (observed: a one-bit mask pins one XCD to one CU and leaves
    # the other seven whole), so the CU a bit selects is the lone CU of the XCD that shows fewer CUs than the rest.
    def pinned(seen):
        per = {}
        for t in seen:
            per.setdefault(t[0], []).append(t[1:])
        few = [(x, c) for x, c in per.items() if len(c) == 1]
        return (few[0][0],) + few[0][1][0] if len(few) == 1 else None

    sel = {b: pinned(per_bit[b]) for b in per_bit}
    for b in range(ncu):
        print(f"bit {b:3d} -> (xcd, se, sh, cu) {sel[b]}")
    xcd_of = {b: sel[b][0] if sel[b] else None for b in sel}
    for n in (8, 16, 32):
        low = [xcd_of[b] for b in range(n)]
        print(f"low {n} bits cover XCDs {sorted(set(low))} ({[low.count(x) for x in range(8)]} CUs per XCD)")
    stride = [xcd_of[b] for b in range(0, ncu, ncu // 8)]
    print(f"bits 0, {ncu // 8}, {2 * ncu // 8}, ...: XCDs {stride}")


if __name__ == "__main__":
    main()
