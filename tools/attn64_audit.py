#!/usr/bin/env python
"""Static audit of attention64.hip's compiled kernel (the accumulator file is owned by hand there, CDNA guide §5.7 item 4):

  * no VGPR spill, no scratch;
  * every v_accvgpr_* / AGPR operand the COMPILER issued (outside ;;#ASMSTART … ;;#ASMEND) stays out of a[0:191];
  * per key-tile loop body: MFMA count and the number of other instructions between consecutive MFMAs.

    python tools/attn64_audit.py [--gaps]         (compiles mmada_parallel_amd/csrc/attention64.hip with -save-temps)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OWNED = 192


def compile_s(src=None):
    src = src or os.path.join(ROOT, "mmada_parallel_amd", "csrc", "attention64.hip")
    d = tempfile.mkdtemp(prefix="attn64_audit_")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", os.path.join(d, "a.o"), "-save-temps"]
    p = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
    if p.returncode:
        raise RuntimeError(p.stderr)
    return open(os.path.join(d, "attention64-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def kernel_text(s, name="attn64_fwd_kernel"):
    i = s.index(name)
    i = s.index(":\n", i)
    return s[i:s.index(".end_amdhsa_kernel", i)]


def audit(s):
    k = kernel_text(s)
    meta = {}
    for key in ("vgpr_spill_count", "private_segment_fixed_size", "vgpr_count", "agpr_count", "sgpr_count"):
        m = re.search(r"\.%s:\s+(\d+)" % key, s[s.index("amdhsa.kernels"):])
        meta[key] = int(m.group(1)) if m else None
    problems = []
    if meta["vgpr_spill_count"]:
        problems.append(f"vgpr_spill_count = {meta['vgpr_spill_count']}")
    if meta["private_segment_fixed_size"]:
        problems.append(f"scratch = {meta['private_segment_fixed_size']} bytes")
    inasm = False
    stray = []
    for ln in k.split("\n"):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        if inasm or not t or t[0] in ";.":
            continue
        for m in re.finditer(r"\ba\[?(\d+)(?::(\d+))?\]?", t):
            lo = int(m.group(1))
            hi = int(m.group(2)) if m.group(2) else lo
            if lo < OWNED:
                stray.append(t)
    if stray:
        problems.append(f"{len(stray)} compiler instructions touch a[0:{OWNED - 1}], e.g. {stray[0]!r}")
    return meta, problems, k


def vregs(operand):
    """Set of VGPR numbers an operand string names (v12 or v[12:15]); AGPRs, SGPRs, immediates -> empty."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", operand)
    return {int(m.group(1))} if m else set()


def mfma_operand_hazards(k):
    """gfx950: a VGPR written by a VALU instruction may be read by an MFMA only after 2 wait states; hipcc pads that for
    MFMAs it emits, not for the asm statements of attention64.hip.  Walk the kernel in program order; for every MFMA that is
    not preceded by its own `s_nop 1`, look at the instructions issued in the two wait states before it (an `s_nop N` is
    N + 1 states; a label inside the window makes the check conservative: the fall-through predecessor is examined and a
    jump into the window is reported as unknown) and report any VALU / accvgpr write to one of its VGPR source operands."""
    instrs = []
    for ln in k.split("\n"):
        t = ln.split(";")[0].strip()
        if not t or t[0] == ".":
            continue
        instrs.append(t)
    bad = []
    for n, t in enumerate(instrs):
        if not t.startswith("v_mfma"):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        dst, srcs = ops[0], ops[1:]
        need = set()
        for o in srcs:
            need |= vregs(o)
        if not need:
            continue
        states, j = 0, n - 1
        while j >= 0 and states < 2:
            p = instrs[j]
            if p.endswith(":"):           # a label: control may arrive here from elsewhere
                bad.append((n, t, "label inside the hazard window: " + p))
                break
            op = p.split()[0]
            if op == "s_nop":
                states += int(p.split()[1]) + 1
            else:
                if op.startswith("v_") and not op.startswith("v_mfma") and not op.startswith("v_cmp"):
                    w = [x.strip() for x in p.split(None, 1)[1].split(",")][0] if " " in p else ""
                    if vregs(w) & need:
                        bad.append((n, t, "written by: " + p))
                        break
                states += 1
            j -= 1
    return bad


def gaps(k):
    """Instruction counts between consecutive MFMAs for each barrier-to-barrier region."""
    regions, cur = [], []
    for ln in k.split("\n"):
        t = ln.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_barrier":
            regions.append(cur)
            cur = []
        else:
            cur.append(op)
    regions.append(cur)
    out = []
    for r in regions:
        n_mfma = sum(1 for o in r if o.startswith("v_mfma"))
        if n_mfma < 8:
            continue
        g, c = [], 0
        for o in r:
            if o.startswith("v_mfma"):
                g.append(c)
                c = 0
            elif o not in ("s_nop",):
                c += 1
        kinds = {}
        for o in r:
            kinds[o] = kinds.get(o, 0) + 1
        out.append((n_mfma, len(r), g, kinds))
    return out


if __name__ == "__main__":
    s = compile_s()
    meta, problems, k = audit(s)
    print(meta)
    if "--gaps" in sys.argv:
        for n_mfma, n, g, kinds in gaps(k):
            if n > 3000:
                continue
            top = sorted(kinds.items(), key=lambda x: -x[1])[:14]
            print(f"region: {n_mfma} MFMAs, {n} instructions, fillers/gap max {max(g)} mean {sum(g) / len(g):.1f}: {g}")
            print("   ", top)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from isa_check import check_m0
    problems += check_m0("attn64_fwd_kernel", k.split("\n"))
    hz = mfma_operand_hazards(k)
    if hz:
        problems.append(f"{len(hz)} MFMA operand hazards, e.g. {hz[0]}")
    if problems:
        print("PROBLEMS:", *problems, sep="\n  ")
        sys.exit(1)
    print("attn64 audit OK")
