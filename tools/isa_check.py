#!/usr/bin/env python
"""Static checks on the gfx950 ISA hipcc emits for the hand-scheduled kernels (no GPU needed).

    python tools/isa_check.py            # prints a report, exit code 1 on a violation

The 8-phase GEMM (csrc/gemm8.hip) keeps four half-tiles of LDS-DMA in flight behind COUNTED `s_waitcnt vmcnt(N)`; the
counts are only meaningful while nothing but LDS-DMA sits in the vector-memory queue.  hipcc knows nothing about that
contract: a register spill (scratch_store / scratch_load) or a hoisted global access inside the pipelined region would
enter the same queue.  Rule checked per kernel, in program order: after any foreign vector-memory instruction, the next
hand-written wait must be `vmcnt(0)` (which is exact whatever is queued); a counted wait with N > 0 there is an error.
Also checked: no static LDS (the kernel forms LDS addresses from integers, i.e. assumes its dynamic segment starts at 0),
and every kernel really uses the LDS-DMA saddr form.

Epilogues of the 8-phase GEMM (STORE / RESID / SwiGLU builds): 16-byte stores only, fed by v_permlane16_swap.  Attention
(check_attention): pipelined fragment reads, no spills, no packed fp32 VALU, no compiler-counted vmcnt wait inside a tile loop.

The tensor-parallel pull transport (csrc/tp_comm.hip) must read remote memory with ONE 16-byte system-scope load per
16 bytes (`global_load_dwordx4 ... sc0 sc1`), never as two 8-byte halves (each would use half of every 64-B fabric
request and the second would re-request the same lines).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mmada_parallel_amd", "csrc")
FOREIGN = re.compile(r"^\s*(scratch_(load|store)|global_(load|store|atomic)(?!_lds)|buffer_(load|store|atomic)|flat_(load|store|atomic))")


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def device_asm(src):
    """gfx950 assembly text of one translation unit, compiled with the flags of the shipped object (build.FLAGS + UNIT_FLAGS)."""
    sys.path.insert(0, ROOT)
    from mmada_parallel_amd import build as _build
    flags = [f for f in _build.FLAGS if f != "-fPIC"] + ["-fPIC"] + _build.UNIT_FLAGS.get(src, [])
    out = subprocess.run([hipcc(), *flags, "--cuda-device-only", "-S", "-I", CSRC, os.path.join(CSRC, src), "-o", "-"],
                         capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError(out.stderr)
    return out.stdout


def kernels(asm):
    """name -> list of instruction lines, and name -> metadata dict (from the .amdhsa_ directives)."""
    body, meta, cur = {}, {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur = m.group(1)
            body[cur] = []
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is not None:
            body[cur].append(line)
        m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            meta_cur = m.group(1)
            meta[meta_cur] = {}
            continue
        m = re.match(r"^\s*\.amdhsa_(\w+)\s+(\S+)", line)
        if m and meta:
            meta[list(meta)[-1]][m.group(1)] = m.group(2)
    return body, meta


def check_m0(name, lines):
    """The LDS-DMA requests of the 8-phase GEMM and of the attention kernels are asm statements that set M0 (the LDS
    destination) themselves.  hipcc treats m0 as a reserved register and ignores it in a clobber list, so the contract is
    checked on the compiled code: every LDS-DMA is directly preceded by the instruction that writes m0, and nothing else
    in the kernel reads m0."""
    errors = []
    ins = [ln.split(";")[0].strip() for ln in lines]
    ins = [t for t in ins if t and not t.startswith(".") and not t.endswith(":")]
    for i, t in enumerate(ins):
        if t.startswith("global_load_lds"):
            prev = ins[i - 1] if i else ""
            if not re.match(r"s_(mov_b32|add_i32|add_u32)\s+m0,", prev):
                errors.append(f"{name}: LDS-DMA not directly behind its m0 write: {prev!r} ; {t!r}")
                break
        elif re.search(r"\bm0\b", t):
            ops = t.split(None, 1)[1] if " " in t else ""
            if not re.match(r"m0\s*,", ops) or re.search(r",.*\bm0\b", ops):
                errors.append(f"{name}: m0 is read by {t!r}")
                break
    return errors


def check_gemm8(asm=None):
    asm = asm or device_asm("gemm8.hip")
    body, meta = kernels(asm)
    report, errors = [], []
    for name, lines in body.items():
        if "gemm8_kernel" not in name:
            continue
        # Control-flow graph over basic blocks, then a forward may-analysis of one bit: "a foreign vector-memory op may be
        # in the queue" (set by a foreign op, cleared by any vmcnt(0) wait).  A hand-written counted wait reached with
        # the bit set is an error.
        blocks, cur, labels = [], [], {}
        for ln in lines:
            s = ln.strip()
            m = re.match(r"^(\.LBB\w+):", ln)
            if m:
                if cur:
                    blocks.append(cur)
                cur = []
                labels[m.group(1)] = len(blocks)
                continue
            if not s or s.startswith(".") or (s.startswith(";") and not s.startswith(";;#ASM")):
                continue
            cur.append(s)
            if re.match(r"s_c?branch|s_endpgm|s_setpc", s):
                blocks.append(cur)
                cur = []
        if cur:
            blocks.append(cur)
        succ = []
        for i, b in enumerate(blocks):
            last = b[-1] if b else ""
            out = []
            m = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", last)
            if m and m.group(1) in labels:
                out.append(labels[m.group(1)])
            if not re.match(r"s_branch|s_endpgm|s_setpc", last) and i + 1 < len(blocks):
                out.append(i + 1)
            succ.append(out)
        n_dma = n_saddr = n_wait = n_foreign = 0
        for b in blocks:
            for s in b:
                if s.startswith("global_load_lds_dwordx4"):
                    n_dma += 1
                    n_saddr += bool(re.search(r"global_load_lds_dwordx4\s+v\d+,\s*s\[", s))
                n_foreign += bool(FOREIGN.match(s))

        def transfer(b, dirty, report_to=None):
            in_asm = False
            for s in b:
                if s.startswith(";;#ASMSTART"):
                    in_asm = True
                elif s.startswith(";;#ASMEND"):
                    in_asm = False
                elif FOREIGN.match(s):
                    dirty = s.split(";")[0].strip()
                else:
                    m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", s)
                    if m and int(m.group(1)) == 0:
                        dirty = None
                    elif m and in_asm and dirty and report_to is not None:
                        report_to.append(f"{name}: counted wait vmcnt({m.group(1)}) may follow foreign vector-memory op '{dirty}'")
            return dirty

        state = [None] * len(blocks)   # dirty-at-entry per block (None = clean / not reached dirty)
        work = [0]
        seen = {0}
        while work:
            i = work.pop()
            out = transfer(blocks[i], state[i])
            for j in succ[i]:
                if (out and not state[j]) or j not in seen:
                    if out and not state[j]:
                        state[j] = out
                    seen.add(j)
                    work.append(j)
        found = []
        for i in sorted(seen):
            transfer(blocks[i], state[i], found)
            n_wait += sum(1 for s in blocks[i] if re.match(r"s_waitcnt vmcnt\(", s))
        errors += sorted(set(found))
        # The steady-state loops (a back edge; the body is two K-tiles): the literal immediate of every counted wait must be
        # this wave's LDS-DMA instructions of FOUR half-tiles = the loop's own LDS-DMA count per K-tile — the quantity
        # `four` of the schedule model tests/test_host_logic.py::_gemm8_program simulates — and a K-tile has three (first
        # schedule) or four (balanced) such waits.  An edit of a wait_vm<> count in csrc/gemm8.hip fails here, not only in
        # a timing-dependent race on the GPU.
        n_loops = 0
        for i, outs in enumerate(succ):
            for j in outs:
                if j != i:
                    continue   # the steady-state loop is one basic block branching to itself
                loop = blocks[i]
                dma = sum(1 for x in loop if x.startswith("global_load_lds_dwordx4"))
                waits = [int(m.group(1)) for x in loop for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)", x)] if m]
                if dma < 8 or not waits:
                    continue   # not a main loop (epilogue / planner loops)
                n_loops += 1
                if dma % 2:
                    errors.append(f"{name}: a main loop issues {dma} LDS-DMA instructions (two K-tiles expected)")
                    continue
                four = dma // 2
                if any(w != four for w in waits):
                    errors.append(f"{name}: main-loop waits {sorted(set(waits))} != {four} (this wave's LDS-DMA of four half-tiles)")
                if len(waits) not in (6, 8):
                    errors.append(f"{name}: {len(waits)} counted waits in a two-K-tile loop (6 or 8 expected)")
        if n_loops < 1:
            errors.append(f"{name}: no steady-state loop found")
        lds = int(meta.get(name, {}).get("group_segment_fixed_size", "0"))
        if lds != 0:
            errors.append(f"{name}: static LDS of {lds} bytes (the kernel assumes its dynamic LDS segment starts at 0)")
        # no scratch at all (round 5): every shipped instantiation keeps its accumulators in registers from the first MFMA to the
        # last store.  (Round 4 shipped 20 / 52 bytes of scratch in the 320x256 SwiGLU / QKV builds: an accumulator spilled in
        # the tail K-tiles, where hipcc picked the three-address MFMA form; csrc/gemm8.hip run() keeps them in a one-trip loop.)
        priv = int(meta.get(name, {}).get("private_segment_fixed_size", "0"))
        n_scratch = sum(1 for b in blocks for x in b if x.startswith("scratch_"))
        if priv != 0 or n_scratch:
            errors.append(f"{name}: private segment of {priv} bytes, {n_scratch} scratch instructions (a register spill)")
        if n_dma == 0 or n_saddr != n_dma:
            errors.append(f"{name}: {n_saddr} of {n_dma} LDS-DMA loads use the scalar-base form")
        # epilogues of the STORE / RESID / SwiGLU builds (template argument EPI = 0, 1, 2: every wave has the transposed
        # accumulator): all output goes out as 16-byte stores after a half-row lane exchange — no 2-, 4- or 8-byte store left
        m_epi = re.search(r"gemm8_kernelILi(\d)E", name)
        if m_epi and int(m_epi.group(1)) in (0, 1, 2):
            flat = [x for b in blocks for x in b]
            narrow = sum(1 for x in flat if re.match(r"global_store_(short|dword|dwordx2)\b", x))
            wide = sum(1 for x in flat if x.startswith("global_store_dwordx4"))
            swaps = sum(1 for x in flat if x.startswith("v_permlane16_swap"))
            if narrow or not wide or not swaps:
                errors.append(f"{name}: epilogue stores: {wide} x 16 B, {narrow} narrower, {swaps} v_permlane16_swap "
                              f"(expected 16-byte stores only, fed by the half-row exchange)")
        report.append((name, n_dma, n_wait, n_foreign))
    if not report:
        errors.append("no gemm8 kernel found")
    for name, lines in body.items():
        if "gemm8_kernel" in name:
            errors += check_m0(name, lines)
    return report, errors


def check_attention(asm=None):
    """The attention kernel (csrc/attention.hip, attn16_kernel): its matrix blocks must really be software-pipelined (counted
    lgkmcnt waits, not a full drain in front of every MFMA group), it forms LDS addresses from integers (no static LDS), must not
    spill, keeps its fp32 row sums / rescales as single instructions (no v_pk_*_f32: built with -fno-slp-vectorize) and — the
    hazard found in round 6 — carries no compiler-made `s_waitcnt vmcnt(N > 0)` inside a loop: hipcc counts only the loads it
    knows, the LDS-DMA requests of the next tile are in the same queue, so such a wait (the Q fragment loads, if their first
    use sinks into the loop) would stall every tile until the DMA lands."""
    asm = asm or device_asm("attention.hip")
    body, meta = kernels(asm)
    report, errors = [], []
    for name, lines in body.items():
        if "attn16_kernel" not in name:
            continue
        lds = int(meta.get(name, {}).get("group_segment_fixed_size", "0"))
        if lds != 0:
            errors.append(f"{name}: static LDS of {lds} bytes")
        n_scratch = sum(1 for ln in lines if re.match(r"\s*scratch_", ln))
        if n_scratch:
            errors.append(f"{name}: {n_scratch} scratch accesses (register spills)")
        counted = sum(1 for ln in lines if re.match(r"\s*s_waitcnt lgkmcnt\([1-9]\d*\)", ln))
        n_dma = sum(1 for ln in lines if ln.strip().startswith("global_load_lds_dwordx4"))
        n_saddr = sum(1 for ln in lines if re.search(r"global_load_lds_dwordx4\s+v\d+,\s*s\[", ln))
        if n_dma == 0 or n_saddr != n_dma:
            errors.append(f"{name}: {n_saddr} of {n_dma} LDS-DMA loads use the scalar-base form")
        if counted < 48:
            errors.append(f"{name}: only {counted} counted lgkmcnt waits: the fragment prefetch was serialised by the compiler")
        n_pk = sum(1 for ln in lines if re.match(r"\s*v_pk_(add|mul|fma)_f32", ln))
        if n_pk:
            errors.append(f"{name}: {n_pk} v_pk_*_f32 instructions (the unit must be built with -fno-slp-vectorize)")
        n_canon = sum(1 for ln in lines if re.match(r"\s*v_max_f32_e32 (v\d+), (v\d+), \2\b", ln))
        if n_canon:
            errors.append(f"{name}: {n_canon} canonicalising v_max x,x (the unit must be built with -fno-honor-nans)")
        in_loop = False
        for ln in lines:
            if re.match(r"^\.LBB\d+_\d+:", ln):
                in_loop = "Loop Header" in ln or "in Loop:" in ln
            m = re.match(r"\s*s_waitcnt .*vmcnt\((\d+)\)", ln)
            if in_loop and m and int(m.group(1)) > 0:
                errors.append(f"{name}: compiler-counted `{ln.strip()}` inside a tile loop (LDS-DMA shares that queue)")
        errors += check_m0(name, lines)
        report.append((name, n_dma, sum(1 for ln in lines if "v_mfma" in ln), counted))
    if not report:
        errors.append("attn16_kernel not found")
    return report, errors


def check_tp_pull(asm=None):
    asm = asm or device_asm("tp_comm.hip")
    body, _ = kernels(asm)
    report, errors = [], []
    for name, lines in body.items():
        if "tp_reduce_norm_kernel" not in name and "tp_gather_kernel" not in name:
            continue
        x4 = sum(1 for ln in lines if re.search(r"(global|buffer)_load_dwordx4 .*sc0 sc1", ln))
        x2 = sum(1 for ln in lines if re.search(r"(global|buffer|flat)_load_(dword|dwordx2|dwordx3) .*sc0 sc1", ln))
        report.append((name, x4, x2))
        if x2:
            errors.append(f"{name}: {x2} narrow system-scope loads (remote pulls must be single 16-byte loads)")
        if x4 == 0:
            errors.append(f"{name}: no 16-byte system-scope load found")
        # a buffer descriptor hipcc cannot prove wave-uniform is wrapped in a v_readfirstlane / s_and_saveexec loop per load
        if any("v_readfirstlane" in ln for ln in lines) and sum("s_and_saveexec" in ln for ln in lines) > 2:
            errors.append(f"{name}: waterfall loops around the buffer loads (descriptor not provably wave-uniform)")
    if not report:
        errors.append("no pull-transport kernel found")
    return report, errors


def main():
    bad = []
    rep, err = check_gemm8()
    for name, n_dma, n_wait, n_foreign in rep:
        print(f"gemm8  {name[:90]:90s} LDS-DMA {n_dma:3d}  counted waits {n_wait:3d}  foreign VMEM after the first DMA {n_foreign}")
    bad += err
    rep, err = check_attention()
    for name, n_dma, n_mfma, counted in rep:
        print(f"attn   {name[:90]:90s} LDS-DMA {n_dma:3d}  MFMA {n_mfma}  counted lgkmcnt waits {counted}")
    bad += err
    rep, err = check_tp_pull()
    for name, x4, x2 in rep:
        print(f"tp     {name[:90]:90s} sc0 sc1 loads: dwordx4 {x4}  dwordx2 {x2}")
    bad += err
    for e in bad:
        print("ERROR:", e)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
