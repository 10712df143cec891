#!/usr/bin/env python3
"""Compact instruction-class trace of one kernel's loops from hipcc's -save-temps assembly.

usage: tools/asm_trace.py file.s 'kernel-name-regex' [--all]
Prints, per basic block that contains MFMAs (or all with --all): one letter per instruction
  M mfma  r ds_read  w ds_write  G global/buffer load  S global/buffer store  D LDS-DMA  v VALU  s SALU
and a line break at every s_waitcnt (shown with its counters) / s_barrier, so that a read that is consumed right after
it was issued ("r r r [lgkmcnt(1)] v") is visible at a glance.  Works on CPU only (no GPU needed)."""
import re, sys

def main():
    path, pat = sys.argv[1], re.compile(sys.argv[2])
    show_all = "--all" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):", l)
        if m and pat.search(m.group(1)):
            start = i
            name = m.group(1)
            break
    if start is None:
        sys.exit("kernel not found")
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    print("#", name, "lines", start, end)
    blocks, cur, label = [], [], "entry"
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\S+):", t)
            if m:
                blocks.append((label, cur)); cur, label = [], m.group(1)
            continue
        cur.append(t.split(";")[0].strip())
    blocks.append((label, cur))
    for label, ins in blocks:
        nm = sum(1 for x in ins if x.startswith("v_mfma"))
        if not ins or (nm == 0 and not show_all):
            continue
        cnt = {}
        out = []
        for x in ins:
            op = x.split()[0]
            if op.startswith("v_mfma"): c = "M"
            elif op.startswith("ds_read") or op.startswith("ds_load"): c = "r"
            elif op.startswith("ds_write") or op.startswith("ds_store"): c = "w"
            elif re.match(r"(buffer|global)_load.*lds", x) or " lds" in x and op.startswith(("buffer_load", "global_load")): c = "D"
            elif op.startswith(("buffer_load", "global_load", "flat_load")): c = "G"
            elif op.startswith(("buffer_store", "global_store", "flat_store")): c = "S"
            elif op == "s_waitcnt": c = "\n  [%s] " % " ".join(x.split()[1:])
            elif op == "s_barrier": c = "\n  ===BARRIER===\n  "
            elif op.startswith("v_"): c = "v"
            elif op.startswith("s_"): c = "s"
            else: c = "?"
            key = c.strip()[:1] if c.strip() else "?"
            cnt[op] = cnt.get(op, 0) + 1
            out.append(c if c.startswith("\n") else c + " ")
        print("== %s: %d instr, %d MFMA" % (label, len(ins), nm))
        print("  " + "".join(out))
        top = sorted(cnt.items(), key=lambda kv: -kv[1])[:14]
        print("  ops: " + ", ".join("%s x%d" % kv for kv in top))

main()
