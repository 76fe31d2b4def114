#!/usr/bin/env python3
"""Lists the loops (backward branches) of one kernel in a hipcc -S dump with their instruction mix.

    python tools/asm_loops.py kernels.s swc_inflate_sync_kernel [min_len]
"""
import re
import sys


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % name, l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    insts = []   # (index in body, text)
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        insts.append(s.split(";")[0].strip())
    loops = []
    for k, ins in enumerate(insts):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", ins)
        if m and m.group(1) in labels and labels[m.group(1)] <= k:
            loops.append((labels[m.group(1)], k, m.group(1)))
    print("%d instructions, %d loops" % (len(insts), len(loops)))
    for a, b, lab in sorted(loops):
        seg = insts[a:b + 1]
        if len(seg) < min_len:
            continue
        mix = {}
        for ins in seg:
            op = ins.split()[0]
            cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_nop", "s_barrier"))
                   else "branch" if op.startswith(("s_cbranch", "s_branch")) else "wait" if op.startswith(("s_waitcnt", "s_nop", "s_barrier")) else "lds" if op.startswith("ds_")
                   else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
            mix[cls] = mix.get(cls, 0) + 1
        print("%-12s [%5d..%5d] len %4d  %s" % (lab, a, b, len(seg), " ".join("%s=%d" % kv for kv in sorted(mix.items()))))


if __name__ == "__main__":
    main()
