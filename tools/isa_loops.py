"""Instruction mix of the loops of one kernel in a hipcc -S listing: python tools/isa_loops.py file.s <symbol substring>
For every backward branch (a loop) prints its line span and the count of MFMA / transcendental / other VALU / LDS / VMEM /
SALU / waitcnt / barrier instructions inside it (nested loops are counted in every enclosing span)."""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(key), l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith(("v_cvt",)): return "valu_cvt"
    if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")) or "dpp" in op: return "valu_xlane"
    if op.startswith(("v_accvgpr",)): return "acc_mov"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"


def count(lo, hi):
    c = Counter()
    ops = Counter()
    for l in body[lo:hi]:
        m = re.match(r"^\t([a-z_0-9]+)", l)
        if m and not m.group(1).startswith("."):
            c[cls(m.group(1))] += 1
            ops[m.group(1)] += 1
    return c, ops

tot, _ = count(0, len(body))
print("kernel %s: %d lines, totals %s" % (key, len(body), dict(tot)))
for i, l in enumerate(body):
    m = re.match(r"^\t(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        lo = labels[m.group(2)]
        c, ops = count(lo, i + 1)
        n = sum(c.values())
        if n < 20:
            continue
        print("loop %s lines %d..%d (%d instr): %s" % (m.group(2), lo, i, n, dict(sorted(c.items()))))
        if "-v" in sys.argv:
            print("   top ops:", ops.most_common(25))
