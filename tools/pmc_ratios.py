"""Per-kernel wave-cycle breakdown from one rocprofv3 --pmc pass holding SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (+ optional SQ_WAIT_INST_LDS): python tools/pmc_ratios.py DIR"""
import collections
import glob
import sqlite3
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    ki, ni, vi = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
    for r in c.execute("select * from counters_collection"):
        agg[r[ki]][r[ni]].append(r[vi])


def short(n):
    for a in ("void (anonymous namespace)::", "(anonymous namespace)::"):
        n = n.replace(a, "")
    return n.replace("BigCfg", "Cfg").replace(", ", ",")[:58]


print("%-58s %5s %9s %7s %7s %7s %7s %7s" % ("kernel", "n", "cycles", "mfma%", "parked%", "istall%", "active%", "ldsst%"))
for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    m = lambda n: (sum(cs[n]) / len(cs[n])) if n in cs and cs[n] else None
    wave, gui = m("SQ_WAVE_CYCLES"), m("GRBM_GUI_ACTIVE")
    if not wave or not gui:
        continue
    cyc = gui / 8.0
    pct = lambda n: ("%7.1f" % (100.0 * m(n) / wave)) if m(n) is not None else "      -"
    busy = m("SQ_VALU_MFMA_BUSY_CYCLES")
    print("%-58s %5d %9.0f %7.1f %s %s %s %s" % (short(k), len(cs["SQ_WAVE_CYCLES"]), cyc, 100.0 * busy / (1024.0 * cyc) if busy else 0,
                                               pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"), pct("SQ_ACTIVE_INST_ANY"), pct("SQ_WAIT_INST_LDS")))
