"""Per-kernel PMC table over whole train steps: merge the rocpd databases of several `rocprofv3 --kernel-trace
--pmc <set> -- python bench.py ...` passes (one directory per pass) into one table.

  python tools/step_pmc.py OUT.txt DIR_PASS1 DIR_PASS2 [...]

Counter collection serialises the dispatches, so cycles (GRBM_GUI_ACTIVE / 8 XCDs) is each kernel's stand-alone
duration in shader clocks, not its contention-stretched in-step duration.  HBM bytes follow
MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB, collected in separate passes, and the
gfx950 FETCH_SIZE is doubled for wide coalesced streams."""
import os as _os; _os.environ.setdefault("FACT_DEBUG_ABI", "1")  # these tools drive the test / bench surface (mint_amd/_lib.py)
import collections
import glob
import sqlite3
import sys


def load(d, agg, calls):
    for db in glob.glob(d + "/**/*.db", recursive=True):
        c = sqlite3.connect(db)
        try:
            cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        except Exception:
            continue
        if "kernel_name" not in cols:
            continue
        ki, ni, vi = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
        seen = collections.defaultdict(lambda: collections.defaultdict(int))
        for r in c.execute("select * from counters_collection"):
            agg[r[ki]][r[ni]][0] += r[vi]
            agg[r[ki]][r[ni]][1] += 1
            seen[r[ki]][r[ni]] += 1
        for k, cs in seen.items():
            calls[k] = max(calls[k], max(cs.values()))


def main():
    argv = list(sys.argv)
    traffic_json = None
    if "--traffic-json" in argv:  # also write the dominant wgrad kernel's measured HBM bytes per launch (bench.py reads it)
        i = argv.index("--traffic-json")
        traffic_json = argv[i + 1]
        del argv[i:i + 2]
    sys.argv = argv
    out = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    calls = collections.defaultdict(int)
    for d in sys.argv[2:]:
        load(d, agg, calls)

    def mean(k, n):
        s, c = agg[k].get(n, (0.0, 0))
        return s / c if c else None

    rows = []
    for k in agg:
        gui = mean(k, "GRBM_GUI_ACTIVE")
        cyc = gui / 8.0 if gui else None
        busy, wave = mean(k, "SQ_VALU_MFMA_BUSY_CYCLES"), mean(k, "SQ_WAVE_CYCLES")
        wait, conf = mean(k, "SQ_WAIT_INST_ANY"), mean(k, "SQ_LDS_BANK_CONFLICT")
        fetch, write = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
        rows.append(dict(
            name=k, calls=calls[k], cycles=cyc,
            mfma=(100.0 * busy / (1024.0 * cyc)) if busy is not None and cyc else None,
            conflict=(100.0 * conf / wave) if conf is not None and wave else None,
            wait=(100.0 * wait / wave) if wait is not None and wave else None,
            rd_mb=(2.0 * fetch * 1024 / 1e6) if fetch is not None else None,
            wr_mb=(write * 1024 / 1e6) if write is not None else None))
    rows.sort(key=lambda r: -((r["cycles"] or 0) * r["calls"]))
    f = lambda v, w, p: ("%*.*f" % (w, p, v)) if v is not None else " " * (w - 1) + "-"
    lines = ["%-66s %6s %10s %7s %9s %7s %9s %9s" % ("kernel", "calls", "cycles", "mfma%", "conflict%", "wait%",
                                                   "read_MB", "write_MB")]
    def short(n):
        for a in ("void (anonymous namespace)::", "(anonymous namespace)::", "_ZN12_GLOBAL__N_1"):
            n = n.replace(a, "")
        return n.replace("BigCfg", "Cfg").replace(", ", ",")
    for r in rows[:48]:
        lines.append("%-66s %6d %s %s %s %s %s %s" % (short(r["name"])[:66], r["calls"], f(r["cycles"], 10, 0),
                                                     f(r["mfma"], 7, 1), f(r["conflict"], 9, 2), f(r["wait"], 7, 1),
                                                     f(r["rd_mb"], 9, 1), f(r["wr_mb"], 9, 1)))
    text = "\n".join(lines)
    print(text)
    open(out, "a").write(text + "\n")
    if traffic_json:
        import json
        import os
        tn = [r for r in rows if "big_tn_kernel" in r["name"] and r["rd_mb"] is not None and r["wr_mb"] is not None]
        if tn:
            r = tn[0]
            # a cross-modal layer's eight bf16 operand matrices read once (144.6 MB at 5760 tokens) + its four fp32 weight
            # gradients written once (29.9 MB; grad_overwrite: plain stores, no read-modify-write), two launches per layer
            algo = (144.6e6 + 29.9e6) / 2
            json.dump({"class": "wgrad_group", "kernel": short(r["name"]),
                       "read_MB_per_launch": round(r["rd_mb"], 1), "write_MB_per_launch": round(r["wr_mb"], 1),
                       "traffic_bytes": int((r["rd_mb"] + r["wr_mb"]) * 1e6), "algorithmic_bytes_per_launch": int(algo),
                       "launches_averaged": r["calls"],
                       "source": "%s (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py, same run "
                                 "as that table; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 wide-read correction; mean over "
                                 "every launch of the kernel, encoder and supervised-rows layers included)" % os.path.basename(out),
                       "note": "algorithmic = half of a cross-modal layer's eight bf16 operand matrices read once (144.6 MB at "
                               "5760 tokens) + its four fp32 weight gradients written once (29.9 MB, grad_overwrite); FETCH_SIZE counts every L2 fill, also those the 256 MB Infinity Cache serves: each of the 8 XCD L2s fills its own copy of the operand panels its ~12 tiles of a launch share (5 + 3 panels of 1.8 / 2.9 MB per K pass for a 5 x 2.4 patch, against 8 panels for the whole launch), so read traffic above the read-once figure is the cost of eight private L2s, not of re-reads within one (L2 hit rate of the kernel 69 %: a line is used by 2.4-5 workgroups of an XCD)"},
                      open(traffic_json, "w"), indent=1)


if __name__ == "__main__":
    main()
