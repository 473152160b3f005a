"""Summarise rocprofv3 --pmc output (rocpd sqlite): per-kernel mean of each counter."""
import glob, sqlite3, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for db in glob.glob(d + "/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select * from counters_collection limit 1").fetchall()
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    except Exception as e:
        print("no counters_collection", e); continue
    ki = cols.index("kernel_name") if "kernel_name" in cols else None
    ni = cols.index("counter_name"); vi = cols.index("value")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in c.execute("select * from counters_collection"):
        k = r[ki] if ki is not None else "?"
        if pat and pat not in k: continue
        agg[k[:110]][r[ni]].append(r[vi])
    for k, cs in agg.items():
        print(k)
        for n, v in sorted(cs.items()):
            print("   %-32s n=%d mean=%.4g" % (n, len(v), sum(v) / len(v)))
