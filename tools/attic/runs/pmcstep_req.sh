#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/pmcstepreq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
d=$O/tmp
(cd $R && timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $d -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1)
python - <<PY
import glob, sqlite3, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for db in glob.glob("$d/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    ki, ni, vi = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
    for r in c.execute("select * from counters_collection"):
        agg[r[ki]][r[ni]].append(r[vi])
def short(n):
    for a in ("void (anonymous namespace)::", "(anonymous namespace)::", "_ZN12_GLOBAL__N_1"):
        n = n.replace(a, "")
    return n.replace("BigCfg", "Cfg").replace(", ", ",")[:60]
print("%-60s %6s %10s %10s %8s" % ("kernel", "calls", "rd_req", "wr_req", "L2hit%"))
rows = []
for k, cs in agg.items():
    m = lambda n: sum(cs[n]) / len(cs[n]) if cs.get(n) else 0
    rows.append((m("TCP_TCC_READ_REQ_sum") * len(cs["TCP_TCC_READ_REQ_sum"]), short(k), len(cs["TCP_TCC_READ_REQ_sum"]), m("TCP_TCC_READ_REQ_sum"), m("TCP_TCC_WRITE_REQ_sum"), 100.0 * m("TCC_HIT_sum") / max(1.0, m("TCC_HIT_sum") + m("TCC_MISS_sum"))))
for r in sorted(rows, reverse=True)[:22]:
    print("%-60s %6d %10.0f %10.0f %8.1f" % r[1:])
PY
rm -rf $d
