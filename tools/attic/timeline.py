"""Timeline analysis of a rocprofv3 --kernel-trace run (rocpd sqlite): per-stream busy time,
union busy time, idle gaps, for the LAST train step in the trace."""
import glob, sqlite3, sys, collections

d = sys.argv[1]
db = glob.glob(d + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = list(c.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
# find step boundaries: mse_loss_kernel appears once per step
idx = [i for i, r in enumerate(rows) if "mse_loss" in r[0]]
print("steps in trace:", len(idx))
if len(idx) < 3:
    sys.exit(0)
# step = from first kernel after the previous step's last adam to ... approximate with mse-to-mse window
a, b = idx[-3], idx[-2]
win = rows[a:b]
t0, t1 = win[0][1], win[-1][2]
print("window (mse->mse) wall: %.1f us, kernels %d" % ((t1 - t0) / 1e3, len(win)))
by_stream = collections.defaultdict(list)
for r in win:
    by_stream[r[3]].append(r)
for s, rs in by_stream.items():
    busy = sum(r[2] - r[1] for r in rs)
    print("stream %s: %d kernels, busy %.1f us" % (s, len(rs), busy / 1e3))
# union busy
ev = sorted((r[1], r[2]) for r in win)
busy = 0; cs, ce = ev[0]
for s, e in ev[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("union busy %.1f us (idle %.1f us)" % (busy / 1e3, (t1 - t0 - busy) / 1e3))
# main-stream gaps: the stream with most kernels
main = max(by_stream.items(), key=lambda kv: len(kv[1]))[1]
gaps = collections.defaultdict(lambda: [0, 0.0])
tot_gap = 0
for p, n in zip(main, main[1:]):
    g = (n[1] - p[2]) / 1e3
    def short(nm):
        nm = nm.replace("void ", "").replace("(anonymous namespace)::", "")
        if nm.startswith("_ZN12_GLOBAL__N_1"):
            nm = nm[len("_ZN12_GLOBAL__N_1"):].lstrip("0123456789")
        return nm.split("(")[0][:34]
    key = short(p[0]) + " -> " + short(n[0])
    gaps[key][0] += 1; gaps[key][1] += g; tot_gap += g
print("main stream: sum of gaps %.1f us over %d kernels" % (tot_gap, len(main)))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:22]:
    print("  %-72s n=%3d total %6.1f us avg %5.2f" % (k, v[0], v[1], v[1] / v[0]))
