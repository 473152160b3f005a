"""Tail of a train step (encoder backward chains + optimizer) from a rocprofv3 kernel-trace database."""
import glob, sqlite3, sys
db = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[-1]
c = sqlite3.connect(db)
rows = list(c.execute("select name, queue_id, start, end, grid_x, workgroup_x, grid_y from kernels order by start"))
adam = [i for i, r in enumerate(rows) if "adam_fused" in r[0]]
ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] - i > 40]
lo, hi = ends[-3] + 1, ends[-2] + 1
step = rows[lo:hi]
t0 = min(r[2] for r in step)
sg = [r for r in step if "split_grad" in r[0]][0]
print("wall %.1f us; split_grad ends at %.1f" % ((max(r[3] for r in step) - t0) / 1e3, (sg[3] - t0) / 1e3))
for q in sorted(set(r[1] for r in step)):
    ks = [r for r in step if r[1] == q and r[2] >= sg[3]]
    if ks:
        print("queue %d after split: first start %.1f, last end %.1f, busy %.1f us, %d kernels" % (
            q, (ks[0][2] - t0) / 1e3, (ks[-1][3] - t0) / 1e3, sum(r[3] - r[2] for r in ks) / 1e3, len(ks)))
if len(sys.argv) > 2:
    q3 = [r for r in step if r[1] == 3 and r[2] >= sg[3]]
    ts = (q3[0][2] - t0) / 1e3
    def short(n):
        for a in ("void (anonymous namespace)::", "(anonymous namespace)::", "_ZN12_GLOBAL__N_1"):
            n = n.replace(a, "")
        return n[:48]
    for r in step:
        t = (r[2] - t0) / 1e3
        if (sg[2] - t0) / 1e3 - 5 < t < ts + 80:
            print("q%d %9.2f -> %9.2f (%6.2f) wgs %5d  %s" % (r[1], t, (r[3] - t0) / 1e3, (r[3] - r[2]) / 1e3,
                                                             (r[4] // max(1, r[5])) * r[6], short(r[0])))
