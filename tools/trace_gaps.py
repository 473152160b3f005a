"""Analyse a rocprofv3 --kernel-trace database of bench.py: for the LAST full train step print, per queue, the kernel
busy time, the idle gaps between consecutive kernels, and the chip-level union (time with >= 1 / >= 2 kernels running).
  python tools/trace_gaps.py <results.db> [--list QUEUE]"""
import sqlite3
import sys


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    n = n.replace("_ZN12_GLOBAL__N_1", "")
    return n[:58]


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = list(c.execute("select name, queue_id, start, end from kernels order by start"))
    adam = [i for i, r in enumerate(rows) if "adam_fused" in r[0]]
    # a step ends with the last adam launch of a run of adam launches; take the last two step boundaries
    ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] - i > 40]
    if len(ends) < 3:
        print("not enough steps in trace"); return
    lo, hi = ends[-3] + 1, ends[-2] + 1   # the step before the last (the last may carry profiling extras)
    step = rows[lo:hi]
    t0, t1 = min(r[2] for r in step), max(r[3] for r in step)
    print("step: %d kernels, wall %.3f ms" % (len(step), (t1 - t0) / 1e6))
    queues = sorted(set(r[1] for r in step))
    for q in queues:
        ks = [r for r in step if r[1] == q]
        busy = sum(r[3] - r[2] for r in ks)
        gaps = [ks[i + 1][2] - ks[i][3] for i in range(len(ks) - 1)]
        pos = [g for g in gaps if g > 0]
        small = [g for g in pos if g < 20000]
        print("queue %d: %4d kernels, busy %.3f ms, span %.3f ms, idle gaps: %d (<20us: %d, sum %.3f ms, median %.2f us); "
              "overlapping starts %d" % (q, len(ks), busy / 1e6, (ks[-1][3] - ks[0][2]) / 1e6, len(pos), len(small),
                                         sum(small) / 1e6, (sorted(small)[len(small) // 2] / 1e3 if small else 0),
                                         sum(1 for g in gaps if g <= 0)))
    # union coverage
    ev = []
    for r in step:
        ev.append((r[2], 1)); ev.append((r[3], -1))
    ev.sort()
    cur, last, cov = 0, t0, {}
    for t, d in ev:
        cov[cur] = cov.get(cur, 0) + (t - last)
        cur += d; last = t
    tot = t1 - t0
    print("time with N kernels running: " + ", ".join("%d: %.1f%%" % (k, 100.0 * v / tot) for k, v in sorted(cov.items())))
    if "--list" in sys.argv:
        q = int(sys.argv[sys.argv.index("--list") + 1])
        prev = None
        for r in step:
            if r[1] != q: continue
            print("%9.2f us  dur %7.2f  gap %6.2f  %s" % ((r[2] - t0) / 1e3, (r[3] - r[2]) / 1e3,
                                                        ((r[2] - prev) / 1e3 if prev else 0), short(r[0])))
            prev = r[3]


if __name__ == "__main__":
    main()
