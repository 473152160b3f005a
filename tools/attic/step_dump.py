"""Dump ONE train step of a rocprofv3 --kernel-trace database as TSV (start_us, dur_us, stream, kernel) so that the
timeline can be analysed off the GPU box.  python tools/step_dump.py <dir-with-db> <out.tsv>"""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, stream_id from kernels order by start"))


def short(nm):
    nm = nm.replace("void ", "").replace("(anonymous namespace)::", "")
    if nm.startswith("_ZN12_GLOBAL__N_1"):
        nm = nm[len("_ZN12_GLOBAL__N_1"):].lstrip("0123456789")
    return nm.split("(")[0][:60]


idx = [i for i, r in enumerate(rows) if "mse_loss" in r[0]]
# a step = from the first kernel after the previous step's last adam launch up to this step's last adam launch; use the
# loss kernel as the anchor: window = [loss of step k-1 .. loss of step k) shifted to start at the first pad_cast before it
a, b = idx[-3], idx[-2]
pc = [i for i, r in enumerate(rows) if "pad_cast" in r[0] and i < a]
lo = a
for i in reversed(pc):  # the two input pad_casts of this step's forward start
    if a - i < 200:
        lo = i
hi_candidates = [i for i, r in enumerate(rows) if "pad_cast" in r[0] and a < i < b]
hi = hi_candidates[0] if hi_candidates else b
t0 = rows[lo][1]
with open(sys.argv[2], "w") as f:
    for r in rows[lo:hi]:
        f.write("%.2f\t%.2f\t%d\t%s\n" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], short(r[0])))
print("step: %d kernels, %.1f us" % (hi - lo, (rows[hi - 1][2] - t0) / 1e3))
