"""Everything between split_grad (end of the cross-modal backward) and the next step's concat (rocprofv3 db)."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, stream_id from kernels order by start"))
def short(nm):
    nm = nm.replace("void ", "").replace("(anonymous namespace)::", "")
    if nm.startswith("_ZN12_GLOBAL__N_1"):
        nm = nm[len("_ZN12_GLOBAL__N_1"):].lstrip("0123456789")
    return nm.split("(")[0][:40]
sp = [i for i, r in enumerate(rows) if "split_grad" in r[0]]
cc = [i for i, r in enumerate(rows) if "concat_seq" in r[0]]
a = sp[-2]
b = [i for i in cc if i > a][0]
t0 = rows[a][1]
for r in rows[a - 6:b + 1]:
    print("s%-2d %8.1f -> %8.1f (%6.1f us)  %s" % (r[3], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, short(r[0])))
