"""Last 1.5 ms before the Adam kernel of a train step: which stream runs what (rocprofv3 --kernel-trace db)."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, stream_id from kernels order by start"))
ad = [i for i, r in enumerate(rows) if "adam_fused" in r[0]][-2]
t_ad = rows[ad][1]
def short(nm):
    nm = nm.replace("void ", "").replace("(anonymous namespace)::", "")
    if nm.startswith("_ZN12_GLOBAL__N_1"):
        nm = nm[len("_ZN12_GLOBAL__N_1"):].lstrip("0123456789")
    return nm.split("(")[0][:30]
for r in rows[:ad]:
    if r[1] > t_ad - float(sys.argv[2] if len(sys.argv) > 2 else 1500) * 1e3:
        print("s%d  %8.1f -> %8.1f  (%6.1f us)  %s" % (r[3], (r[1] - t_ad) / 1e3, (r[2] - t_ad) / 1e3, (r[2] - r[1]) / 1e3, short(r[0])))
