"""How much of each adam_kernel / multi_cast launch overlaps other kernels (rocprofv3 rocpd DB)."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
import bisect
adam = [(n, s, e) for n, s, e in rows if "adam_kernel" in n]
others = [(s, e) for n, s, e in rows if "adam_kernel" not in n and "multi_cast" not in n]
tot = ov = 0
for n, s, e in adam[-40:]:
    o = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2 in others if e2 > s and s2 < e)
    tot += e - s; ov += min(o, e - s)
print("adam launches", len(adam), "last-40 total %.1f us, overlapped with other kernels %.1f us" % (tot / 1e3, ov / 1e3))
print("adam durations (us):", [round((e - s) / 1e3, 1) for _, s, e in adam[-19:]])
