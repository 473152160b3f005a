"""Kernel timeline of ONE cross-modal layer of the backward pass and one of the forward pass (rocprofv3
--kernel-trace db): start/end relative to the window, stream, duration; plus the step anatomy."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, stream_id from kernels order by start"))
def short(nm):
    nm = nm.replace("void ", "").replace("(anonymous namespace)::", "")
    if nm.startswith("_ZN12_GLOBAL__N_1"):
        nm = nm[len("_ZN12_GLOBAL__N_1"):].lstrip("0123456789")
    return nm.split("(")[0][:52]
idx = [i for i, r in enumerate(rows) if "mse_loss" in r[0]]
a, b = idx[-3], idx[-2]
win = rows[a:b]
t0 = win[0][1]
print("step window %.0f us" % ((win[-1][2] - t0) / 1e3))
tn = [i for i, r in enumerate(win) if "big_tn_kernel" in r[0]]
def dump(lo, hi, title):
    print("----", title)
    base = win[lo][1]
    for r in win[lo:hi]:
        print("s%-2d %8.1f -> %8.1f (%6.1f us)  %s" % (r[3], (r[1] - base) / 1e3, (r[2] - base) / 1e3, (r[2] - r[1]) / 1e3, short(r[0])))
if len(tn) > 7:
    dump(tn[5], tn[7] + 1, "backward: two cross layers (between wgrad launches 5..7)")
# forward: the first ln_fwd after concat_seq of the NEXT step portion inside this window
cc = [i for i, r in enumerate(win) if "concat_seq" in r[0]]
if cc:
    lo = cc[0] + 1 + 16
    dump(lo, lo + 18, "forward: ~two cross layers")
