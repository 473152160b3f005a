"""Per-kernel comparison of two hipcc -S listings (labels normalised, comments and directives dropped):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o new.s mint_amd/csrc/gemm_big.hip
    python tools/isa_diff.py old.s new.s
Used to show that a refactor or a build knob leaves the device code of the shipped kernels bit-for-bit alone (so the last
GPU test run still covers them)."""
import re
import sys


def kernels(path):
    out, cur = {}, None
    for l in open(path).read().split("\n"):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        if l.startswith(".Lfunc_end"):
            cur = None
            continue
        x = l.split(";")[0].rstrip()
        if not x.strip() or x.strip().startswith("."):
            continue
        out[cur].append(re.sub(r"\.LBB\d+_\d+", "L", x))
    return out


a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
same = [k for k in a if k in b and a[k] == b[k]]
changed = [k for k in a if k in b and a[k] != b[k]]
print("kernels: %d -> %d; identical %d, changed %d, removed %d, new %d" % (
    len(a), len(b), len(same), len(changed), len([k for k in a if k not in b]), len([k for k in b if k not in a])))
for k in changed:
    print("  changed:", k[:140], "(%d -> %d instructions)" % (len(a[k]), len(b[k])))
sys.exit(1 if changed else 0)
