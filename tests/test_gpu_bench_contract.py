"""The driver's contract with bench.py, run for real on the GPU box: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON
line on stdout with the headline fields, a `roofline` object for the dominant kernel and a `cpu_baseline` object - and the timed
region ran on the PRODUCTION library (round 6), the kernel table on the test / bench build in a child process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900):
    env = dict(os.environ)
    for k in ("FACT_DEBUG_ABI", "FACT_LIB", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly ONE line on stdout, got %d:\n%s" % (len(lines), out.stdout[-2000:])
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields_and_runs_the_production_library():
    d = _run(["--gpus", "1", "--steps", "4", "--warmup", "2", "--profile-steps", "1"])
    assert d["metric"].startswith("motion frames/sec (train step) fact_v5_deeper_t10_cm12")
    assert d["unit"] == "motion frames/sec" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith("fact_v5_deeper_t10_cm12 train step") and d["config"]["global_batch"] == 16
    assert d["value"] == pytest.approx(16 * 120 / (d["ms_per_step"] * 1e-3), rel=1e-3)
    assert 3.0 < d["ms_per_step"] < 60.0
    assert d["library"] == "libfact_hip.so", d["library"]              # what ran the timed steps
    assert "libfact_hip_dbg.so" in d["kernels_from"]                     # where the kernel-class table comes from
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] in ("TFLOP/s", "GB/s")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=2e-3) and 0.0 < r["frac"] < 1.0
    assert "traffic" in r and r["avg_launch_us"] > 0 and r["kernel"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "motion frames/sec" and c["sample"]
    names = {k["name"] for k in d["kernels"]}
    assert {"wgrad_group", "attention_fwd", "attention_bwd", "adam+shadows"} <= names


def test_bench_ab_runs_bind_the_debug_build_and_say_so():
    d = _run(["--steps", "3", "--warmup", "1", "--profile-steps", "1", "--no-cpu-baseline", "--opt", "bias_in_wgrad=0"])
    assert d["library"] == "libfact_hip_dbg.so" and "this process" in d["kernels_from"]
    assert "cpu_baseline" not in d and "roofline" in d
