"""The PRODUCTION library (mint_amd/lib/libfact_hip.so: -fvisibility=hidden, include/fact_hip.h only) runs the hot path by
itself: every other GPU test binds the test / bench build (libfact_hip_dbg.so, tests/conftest.py), so this one starts a
fresh interpreter without FACT_DEBUG_ABI and runs what the driver's smoke() runs - tiny FACT forward vs the oracle, one
fused train step vs the oracle's loss - plus a check that debug entry points are refused there."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
import __graft_entry__ as g
from mint_amd import _lib as L
assert not L.DEBUG_ABI and L.LIB_PATH.endswith("libfact_hip.so"), L.LIB_PATH
g.smoke()
assert not hasattr(L.lib(), "fact_debug_set_option") and not hasattr(L.lib(), "fact_kprof")
from mint_amd import configs, model_builder
m = model_builder.build(configs.tiny_fact(), True)
m.build(2, 225, 35)
try:
    m.debug_option("skip", 1)
except RuntimeError as e:
    assert "FACT_DEBUG_ABI" in str(e)
else:
    raise SystemExit("debug_option accepted by the production library")
maps = open("/proc/self/maps").read()
assert "libfact_hip.so" in maps and "libfact_hip_dbg.so" not in maps
print("production library ok")
""" % ROOT


def test_production_library_runs_smoke_and_refuses_debug_entry_points():
    env = dict(os.environ)
    env.pop("FACT_DEBUG_ABI", None)
    env.pop("FACT_LIB", None)
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "smoke ok" in out.stdout and "production library ok" in out.stdout


def test_production_library_trainer_steps_vs_oracle_at_the_benchmark_config():
    """Round-5 review item 3: the benchmark path (fact_v5, batch 16, SingleTaskTrainer defaults, three optimizer steps)
    against the oracle ON THE PRODUCTION LIBRARY - the same bounds tests/test_gpu_model.py::
    test_fact_v5_trainer_steps_vs_oracle enforces on the test / bench build (tests/_trainer_parity.py)."""
    env = dict(os.environ)
    env.pop("FACT_DEBUG_ABI", None)
    env.pop("FACT_LIB", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_trainer_parity.py")], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "trainer parity ok on libfact_hip.so" in out.stdout, out.stdout[-2000:]
