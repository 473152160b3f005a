"""bench.py's FLOP accounting against an independent count from the shipped model dimensions (SURVEY section 8d:
80.97 GFLOP forward per sample = linear 75.35 + attention 5.44 + embed / head 0.19; train step = 3x = 242.92 GFLOP per
sample = 2.024 GFLOP per motion frame), and the share the supervised-rows shortcut does not execute."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)   # main() only runs under __main__
    return mod


D, FF, H, OUT = 800, 3072, 10, 225
STACKS = {"motion": (120, 2, 225), "audio": (240, 2, 35), "cross": (360, 12, None)}


def _forward_flops():
    linear = attention = 0.0
    for n, layers, _ in STACKS.values():
        linear += layers * 2.0 * n * (D * 3 * D + D * D + 2 * D * FF)
        attention += layers * 4.0 * n * n * D          # QK^T and PV, all heads
    embed_head = sum(2.0 * n * f * D for n, _, f in STACKS.values() if f) + 2.0 * 360 * D * OUT
    return linear, attention, embed_head


def test_flops_per_sample_and_frame(bench):
    linear, attention, embed_head = _forward_flops()
    assert linear / 1e9 == pytest.approx(75.35, abs=0.01)
    assert attention / 1e9 == pytest.approx(5.44, abs=0.01)
    assert embed_head / 1e9 == pytest.approx(0.19, abs=0.01)
    fwd = linear + attention + embed_head
    assert fwd / 1e9 == pytest.approx(80.97, abs=0.02)
    assert 3 * fwd / 120 == pytest.approx(bench.FLOP_PER_FRAME, rel=1e-3)
    assert bench.BATCH_PER_GPU == 16 and bench.TARGET_LEN == 20 and bench.PEAK_BF16_TFLOPS == 2500.0
    # headline arithmetic: frames/s at 100 % of the bf16 roofline (1.235 M frames/s per GPU)
    assert bench.PEAK_BF16_TFLOPS * 1e12 / bench.FLOP_PER_FRAME == pytest.approx(1.235e6, rel=2e-3)


def test_executed_fraction_of_the_supervised_rows_shortcut(bench):
    # last cross-modal layer: queries / to_out / MLP / head on 20 of 360 rows; LN1 + QKV stay on all rows
    n, t = 360, 20
    skipped = (n - t) * (2.0 * D * D + 4.0 * D * FF + 4.0 * n * D + 2.0 * D * OUT)
    total = 3 * sum(_forward_flops())
    assert bench.executed_flop_fraction() == pytest.approx(1.0 - 3 * skipped / total, rel=1e-4)
    assert bench.executed_flop_fraction() == pytest.approx(0.947, abs=1e-3)
    assert bench.executed_flop_fraction(target_len=360) == pytest.approx(1.0)


def test_kernel_name_and_cu_share_come_from_the_engine_record(bench):
    """The `kernel` / `cu_share` of a bench row are derived from what the engine's recorder saw launched (symbol, grid,
    occupancy), not from a hand-kept table: most frequent launch shape first, CUs touched = min(256, grid)."""
    rec = {"name": "attention_bwd", "kernels": [
        {"count": 24, "grid": 160, "block": 768, "lds_bytes": 147456, "workgroups_per_cu": 1, "name": "dq<80, 0>(AttnParams)"},
        {"count": 24, "grid": 160, "block": 512, "lds_bytes": 150528, "workgroups_per_cu": 1, "name": "dkdv<80, 0>(AttnParams)"},
        {"count": 8, "grid": 160, "block": 512, "lds_bytes": 100352, "workgroups_per_cu": 1, "name": "dq<80, 0>(AttnParams)"}]}
    ck = bench.class_kernels(rec)
    assert ck["kernel"] == "dq<80, 0>(AttnParams) + dkdv<80, 0>(AttnParams)"
    assert ck["cus_held"] == 160 and ck["cu_share"] == 0.625
    two = bench.class_kernels({"name": "out_proj+resid", "kernels": [
        {"count": 12, "grid": 162, "block": 512, "lds_bytes": 73728, "workgroups_per_cu": 2, "name": "big_nt<...>"}]})
    assert two["cus_held"] == 162 and two["workgroups_per_cu"] == 2
    full = bench.class_kernels({"name": "ln_fwd", "kernels": [
        {"count": 31, "grid": 1440, "block": 256, "lds_bytes": 0, "workgroups_per_cu": 8, "name": "ln_fwd_kernel<4>"}]})
    assert full["cus_held"] == 256
    assert bench.class_kernels({"name": "x", "kernels": []}) == {"kernel": "x", "cu_share": None}


# ---- `python bench.py --gpus N` is its own launcher (round 5): N ranks on one node, one process per GPU --------------------
def test_launcher_argv_is_the_drivers_command(bench):
    cmd = bench.launcher_argv(8, "/x/bench.py", ["--gpus", "8", "--steps", "5"], 29511, python="python")
    assert cmd == ["python", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                   "127.0.0.1", "--master-port", "29511", "/x/bench.py", "--gpus", "8", "--steps", "5"]


def test_gpus_flag_spawns_ranks_only_without_a_launcher(bench):
    seen = {}

    def fake_exec(path, argv, env):
        seen.update(path=path, argv=argv, env=env)

    # N = 1 and "already under a launcher" never re-exec
    assert bench.maybe_spawn_ranks(["bench.py"], {}, fake_exec) is None
    assert bench.maybe_spawn_ranks(["bench.py", "--gpus", "1"], {}, fake_exec) is None
    assert bench.maybe_spawn_ranks(["bench.py", "--gpus", "4"], {"WORLD_SIZE": "4"}, fake_exec) is None
    assert not seen
    cmd = bench.maybe_spawn_ranks(["bench.py", "--gpus=2", "--dist-backend", "gloo", "--steps", "3"], {"PATH": "/bin"},
                                  fake_exec)
    assert seen["argv"] == cmd and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == ["--gpus=2", "--dist-backend", "gloo", "--steps", "3"] and cmd[-6].endswith("bench.py")
    # the ranks inherit what must be set before the HIP runtime starts
    assert seen["env"]["GPU_MAX_HW_QUEUES"] == "7" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["env"]["PATH"] == "/bin"
    assert bench.requested_gpus(["--steps", "5", "--gpus", "8"]) == 8 and bench.requested_gpus(["--steps", "5"]) == 1


def test_library_selection_of_a_bench_process(bench):
    """Round 6: the driver's line (--mode train) times the PRODUCTION library; anything that needs the debug surface in the
    timed process itself - A/B knobs, the kernel-table child, the ar / scaled artefact modes - binds the test / bench build."""
    assert not bench.wants_debug_build(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert not bench.wants_debug_build(["--mode", "train", "--breakdown"])
    assert bench.wants_debug_build(["--opt", "bias_in_wgrad=0"]) and bench.wants_debug_build(["--opt=wgrad_parts=1"])
    assert bench.wants_debug_build(["--kernel-table-child", "--steps", "5"])
    assert bench.wants_debug_build(["--mode", "ar"]) and bench.wants_debug_build(["--mode=scaled"])
    assert bench.raw_flag(["--steps", "7", "--mode=ar"], "--mode") == "ar" and bench.raw_flag([], "--mode", "train") == "train"
