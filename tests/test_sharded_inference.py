"""Rank sharding of auto-regressive inference (BASELINE.json configs[3]: 256 sequences over 8 GPUs, no
collective while generating, gather at the end): shard arithmetic, and a gloo world-2 run whose gathered rollout equals
the single-process one - sequences are independent in fact_model.py:103-132, so sharding must not change a single value."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mint_amd import sharding
from mint_amd.evaluator import SingleTaskEvaluator
from oracle import fact_oracle as O

CFG = {
    "motion": {"seq_len": 6, "feature_dim": 9, "hidden": 16, "layers": 1, "heads": 2, "ff": 32},
    "audio": {"seq_len": 10, "feature_dim": 4, "hidden": 16, "layers": 1, "heads": 2, "ff": 32},
    "cross": {"hidden": 16, "layers": 1, "heads": 2, "ff": 32},
    "out_dim": 9,
}
STEPS, SEQS = 5, 5  # 5 sequences over 2 ranks: uneven shards (3 + 2)


class _OracleSampler:
    """infer_auto_regressive of the reference (oracle restatement) behind the model surface the evaluator calls."""

    def __init__(self):
        self.params = O.init_params(CFG, seed=4)

    def infer_auto_regressive(self, inputs, steps=1200):
        return O.infer_auto_regressive(self.params, CFG, inputs["motion_input"], inputs["audio_input"], steps=steps)


def _global_batch():
    g = torch.Generator().manual_seed(11)
    return {"motion_input": torch.randn(SEQS, 6, 9, generator=g, dtype=torch.float64),
            "audio_input": torch.randn(SEQS, 10 + STEPS - 1, 4, generator=g, dtype=torch.float64),
            "motion_name": ["m%d" % i for i in range(SEQS)], "audio_name": ["a%d" % i for i in range(SEQS)]}


def test_shard_range_partitions_every_row_once():
    for total in (0, 1, 5, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert [sharding.shard_range(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]  # configs[3]
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def test_shard_inputs_cuts_per_sample_entries_only():
    b = dict(_global_batch(), note="x")
    mine = sharding.shard_inputs(b, 1, 2)
    assert mine["motion_input"].shape[0] == 2 and mine["audio_input"].shape[0] == 2
    assert mine["motion_name"] == ["m3", "m4"] and mine["note"] == "x"
    assert sharding.shard_inputs(b, 0, 1) is b
    assert sharding.gather_rows(b["motion_input"], SEQS, 0, 1) is b["motion_input"]


class _MeanAbs:
    """A metric in the evaluator's protocol (reset_states / update_state(inputs, outputs) / result)."""
    name = "mean_abs"

    def reset_states(self):
        self.s, self.n, self.names = 0.0, 0, []

    def update_state(self, inputs, outputs):
        self.s += float(outputs.abs().sum())
        self.n += outputs.numel()
        self.names += list(inputs["motion_name"])

    def result(self):
        return (self.s / max(self.n, 1), tuple(self.names))


def _worker(rank, world, path, outdir, q):
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", init_method="file://" + path, rank=rank, world_size=world)
    torch.set_num_threads(1)
    ev = SingleTaskEvaluator([_global_batch()], _OracleSampler(), [_MeanAbs()], output_dir=outdir, steps=STEPS)
    assert (ev.rank, ev.world_size) == (rank, world)
    ev.eval_begin()
    outputs, paths = ev.eval_step(iter(ev.eval_dataset))
    full = sharding.gather_rows(outputs, SEQS)
    # metrics saw the GLOBAL batch and the gathered outputs: the same whole-set value on every rank
    m, names = ev.eval_end()["mean_abs"]
    assert names == tuple("m%d" % i for i in range(SEQS)) and abs(m - float(full.abs().mean())) < 1e-12, (rank, m, names)
    q.put((rank, full.numpy().copy(), [os.path.basename(p) for p in paths]))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gathered_rollout_equals_single_process(tmp_path):
    fd, path = tempfile.mkstemp(prefix="mint_amd_rdzv_")
    os.close(fd)
    os.remove(path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, path, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, full, names = q.get(timeout=240)
        res[rank] = (full, names)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, all sequences
    batch = _global_batch()
    ref, _ = SingleTaskEvaluator([batch], _OracleSampler(), [], output_dir=None, steps=STEPS, rank=0,
                                 world_size=1).eval_step(iter([batch]))
    assert ref.shape == (SEQS, 6 + STEPS, 9)
    for rank in (0, 1):
        assert res[rank][0].shape == tuple(ref.shape)
        assert np.array_equal(res[rank][0], ref.numpy()), "rank %d: gathered rollout differs from the single-process one" % rank
    assert res[0][1] == ["m0_a0.npy", "m1_a1.npy", "m2_a2.npy"] and res[1][1] == ["m3_a3.npy", "m4_a4.npy"]
    for i in range(SEQS):  # every sequence saved exactly once, seed-prefixed (single_task_evaluator.py:71-84)
        a = np.load(tmp_path / ("m%d_a%d.npy" % (i, i)))
        assert np.array_equal(a, ref[i].numpy())
