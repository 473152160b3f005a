"""Host-side data-parallel logic on CPU (gloo, world_size 2): per-replica loss/R, SUMMED gradient
all-reduce in buckets, identical replicas after the step, equality with single-process training on
the global batch (single_task_trainer.py:157-158, 186-187)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mint_amd.trainer import Adam, SingleTaskTrainer, allreduce_gradients, train
from oracle import fact_oracle as O
from tests._oracle_model import BucketedOracleModel, OracleModel

CFG = {  # smaller than TINY to keep the CPU suite fast
    "motion": {"seq_len": 8, "feature_dim": 12, "hidden": 32, "layers": 1, "heads": 2, "ff": 64},
    "audio": {"seq_len": 16, "feature_dim": 5, "hidden": 32, "layers": 1, "heads": 2, "ff": 64},
    "cross": {"hidden": 32, "layers": 1, "heads": 2, "ff": 64},
    "out_dim": 12,
}


def _free_port():
    """A rendezvous FILE, not a TCP port: picking a free port and handing it to two spawned processes races with
    everything else on the host (seen as a rare spurious failure); a file store has no such window."""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="mint_amd_rdzv_")
    os.close(fd)
    os.remove(path)  # the file store creates it
    return path


def _plain(metrics):
    """metric values as Python floats (no tensors through the multiprocessing queue)"""
    return {k: (float(v) if torch.is_tensor(v) else v) for k, v in metrics.items()} if isinstance(metrics, dict) else metrics


def _init(rank, world, path):
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", init_method="file://" + path, rank=rank, world_size=world)


def _worker(rank, world, port, q):
    _init(rank, world, port)
    torch.set_num_threads(1)
    full = O.synthetic_batch(CFG, 4, 4, seed=3)
    mine = {k: v[2 * rank:2 * rank + 2] for k, v in full.items()}
    model = OracleModel(CFG)
    trainer = SingleTaskTrainer([mine, mine], "target", model, optimizer=Adam(1e-2))
    assert trainer.num_replicas_in_sync == 2
    hist = train(trainer, steps=2, steps_per_loop=2)
    # numpy, not a tensor: a tensor crosses the queue as a shared-memory fd that dies with this process (rare
    # ConnectionResetError in the parent when the worker exits first)
    q.put((rank, model.flat_params().detach().cpu().numpy().copy(), _plain(hist[-1][1])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_replicas_match_single_process_global_batch():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, flat, metrics = q.get(timeout=240)
        res[rank] = (torch.from_numpy(flat).clone(), metrics)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(res[0][0], res[1][0]), "replicas diverged"
    # single process, global batch of 4
    full = O.synthetic_batch(CFG, 4, 4, seed=3)
    ref = OracleModel(CFG)
    t = SingleTaskTrainer([full, full], "target", ref, optimizer=Adam(1e-2))
    hist = train(t, steps=2, steps_per_loop=2)
    assert torch.allclose(res[0][0], ref.flat_params(), rtol=1e-9, atol=1e-12)
    # training_loss aggregates with SUM over replicas of (local mean / R) == global mean
    assert abs(res[0][1]["training_loss"] - hist[-1][1]["training_loss"]) < 1e-6  # metric all-reduce is fp32
    assert res[0][1]["learning_rate"] == pytest.approx(1e-2)


def _worker_overlap(rank, world, port, q, bf16):
    """The REAL OverlappedGradReducer (bucket callback protocol, async all-reduces in flight while later buckets
    are still being produced, finish() before the optimizer) under world_size 2."""
    _init(rank, world, port)
    torch.set_num_threads(1)
    full = O.synthetic_batch(CFG, 4, 4, seed=3)
    mine = {k: v[2 * rank:2 * rank + 2] for k, v in full.items()}
    model = BucketedOracleModel(CFG)
    trainer = SingleTaskTrainer([mine, mine], "target", model, optimizer=Adam(1e-2), overlap_grad_allreduce=True,
                                bf16_grad_buckets=bf16)
    seen = []
    orig = None

    def spy(bucket, off, cnt):
        seen.append((bucket, off, cnt))
        orig(bucket, off, cnt)
    trainer.train_loop_begin()
    trainer.train_step()                      # creates the reducer on first use
    red = trainer._reducer
    assert red is not None and red.bf16 == bf16
    orig = red._on_bucket
    model.set_grad_callback(spy, None)         # observe the second step's bucket traffic
    trainer.train_step()
    metrics = trainer.train_loop_end()
    q.put((rank, model.flat_params().detach().cpu().numpy().copy(), _plain(metrics), seen, list(model.buckets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32-buckets", "bf16-buckets"])
def test_overlapped_reducer_world2(bf16):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_overlap, args=(r, 2, port, q, bf16)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, flat, metrics, seen, buckets = q.get(timeout=240)
        res[rank] = (torch.from_numpy(flat).clone(), metrics, seen, buckets)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    flat0, _, seen, buckets = res[0]
    assert torch.isfinite(flat0).all(), "a bucket was reduced before it was reported (NaN poison)"
    assert torch.equal(flat0, res[1][0]), "replicas diverged"
    # every bucket reported exactly once, in the engine's order, covering the arena
    assert [(b, o, c) for b, (o, c) in enumerate(buckets)] == seen
    assert sorted(o for o, _ in buckets)[0] == 0 and sum(c for _, c in buckets) == flat0.numel()
    # against single-process training on the global batch of 4
    full = O.synthetic_batch(CFG, 4, 4, seed=3)
    ref = OracleModel(CFG)
    t = SingleTaskTrainer([full, full], "target", ref, optimizer=Adam(1e-2))
    train(t, steps=2, steps_per_loop=2)
    if not bf16:
        assert torch.allclose(flat0, ref.flat_params(), rtol=1e-9, atol=1e-12)
    else:
        # each replica's contribution is rounded to bf16 (2^-9 relative) before the sum; Adam's first steps move
        # every weight by ~lr regardless of the gradient scale, so compare the UPDATE direction and size
        p0 = OracleModel(CFG).flat_params()
        du, dr = flat0 - p0, ref.flat_params() - p0
        cosv = float((du @ dr) / (du.norm() * dr.norm()))
        assert cosv > 0.999, cosv
        assert float((du - dr).norm() / dr.norm()) < 5e-2


def test_allreduce_is_noop_without_process_group():
    g = torch.arange(10, dtype=torch.float32)
    assert allreduce_gradients(g) == []
    assert torch.equal(g, torch.arange(10, dtype=torch.float32))


def test_trainer_pops_label_and_reports_metrics():
    full = O.synthetic_batch(CFG, 2, 4, seed=1)
    model = OracleModel(CFG)
    seen = []
    tr = SingleTaskTrainer([dict(full, motion_name="x")], "target", model, optimizer=Adam(1e-3),
                           summary_fn=lambda d, step: seen.append((sorted(d), step)))
    tr.train_loop_begin()
    loss = tr.train_step()
    m = tr.train_loop_end()
    assert set(m) == {"training_loss", "task_loss", "regularization_loss", "learning_rate"}
    assert m["regularization_loss"] == 0.0 and abs(m["training_loss"] - float(loss)) < 1e-12
    assert seen == [(["loss:", "reg_loss", "total_loss"], 0)]  # reference's key typo kept
    assert model.global_step == 1 and tr.optimizer.iterations == 1
