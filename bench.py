"""Headline benchmark: motion frames/sec of the FACT train step (fact_v5_deeper_t10_cm12, bf16
MFMA compute / fp32 master weights, batch 16 per GPU, AIST++-shaped synthetic tensors).

  python bench.py --gpus N --steps K --warmup W
  (N>1: either under a launcher - python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ... -
   or bare: without WORLD_SIZE in the environment `python bench.py --gpus N` re-executes itself under that launcher,
   one rank per GPU; every rank checks WORLD_SIZE == --gpus)

A step = one pass of the hot path over one batch: forward, MSE loss on 20 target frames, backward,
RCCL gradient all-reduce (N>1), Keras-Adam update, bf16 weight-shadow refresh.  Prints ONE JSON line
(rank 0) with
  * `kernels`: every kernel class of the step, timed IN the step (all streams overlapping as usual) with HIP
    events recorded on the stream each launch goes to (engine option fact_kprof, a few extra steps after the
    timed region): launches per step, average launch duration, share of the summed kernel time, achieved
    TFLOP/s or GB/s and the fraction of the bound that applies;
  * `roofline`: the DOMINANT class of that table (largest time share);
  * `attention`: the cross-modal attention sub-row the north star names (forward / backward TFLOP/s);
  * `cpu_baseline`: the oracle (PyTorch-CPU fp32 restatement of the reference train step) timed on this box's
    host cores at N=1: 1 warm-up + 2-3 timed steps at the per-GPU batch of 16 (a batch-4 sample as a side note).

Other modes (artefacts for profiles/, not the driver's line):
  --mode ar      BASELINE.json configs[3] per-GPU share: 32 sequences, 120-frame seed, --steps generated frames
  --mode scaled  BASELINE.json configs[4]: one full-depth scaled-FACT train step (d=1536, 24 cross layers,
                 seq 480/960) at --batch sequences
"""
import argparse
import json
import os
import sys
import time



def launcher_argv(gpus, script, script_args, port, python=None):
    """argv that starts `gpus` ranks of this script on ONE node, one process per GPU (what the driver types for N > 1):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ..."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), script] + list(script_args)


def requested_gpus(argv):
    """--gpus N / --gpus=N from a raw argv (read before argparse and before torch is imported); 1 when absent."""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    return n


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def maybe_spawn_ranks(argv=None, environ=None, execve=None):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): become the launcher - re-exec
    under torch.distributed.run with N ranks (reference: trainer.py:125-135,146-147 builds a MirroredStrategy over all
    visible GPUs from ONE command).  Under a launcher (WORLD_SIZE set) this is a no-op; main() then checks
    WORLD_SIZE == --gpus.  Returns the argv it would exec (tests pass `execve` to capture it)."""
    argv = sys.argv if argv is None else argv
    environ = os.environ if environ is None else environ
    n = requested_gpus(argv[1:])
    if n <= 1 or "WORLD_SIZE" in environ:
        return None
    env = dict(environ)
    env.setdefault("GPU_MAX_HW_QUEUES", "7")          # see below: must be in the ranks' environment before HIP starts
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launcher_argv(n, os.path.abspath(argv[0]), argv[1:], free_port())
    (execve or os.execve)(cmd[0], cmd, env)
    return cmd


if __name__ == "__main__":
    maybe_spawn_ranks()

# Which build of the library a bench process binds (mint_amd/_lib.py; ONE per process):
#   --mode train (the driver's line): the PRODUCTION library libfact_hip.so runs the warm-up and the timed K steps; the
#     `kernels` / `roofline` objects need the in-step kernel-class recorder of include/fact_hip_debug.h, so rank 0 then starts a
#     CHILD of this script (--kernel-table-child) that binds libfact_hip_dbg.so - the same objects + the recorder - on the same
#     GPU and reports the table of the same single-replica step;
#   --opt KEY=INT (A/B knobs are debug options), --kernel-table-child, --mode ar / scaled (artefacts with a class table):
#     the test / bench build for the whole process, and the line says so in `library`.
# FACT_DEBUG_ABI in the environment overrides either way.
def raw_flag(argv, name, default=None):
    for i, a in enumerate(argv):
        if a == name and i + 1 < len(argv):
            return argv[i + 1]
        if a.startswith(name + "="):
            return a.split("=", 1)[1]
    return default


def wants_debug_build(argv):
    return (any(a == "--opt" or a.startswith("--opt=") for a in argv) or "--kernel-table-child" in argv
            or raw_flag(argv, "--mode", "train") != "train")


if __name__ == "__main__":
    os.environ.setdefault("FACT_DEBUG_ABI", "1" if wants_debug_build(sys.argv[1:]) else "0")

# Data-parallel runs add a communication stream and RCCL's own to the engine's three: with HIP's default of 4
# hardware queues some of them share a queue and serialise (one-GPU dry run with RCCL initialised, tools/attic/dp_probe.py:
# 11.4 ms per step at 4 queues, 8.2 at 7, 16.5 at 8).  Must be set before the HIP runtime starts.
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "7")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_FRAME = 2.024e9      # 242.92 GFLOP / sample / 120 frames


def executed_flop_fraction(target_len=20, n=360, d=800, ff=3072, out_dim=225, total_per_sample=242.92e9):
    """The train step skips work that cannot reach the loss (supervised-rows shortcut): in the last
    cross-modal layer the attention queries, to_out, the MLP and the head run on the `target_len` supervised rows
    of each sequence only, forward and backward.  Returns executed / algorithmic FLOPs of the reference step."""
    skipped_rows = n - target_len
    fwd = skipped_rows * (2.0 * d * d + 4.0 * d * ff + 4.0 * n * d + 2.0 * d * out_dim)
    return 1.0 - 3.0 * fwd / total_per_sample
PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
BATCH_PER_GPU = 16
TARGET_LEN = 20


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # >= 50 timed steps after >= 10 warm-up
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 16 train, 32 ar, 8 scaled)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="--mode ar: total sequences over all ranks (strong scaling, e.g. 256 = configs[3]); default "
                         "--batch per GPU (weak scaling: 256 at N = 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-phase timing to stderr")
    ap.add_argument("--side-stream", type=int, default=1, help="0 = single-stream engine (A/B knob)")
    ap.add_argument("--fuse-optimizer", type=int, default=1, help="0 = Adam as a separate pass after backward (A/B knob)")
    ap.add_argument("--mode", choices=["train", "ar", "scaled"], default="train")
    ap.add_argument("--profile-steps", type=int, default=4, help="extra steps for the in-step kernel table")
    ap.add_argument("--parity", action="store_true",
                    help="--mode scaled: also run ONE full-depth step at batch 1 against the fp32 CPU oracle "
                         "(loss and every gradient tensor; takes a few minutes of host time)")
    ap.add_argument("--grad-buckets", choices=["bf16", "fp32"], default="bf16",
                    help="N > 1: dtype of the all-reduced gradient buckets (bf16: 240.8 MB per step, fp32: 481.6 MB)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = DRY RUN of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share "
                         "devices, the collective goes through the host): the timing means nothing")
    ap.add_argument("--dp-aux-stream", type=int, default=0,
                    help="N > 1: 1 = keep the engine's third stream (motion-encoder backward chain) under data parallelism. "
                         "Default 0: with the communication stream and RCCL's own the process then stays at <= 6 busy "
                         "hardware queues (GPU_MAX_HW_QUEUES = 7; an 8th busy queue doubled the step in the one-GPU dry "
                         "run, profiles/r02_dp_dry_run.txt) at ~0.2 ms per step")
    ap.add_argument("--kernel-table-child", action="store_true",
                    help="internal: warm up, record the in-step kernel-class table on libfact_hip_dbg.so, print it as JSON")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT",
                    help="engine test / bench knob for A/B runs (fact_debug_set_option), e.g. --opt wgrad_parts=1")
    return ap.parse_args()


PEAK_HBM_GBS = 8000.0         # MI355X HBM3E (MI355X_MICROARCH.md)
N_CUS = 256


def class_kernels(rec):
    """What the engine's recorder saw behind one class of the table (FACT_LAUNCH notes, fact_kprof_kernels): the symbol
    of the most frequent launch shape exactly as rocprofv3 prints it (the cross-modal layers outnumber the encoder
    layers 12 : 4, so this is the cross-modal launch), its grid, the runtime's occupancy answer for the launch shape
    (workgroups per CU) and the CUs the launch is spread over, min(256, grid)."""
    ks = rec.get("kernels") or []
    if not ks:
        return {"kernel": rec["name"], "cu_share": None}
    top = ks[0]
    names = []
    for k in ks:  # distinct symbols of the class, most frequent first (attention backward: dQ and dK/dV kernels)
        if k["name"] not in names:
            names.append(k["name"])
    # the dispatcher deals the workgroups of a launch round-robin over XCDs and CUs: a grid of G workgroups touches
    # min(256, G) CUs (a 2-per-CU kernel with 161 workgroups sits half-filled on 161 CUs, it is not packed onto 81)
    cus = min(N_CUS, top["grid"])
    return {"kernel": " + ".join(names[:2]), "grid": top["grid"], "block": top["block"], "lds_bytes": top["lds_bytes"],
            "workgroups_per_cu": top["workgroups_per_cu"], "cus_held": cus, "cu_share": round(cus / N_CUS, 3)}


def kernel_table(model, step_fn, nsteps):
    """In-step per-kernel-class timing: arm the engine's event recorder, run `nsteps` normal train steps, read
    the records back.  Durations are what rocprofv3 --kernel-trace reports for the same kernels (events bracket
    each launch on its own stream); the encoder stacks' launches of a class are averaged in with the cross-modal
    ones (4 of the 16 layers run at 1/3 and 2/3 of the tokens)."""
    model.kernel_profile(True)
    for _ in range(nsteps):
        step_fn()
    recs = model.kernel_profile()
    model.kernel_profile(False)
    tot = sum(r["total_ms"] for r in recs) or 1.0
    rows = []
    for r in recs:
        if r["launches"] <= 0:
            continue
        ck = class_kernels(r)
        row = {"name": r["name"], "kernel": ck["kernel"], "grid": ck.get("grid"),
               "workgroups_per_cu": ck.get("workgroups_per_cu"),
               "launches_per_step": round(r["launches"] / nsteps, 1),
               "avg_launch_us": round(r["total_ms"] * 1e3 / r["launches"], 2),
               "ms_per_step": round(r["total_ms"] / nsteps, 3),
               "time_share": round(r["total_ms"] / tot, 4)}
        if r["flops"] > 0:
            tf = r["flops"] / (r["total_ms"] * 1e-3) / 1e12
            row.update(bound="mfma", achieved=round(tf, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                       frac=round(tf / PEAK_BF16_TFLOPS, 4), flop_per_launch=r["flops"] / r["launches"])
            if r["bytes"] > 0:  # wgrad launches that carry the optimizer step of their tensors in the epilogue
                row["fused_optimizer_bytes_per_launch"] = r["bytes"] / r["launches"]
            if ck.get("cu_share"):
                # the whole-chip `frac` of a class that is deliberately given part of the chip understates the kernel:
                # frac_of_held_cus = frac / cu_share is the per-CU figure
                row.update(cu_share=ck["cu_share"], frac_of_held_cus=round(tf / PEAK_BF16_TFLOPS / ck["cu_share"], 4))
        else:
            gbs = r["bytes"] / (r["total_ms"] * 1e-3) / 1e9
            row.update(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                       frac=round(gbs / PEAK_HBM_GBS, 4), bytes_per_launch=r["bytes"] / r["launches"])
        rows.append(row)
    rows.sort(key=lambda x: -x["time_share"])
    for row in rows:
        row["clock"] = "engine event recorder"
    return rows, tot / nsteps


def wgrad_standalone(device, K=5760, d=800, ff=3072, iters=20):
    """The dominant kernel ALONE on the chip: one grouped whole-K launch of a cross-modal layer's four weight gradients
    (190 tiles of 160x256), HIP events on the launch stream.  In the step the same kernel runs as two 95-workgroup
    launches beside the dgrad chain, so its in-step per-launch rate (the `roofline` figures) is bounded by the 37 % of
    the CUs it is given; this is the rate of the kernel itself."""
    import ctypes as C
    from mint_amd import _lib as L
    lib = L.lib()
    g = torch.Generator(device=device).manual_seed(1)
    rp = lambda c: (c + 63) // 64 * 64

    def mk(cols):
        t = torch.zeros(K, rp(cols), device=device, dtype=torch.bfloat16)
        t[:, :cols] = torch.randn(K, cols, device=device, generator=g).to(torch.bfloat16)
        return t
    xin, gact, h2, dpre, att, xmid, h1, dqkv = mk(d), mk(ff), mk(d), mk(ff), mk(d), mk(d), mk(d), mk(3 * d)
    outs = [torch.zeros(ff, d, device=device), torch.zeros(d, ff, device=device), torch.zeros(d, d, device=device),
            torch.zeros(d, 3 * d, device=device)]
    probs = [(xin, d, gact, ff, outs[0], 1), (h2, d, dpre, ff, outs[1], 0), (att, d, xmid, d, outs[2], 0),
             (h1, d, dqkv, 3 * d, outs[3], 0)]
    n = 4
    VP, IA = C.c_void_p * n, C.c_int * n
    a_ = VP(*[q[0].data_ptr() for q in probs]); lda = IA(*[q[0].stride(0) for q in probs])
    b_ = VP(*[q[2].data_ptr() for q in probs]); ldb = IA(*[q[2].stride(0) for q in probs])
    o_ = VP(*[q[4].data_ptr() for q in probs]); ldo = IA(*[q[4].stride(0) for q in probs])
    mo = IA(*[q[1] for q in probs]); no = IA(*[q[3] for q in probs]); tr = IA(*[q[5] for q in probs])
    run = lambda: L.check(lib.fact_op_gemm_tn_group(n, a_, lda, b_, ldb, o_, ldo, mo, no, tr, K, L.cur_stream()))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    flop = 2.0 * K * (2.0 * d * ff + 4.0 * d * d)
    tf = flop / us / 1e6
    return {"what": "one launch of all 190 tiles of a cross-modal layer (86.1 GFLOP), alone on the chip",
            "avg_launch_us": round(us, 2), "achieved": round(tf, 1), "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4)}


def rocprof_symbol_table():
    """Average kernel durations by symbol from the newest committed `rocprofv3 --kernel-trace --stats` summary of this very
    command (profiles/rNN_kernel_stats_bench_n1.txt): the SECOND clock beside the engine's event recorder.  The recorder
    brackets a launch with HIP events on its stream and so includes the launch gap (~2-3 us more per launch than the
    profiler's begin-to-end kernel time); a symbol that serves several classes (the 256x160 bf16 GEMM runs both N = 800
    dgrads) carries one average for all of them.  Returns ({symbol: avg_us}, file name)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_kernel_stats_bench_n1.txt")))
    if not files:
        return {}, None
    tab = {}
    for line in open(files[-1]):
        m = re.match(r"^(.*\S)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip("\n"))
        if m:
            tab[m.group(1).strip()] = float(m.group(4))
    return tab, os.path.basename(files[-1])


def measured_traffic(name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes, with provenance (file +
    commit); null when the committed profile is for another kernel."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json")))
        if t.get("class") == name:
            return t.get("traffic_bytes"), t.get("source")
    except Exception:
        pass
    return None, None


def usable_cores():
    """Cores this process may actually run on: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the host's cores even inside a quota-limited container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def _time_oracle(O, cfg, B, timed, budget_s, max_timed):
    """1 warm-up + >= `timed` timed oracle train steps at batch B (as many as fit into budget_s, at most max_timed);
    returns (frames per second, seconds timed, steps timed, warm-up seconds)."""
    params = O.init_params(cfg, seed=0, dtype=torch.float32)
    batch = O.synthetic_batch(cfg, B, TARGET_LEN, seed=0, dtype=torch.float32)
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(x) for k, x in params.items()}
    t_warm = time.perf_counter()
    params, m, v = _oracle_step(O, params, m, v, 0, cfg, batch)
    t_warm = time.perf_counter() - t_warm
    if t_warm > 30.0:  # very slow host: report the warm-up step itself rather than blow the time budget
        return B * 120 / t_warm, t_warm, 0, t_warm
    t0 = time.perf_counter()
    n = 0
    while n < timed or (n < max_timed and time.perf_counter() - t0 < budget_s):
        params, m, v = _oracle_step(O, params, m, v, 1 + n, cfg, batch)
        n += 1
    dt = time.perf_counter() - t0
    return n * B * 120 / dt, dt, n, t_warm


def cpu_baseline():
    """Oracle (PyTorch-CPU fp32 restatement of the reference train step: forward, loss, backward, Keras Adam) on a
    bounded sample of the same workload, on all usable host cores: `value` = fact_v5 at the per-GPU batch of 16 that
    SURVEY 8(d) names, 1 warm-up + 2-3 timed steps (~15 s of CPU work); a batch-4 sample (1 warm-up + 2 timed steps)
    rides along as a side note - the two differ by what the host's caches make of the 4x larger activations."""
    from oracle import fact_oracle as O
    cores = min(usable_cores(), 64)
    torch.set_num_threads(cores)
    cfg = O.FACT_V5_CFG
    fps, dt, n, warm = _time_oracle(O, cfg, BATCH_PER_GPU, timed=2, budget_s=10.0, max_timed=3)
    note = ("1 warm-up + %d timed train steps" % n) if n else "1 (cold) train step"
    out = {"value": round(fps, 2), "unit": "motion frames/sec", "cores": cores, "kind": "port",
           "sample": "%s of fact_v5 at batch %d (%.1f s timed, warm-up %.1f s; fp32 PyTorch-CPU oracle, %d threads)" % (
               note, BATCH_PER_GPU, dt, warm, cores)}
    if n:
        fps4, dt4, n4, _ = _time_oracle(O, cfg, 4, timed=2, budget_s=2.0, max_timed=4)
        out["batch4_sample"] = {"frames_per_sec": round(fps4, 2), "seconds": round(dt4, 2), "timed_steps": n4}
    return out


def _oracle_step(O, params, m, v, step, cfg, batch):
    _loss, _grads, p, m, v = O.train_step(params, m, v, step, cfg, batch, 1e-4)
    return p, m, v


def class_table(model, psteps):
    """Kernel classes recorded by the engine's event recorder since kernel_profile(True), per step."""
    recs = model.kernel_profile()
    model.kernel_profile(False)
    return [{"name": r["name"], "launches_per_step": round(r["launches"] / psteps, 1),
             "avg_launch_us": round(r["total_ms"] * 1e3 / max(r["launches"], 1), 2),
             "ms_per_step": round(r["total_ms"] / psteps, 4),
             **({"tflops": round(r["flops"] / r["total_ms"] / 1e9, 1)} if r["flops"] > 0 and r["total_ms"] > 0 else
                {"GBps": round(r["bytes"] / r["total_ms"] / 1e6, 1)} if r["bytes"] > 0 and r["total_ms"] > 0 else {})}
            for r in sorted(recs, key=lambda r: -r["total_ms"]) if r["launches"] > 0]


def sync_all(world):
    """Barrier over the ranks + device drain: both sides of every timed region."""
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(dt, world, device, dry):
    """The job's time is the slowest rank's (MAX all-reduce; through the host under the gloo dry run)."""
    if world == 1:
        return dt
    import torch.distributed as dist
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dry else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def apply_opts(model, args):
    for kv in args.opt:
        k, v = kv.split("=")
        model.debug_option(k, int(v))


def make_trainer(model, batch, opt, args, world):
    """The data-parallel train-step structure of every training mode: summed bucketed all-reduce of the gradient arena
    on a communication stream overlapped with backward (bf16 buckets unless --grad-buckets fp32), Adam of a bucket
    behind its all-reduce; at N = 1 the single-replica step (optimizer inside backward)."""
    from mint_amd.trainer import SingleTaskTrainer

    class Repeat:
        def __iter__(self):
            return self

        def __next__(self):
            return batch
    trainer = SingleTaskTrainer(Repeat(), "target", model, optimizer=opt, fuse_optimizer=bool(args.fuse_optimizer),
                                bf16_grad_buckets=(world > 1 and args.grad_buckets == "bf16"),
                                overlap_grad_allreduce=(True if world > 1 else None))
    return trainer, iter(Repeat())


def timed_train_steps(trainer, it, args, world, device, dry):
    """W untimed warm-up steps, then exactly K steps between (barrier + synchronize) pairs; MAX over ranks."""
    trainer.train_loop_begin()
    for _ in range(args.warmup):
        trainer.train_step(it)
    sync_all(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.train_step(it)
    sync_all(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, device, dry)
    return dt, float(loss)


def comm_record(trainer, it, args, world):
    """Per-bucket all-reduce time and exposed communication of the N > 1 step (after the timed region)."""
    if world == 1 or getattr(trainer, "_reducer", None) is None:
        return None
    trainer._reducer.profile = True
    for _ in range(max(1, args.profile_steps)):
        trainer.train_step(it)
    comm = trainer._reducer.comm_report()
    trainer._reducer.profile = False
    return comm


def run_ar(args, device, world=1, rank=0, dry=False):
    """BASELINE.json configs[3]: auto-regressive generation, 120-frame seed -> `steps` frames per sequence (full
    forward per frame: no KV cache is possible, fact_model.py:103-132).  The global batch of sequences
    (--global-batch; default 32 per GPU, i.e. the 256 of configs[3] at N = 8) is sharded over the ranks with NO
    collective while sequences are generated (mint_amd/sharding.py): rank r generates sequences
    [r*G/N, (r+1)*G/N) of the same seeded global input; afterwards ONE all-gather collects the (G, steps, 225) result and
    rank 0 checks it."""
    from mint_amd import configs, model_builder, sharding
    per = args.batch or 32
    G = args.global_batch or per * world
    lo, hi = sharding.shard_range(G, rank, world)
    B = hi - lo
    steps = args.steps
    pipe = configs.fact_v5_deeper_t10_cm12()
    model = model_builder.build(pipe.multi_modal_model, False)
    gen = torch.Generator().manual_seed(7)
    glob = {"motion_input": torch.randn(G, 120, 225, generator=gen),
            "audio_input": torch.randn(G, 240 + steps - 1, 35, generator=gen)}
    inp = {k: v.to(device) for k, v in sharding.shard_inputs(glob, rank, world).items()}
    del glob
    model.build(B, 225, 35)
    apply_opts(model, args)
    model.infer_auto_regressive(inp, steps=min(steps, max(1, args.warmup)))
    sync_all(world)
    t0 = time.perf_counter()
    out = model.infer_auto_regressive(inp, steps=steps)
    sync_all(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, device, dry)
    assert out.shape == (B, steps, 225)
    full = sharding.gather_rows(out, G, rank, world)  # after the timed region: the only communication of the path
    gathered = {"shape": list(full.shape), "finite": bool(torch.isfinite(full).all()),
                "rms": round(float(full.double().pow(2).mean().sqrt()), 5)}
    assert tuple(full.shape) == (G, steps, 225) and gathered["finite"], gathered
    del full
    parity = None
    if args.parity and rank == 0:
        # the same generation with the one-kept-row last layer and the split-K small-M GEMMs switched off (sr_rows = 0):
        # every frame of the full-length rollout finite, and the two rollouts equal frame range by frame range
        model.set_option("sr_rows", 0)
        ref = model.infer_auto_regressive(inp, steps=steps)
        torch.cuda.synchronize()
        model.set_option("sr_rows", 1)
        rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
        edges = [0, 1, 10, 100, 400, 800, steps]
        parity = {"finite": bool(torch.isfinite(out).all() and torch.isfinite(ref).all()),
                  "rel_diff_all_frames": round(rel(out, ref), 6),
                  "rel_diff_by_frame_range": {"%d-%d" % (a, b): round(rel(out[:, a:b], ref[:, a:b]), 6)
                                              for a, b in zip(edges[:-1], edges[1:]) if b > a and b <= steps},
                  "rms_output": round(float(out.double().pow(2).mean().sqrt()), 5),
                  "reference": "same engine, sr_rows = 0 (all 360 rows through the last layer, single-pass GEMMs)"}
    fwd_flop = 80.97e9 * G  # BASELINE.md section 2, forward FLOPs per sample
    # kernel classes of the sampler (engine event recorder, outside the timed region): ms per generated frame-step
    kern = None
    if args.profile_steps > 0 and rank == 0:
        psteps = min(args.profile_steps, steps)
        model.kernel_profile(True)
        model.infer_auto_regressive(inp, steps=psteps)
        kern = class_table(model, psteps)
    if world > 1:
        sync_all(world)
    if rank != 0:
        return
    out_line = {
        "metric": "generated motion frames/sec (auto-regressive inference) fact_v5_deeper_t10_cm12",
        "value": round(G * steps / dt, 1), "unit": "generated frames/sec", "n_gpus": world, "steps": steps,
        "warmup": args.warmup, "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak" if not args.global_batch else "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "AR inference 120-frame seed -> %d frames (BASELINE.json configs[3])" % steps,
                   "global_sequences": G, "per_gpu_batch": B, "audio_frames": 240 + steps - 1,
                   "parallelism": "dp%d: sequences sharded by rank, no collective while generating, one all-gather of the "
                                  "(%d, %d, 225) result" % (world, G, steps)},
        "gathered_output": gathered,
        "forward_tflops": round(fwd_flop * steps / dt / 1e12, 1),
        "forward_mfma_frac": round(fwd_flop * steps / dt / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
        "parity_full_rows": parity, "kernels": kern}
    if dry:
        out_line["dry_run"] = "gloo, ranks sharing devices: control-flow check of the N > 1 path, not a measurement"
    print(json.dumps(out_line))


def run_scaled(args, device, world=1, rank=0, dry=False):
    """BASELINE.json configs[4]: full-depth scaled FACT (d=1536, 12 heads, ff=6144, 2+2 encoder and 24
    cross-modal layers, seq 480/960 -> n=1440: streaming attention kernels), train steps at --batch sequences per GPU
    through the same data-parallel step as --mode train (bf16 gradient buckets overlapped with backward; 1.59 GB per
    step at N > 1)."""
    from mint_amd import configs, model_builder
    from mint_amd.trainer import Adam
    B = args.batch or 8
    mm = configs.fact_config(motion=(480, 225, 1536, 2, 12, 6144), audio=(960, 35, 1536, 2, 12, 6144),
                             cross=(1536, 24, 12, 6144))
    model = model_builder.build(mm, True)
    gen = torch.Generator().manual_seed(9 + rank)
    batch = {"motion_input": torch.randn(B, 480, 225, generator=gen).to(device),
             "audio_input": torch.randn(B, 960, 35, generator=gen).to(device),
             "target": torch.randn(B, TARGET_LEN, 225, generator=gen).to(device)}
    model.build(B, 225, 35)
    if world > 1 and not args.dp_aux_stream:
        model.set_option("aux_stream", 0)
    apply_opts(model, args)
    nparams = sum(int(v.numel()) for v in model.trainable_variables)
    tr, it = make_trainer(model, batch, Adam(1e-4), args, world)
    dt, loss = timed_train_steps(tr, it, args, world, device, dry)
    dt /= args.steps
    kern = None
    if args.profile_steps > 0:
        psteps = min(args.profile_steps, 2)
        model.kernel_profile(True)
        for _ in range(psteps):
            tr.train_step(it)
        kern = class_table(model, psteps)
    comm = comm_record(tr, it, args, world)
    parity = None
    if args.parity and rank == 0 and world == 1:
        parity = scaled_parity(model, device)
    if world > 1:
        sync_all(world)
    if rank != 0:
        return
    d, ff, n = 1536, 6144, 1440
    lin = lambda tokens, layers: 2.0 * tokens * layers * (4 * d * d + 2 * d * ff)
    attn = lambda tok, layers: 4.0 * tok * tok * d * layers
    fwd = (lin(480, 2) + lin(960, 2) + lin(n, 24) + attn(480, 2) + attn(960, 2) + attn(n, 24)
           + 2.0 * (480 * 225 + 960 * 35 + n * 225) * d)
    step_flop = 3.0 * fwd * B * world
    out_line = {
        "metric": "motion frames/sec (train step) scaled FACT d=1536 x 24 cross layers",
        "value": round(world * B * 480 / dt, 1), "unit": "motion frames/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "scaled FACT train step (BASELINE.json configs[4])", "global_batch": world * B,
                   "per_gpu_batch": B, "motion_seq": 480, "audio_seq": 960, "hidden": d, "cross_layers": 24,
                   "params": nparams, "parallelism": "dp%d" % world,
                   "grad_allreduce": ("none (single replica)" if world == 1 else
                                      "%s buckets on a communication stream, overlapped with backward" % args.grad_buckets)},
        "step_tflops": round(step_flop / dt / 1e12, 1),
        "step_mfma_frac": round(step_flop / dt / 1e12 / (PEAK_BF16_TFLOPS * world), 4), "final_loss": round(loss, 5),
        "parity_vs_oracle": parity, "kernels": kern}
    if world > 1:
        import torch.distributed as dist
        out_line["comm"] = dict(comm or {}, process_group=comm_backend_record(dist, dry))
    if dry:
        out_line["dry_run"] = "gloo, ranks sharing devices: control-flow check of the N > 1 path, not a measurement"
    print(json.dumps(out_line))


def scaled_parity(model, device):
    """One FULL-DEPTH scaled-FACT step (2 + 2 + 24 layers, d = 1536, n = 1440: tiled attention kernels) at batch 1
    on the engine vs the fp32 PyTorch-CPU oracle with the same weights and inputs: loss and all gradient tensors."""
    from oracle import fact_oracle as O
    cfg = {"motion": {"seq_len": 480, "feature_dim": 225, "hidden": 1536, "layers": 2, "heads": 12, "ff": 6144},
           "audio": {"seq_len": 960, "feature_dim": 35, "hidden": 1536, "layers": 2, "heads": 12, "ff": 6144},
           "cross": {"hidden": 1536, "layers": 24, "heads": 12, "ff": 6144}, "out_dim": 225}
    batch = O.synthetic_batch(cfg, 1, TARGET_LEN, seed=0, dtype=torch.float32)
    gb = {k: v.float().to(device) for k, v in batch.items()}
    params = {n: v.detach().cpu().float().clone() for n, v in zip(model.variable_names, model.trainable_variables)}
    model.grad_arena.zero_()
    t0 = time.perf_counter()
    loss = float(model.forward_backward({k: v for k, v in gb.items() if k != "target"}, gb["target"]))
    torch.cuda.synchronize()
    grads = [g.detach().cpu().double().flatten() for g in model.gradients]
    model.grad_arena.zero_()
    torch.set_num_threads(min(usable_cores(), 64))
    ref_loss, ref_grads, _ = O.loss_and_grads(params, cfg, batch["motion_input"], batch["audio_input"], batch["target"])
    worst_cos, worst_rel, worst_name = 1.0, 0.0, ""
    for name, g in zip(model.variable_names, grads):
        r = ref_grads[name].double().flatten()
        c = float((g @ r) / (g.norm() * r.norm() + 1e-30))
        rl = float((g - r).norm() / (r.norm() + 1e-30))
        if c < worst_cos:
            worst_cos, worst_name = c, name
        worst_rel = max(worst_rel, rl)
    return {"batch": 1, "layers": "2+2+24", "loss_engine": round(loss, 6), "loss_oracle": round(float(ref_loss), 6),
            "loss_rel_diff": round(abs(loss - float(ref_loss)) / abs(float(ref_loss)), 6),
            "gradient_tensors": len(grads), "worst_gradient_cosine": round(worst_cos, 5),
            "worst_gradient_cosine_tensor": worst_name, "worst_gradient_rel_frobenius": round(worst_rel, 4),
            "oracle": "fp32 PyTorch-CPU restatement (oracle/fact_oracle.py)", "seconds": round(time.perf_counter() - t0, 1)}


def kernel_table_from_child(args, local_rank, B):
    """The `kernels` table of a production-library run: a child of this script on the SAME GPU, bound to libfact_hip_dbg.so
    (mint_amd/_lib.py binds one build per process), runs the same single-replica step - warm-up, a short timed region, then
    --profile-steps recorded steps - and prints the rows.  Returns its JSON or None (the headline line never depends on it)."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_PORT",
                        "TORCHELASTIC_RUN_ID", "FACT_DEBUG_ABI", "GPU_MAX_HW_QUEUES")}
    env["FACT_DEBUG_ABI"] = "1"
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    env["HIP_VISIBLE_DEVICES"] = vis.split(",")[local_rank] if vis else str(local_rank)
    env.pop("CUDA_VISIBLE_DEVICES", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--kernel-table-child", "--gpus", "1", "--steps", "5", "--warmup", "5",
           "--batch", str(B), "--profile-steps", str(max(1, args.profile_steps)), "--side-stream", str(args.side_stream),
           "--fuse-optimizer", str(args.fuse_optimizer), "--no-cpu-baseline"]
    for attempt in range(2):
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
            for line in reversed(r.stdout.strip().splitlines()):
                if line.startswith("{") and "kernel_table_child" in line:
                    return json.loads(line)
            print("kernel-table child failed (rc %d, attempt %d): %s" % (r.returncode, attempt, (r.stderr or r.stdout)[-2000:]),
                  file=sys.stderr)
        except Exception as e:
            print("kernel-table child failed (attempt %d): %r" % (attempt, e), file=sys.stderr)
    return None


def reference_profile(kernel):
    """rocprofv3's average for the dominant kernel's symbol from the newest COMMITTED `--kernel-trace --stats` summary of this
    command (profiles/rNN_kernel_stats_bench_n1.txt) - an EARLIER run, kept apart from this run's own clocks (round-5
    advisor: it read as a second clock of the same run)."""
    prof, src = rocprof_symbol_table()
    first = kernel.split(" + ")[0]
    hit = [v for k, v in prof.items() if k.startswith(first) or first.startswith(k)]  # (the summary truncates long symbols)
    if len(hit) != 1:
        return None
    return {"file": "profiles/" + src, "rocprofv3_avg_us_same_symbol": hit[0],
            "note": "committed profile of an earlier run of the same command, NOT this run; compare with avg_launch_us "
                    "only when the kernel has not changed since that file was committed"}


def comm_backend_record(dist, dry):
    """What the live process group is (round-5 review item 8): backend, world size and - for nccl = RCCL - the library version."""
    rec = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    try:
        if rec["backend"] == "nccl":
            v = torch.cuda.nccl.version()
            rec["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:
        rec["rccl_version"] = "unavailable: %r" % (e,)
    if dry:
        rec["note"] = "gloo dry run through the host"
    return rec


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python bench.py --gpus N does it "
                         "itself; under torch.distributed.run pass the same N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    dry = args.dist_backend == "gloo"
    if dry:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    if args.mode in ("ar", "scaled"):
        (run_ar if args.mode == "ar" else run_scaled)(args, device, world, rank, dry)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    from mint_amd import configs, model_builder
    from mint_amd.learning_schedules import create_learning_rate
    from mint_amd.trainer import Adam

    pipe = configs.fact_v5_deeper_t10_cm12()
    model = model_builder.build(pipe.multi_modal_model, True)
    B = args.batch or BATCH_PER_GPU
    gen = torch.Generator().manual_seed(1234 + rank)
    batch = {"motion_input": torch.randn(B, 120, 225, generator=gen).to(device),
             "audio_input": torch.randn(B, 240, 35, generator=gen).to(device),
             "target": torch.randn(B, TARGET_LEN, 225, generator=gen).to(device)}
    model.build(B, 225, 35)
    model.set_option("side_stream", args.side_stream)
    if world > 1 and not args.dp_aux_stream:
        model.set_option("aux_stream", 0)
    apply_opts(model, args)
    opt = Adam(create_learning_rate(pipe.train_config.learning_rate))
    trainer, it = make_trainer(model, batch, opt, args, world)
    dt, final_loss = timed_train_steps(trainer, it, args, world, device, dry)

    if args.breakdown and rank == 0:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        tgt = batch["target"]
        inp = {k: v for k, v in batch.items() if k != "target"}
        s = torch.cuda.current_stream()
        ev[0].record(s)
        for _ in range(5):
            model(inp)
        ev[1].record(s)
        for _ in range(5):
            model.forward_backward(inp, tgt)
        ev[2].record(s)
        for _ in range(5):
            model.apply_adam(1e-4)
        ev[3].record(s)
        torch.cuda.synchronize()
        print("breakdown ms: forward %.3f  forward+backward %.3f  adam+shadow %.3f" % (
            ev[0].elapsed_time(ev[1]) / 5, ev[1].elapsed_time(ev[2]) / 5, ev[2].elapsed_time(ev[3]) / 5),
            file=sys.stderr)

    from mint_amd import _lib as L
    has_recorder = L.DEBUG_ABI and hasattr(L.lib(), "fact_kprof")
    if args.kernel_table_child:
        # child of a production-library run: the table of the same single-replica step, on the test / bench build
        assert has_recorder and world == 1, "--kernel-table-child binds libfact_hip_dbg.so at N = 1"
        rows, ksum_ms = kernel_table(model, lambda: trainer.train_step(it), max(1, args.profile_steps))
        child = {"kernel_table_child": True, "rows": rows, "ksum_ms": ksum_ms, "library": os.path.basename(L.LIB_PATH),
                 "ms_per_step": round(dt / args.steps * 1e3, 3)}
        if rows and rows[0]["name"] == "wgrad_group":
            try:
                child["standalone"] = wgrad_standalone(device)
            except Exception as e:
                child["standalone"] = {"error": repr(e)[:200]}
        print(json.dumps(child))
        return
    comm = comm_record(trainer, it, args, world)
    rows, ksum_ms, standalone, table_from = None, 0.0, None, None
    if has_recorder:  # --opt / FACT_DEBUG_ABI=1: the whole run is on the test / bench build
        rows, ksum_ms = kernel_table(model, lambda: trainer.train_step(it), max(1, args.profile_steps))
        table_from = "this process (%s)" % os.path.basename(L.LIB_PATH)
    elif rank == 0 and args.profile_steps > 0:
        torch.cuda.synchronize()
        child = kernel_table_from_child(args, local_rank, B)
        if child is not None:
            rows, ksum_ms, standalone = child["rows"], child["ksum_ms"], child.get("standalone")
            table_from = ("child process on the same GPU bound to %s (the production objects + the recorder of "
                          "include/fact_hip_debug.h), single-replica step, %.3f ms/step there" % (child["library"], child["ms_per_step"]))
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        frames_per_s = world * B * 120 / (dt / args.steps)
        out = {
            "metric": "motion frames/sec (train step) fact_v5_deeper_t10_cm12",
            "value": round(frames_per_s, 1), "unit": "motion frames/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "fact_v5_deeper_t10_cm12 train step (BASELINE.json configs[%d])"
                                   % (1 if world == 1 else 2),
                       "global_batch": world * B, "per_gpu_batch": B, "motion_seq": 120, "audio_seq": 240,
                       "target_frames": TARGET_LEN, "parallelism": "dp%d" % world, "params": 120406977,
                       "grad_allreduce": ("none (single replica)" if world == 1 else
                                          "%s buckets on a communication stream, overlapped with backward" % args.grad_buckets)},
            "samples_per_sec": round(frames_per_s / 120, 2),
            # MFMA work actually executed (the supervised-rows shortcut skips ~5 % of the reference step's FLOPs)
            "executed_flop_fraction": round(executed_flop_fraction(), 4),
            "step_tflops": round(frames_per_s * FLOP_PER_FRAME * executed_flop_fraction() / 1e12, 1),
            "step_mfma_frac": round(frames_per_s * FLOP_PER_FRAME * executed_flop_fraction() / 1e12
                                    / (PEAK_BF16_TFLOPS * world), 4),
            "final_loss": round(final_loss, 5),
            "library": os.path.basename(L.LIB_PATH),  # what ran the timed K steps
        }
        if world > 1:  # the live process group rides in the comm record (backend, world, RCCL version)
            out["comm"] = dict(comm or {}, process_group=comm_backend_record(dist, dry))
        if dry:
            out["dry_run"] = "gloo, ranks sharing devices: control-flow check of the N > 1 path, not a measurement"
        if rows:
            out["kernels"] = rows
            out["kernels_from"] = table_from
            top = rows[0]
            traffic, src = measured_traffic(top["name"])
            out["roofline"] = {"bound": top["bound"], "kernel": top["kernel"], "class": top["name"],
                               "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
                               "avg_launch_us": top["avg_launch_us"], "launches_per_step": top["launches_per_step"],
                               "time_share": top["time_share"],
                               "flop_per_launch": top.get("flop_per_launch"), "cu_share": top.get("cu_share"),
                               "frac_of_held_cus": top.get("frac_of_held_cus"), "traffic": traffic, "traffic_source": src,
                               "clock": "engine event recorder (HIP events on the launch stream around every launch of the "
                                        "class, inside normal train steps, all streams overlapping; ~2-3 us per launch above "
                                        "the rocprofv3 kernel duration of the same dispatch)",
                               "how": "sum of kernel-class time per step %.2f ms vs %.2f ms wall; table from: %s" % (
                                   ksum_ms, ms_per_step, table_from)}
            ref = reference_profile(top["kernel"])
            if ref:
                out["roofline"]["reference_profile"] = ref
            if standalone is None and has_recorder and top["name"] == "wgrad_group":
                try:
                    standalone = wgrad_standalone(device)
                except Exception as e:  # never lose the headline line to the extra measurement
                    standalone = {"error": repr(e)[:200]}
            if standalone is not None:
                out["roofline"]["standalone"] = standalone
            by = {r["name"]: r for r in rows}
            if "attention_fwd" in by and "attention_bwd" in by:
                out["attention"] = {
                    "fwd_tflops": by["attention_fwd"]["achieved"], "fwd_mfma_frac": by["attention_fwd"]["frac"],
                    "bwd_tflops": by["attention_bwd"]["achieved"], "bwd_mfma_frac": by["attention_bwd"]["frac"],
                    "note": "QK^T+PV algorithmic FLOPs / in-step launch time, all 16 layers; MFMA-busy PMC: profiles/"}
        elif args.profile_steps > 0:
            # the kernel-table child did not deliver (it never should fail; the headline must not depend on it): a whole-step
            # roofline object from this process's own clock, marked as such
            out["roofline"] = {"bound": "mfma", "kernel": "whole train step (kernel-class table unavailable: child process failed)",
                               "achieved": out["step_tflops"], "peak": PEAK_BF16_TFLOPS * world, "unit": "TFLOP/s",
                               "frac": out["step_mfma_frac"], "traffic": None}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
