"""Headline benchmark: motion frames/sec of the FACT train step (fact_v5_deeper_t10_cm12, bf16
MFMA compute / fp32 master weights, batch 16 per GPU, AIST++-shaped synthetic tensors).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one batch: forward, MSE loss on 20 target frames, backward,
RCCL gradient all-reduce (N>1), Keras-Adam update, bf16 weight-shadow refresh.  Prints ONE JSON line
(rank 0).  `roofline` is for the dominant kernel (the bf16 MFMA GEMM, timed with HIP events on its
own stream at the cross-modal FFN shape); `cpu_baseline` is the oracle (PyTorch-CPU fp32 restatement
of the reference train step) timed on this box's host cores at N=1.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_FRAME = 2.024e9      # 242.92 GFLOP / sample / 120 frames
PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
BATCH_PER_GPU = 16
TARGET_LEN = 20


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="per-GPU batch (headline: 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-phase timing to stderr")
    ap.add_argument("--side-stream", type=int, default=1, help="0 = single-stream engine (A/B knob)")
    ap.add_argument("--fuse-optimizer", type=int, default=1, help="0 = Adam as a separate pass after backward (A/B knob)")
    return ap.parse_args()


def gemm_roofline(device):
    """Dominant kernel: the bf16 MFMA NT GEMM with the bias+GELU epilogue (gemm_nt_big_kernel<BIAS_GELU>,
    288x256 tiles) at the cross-modal FFN1 shape M=5760 (16*360 tokens), N=3072, K=800, operands at the
    engine's 832-element row pitch.  Algorithmic FLOPs per launch 2*M*N*K = 28.31 GFLOP; timed with HIP
    events on the stream the kernel is launched on."""
    from mint_amd import _lib as L
    lib = L.lib()
    M, N, K = BATCH_PER_GPU * 360, 3072, 800
    g = torch.Generator(device=device).manual_seed(0)
    LD = 832  # bf16 row pitch used by the engine for 800-wide activations / weight shadows
    A = torch.randn(M, LD, device=device, generator=g).to(torch.bfloat16)
    B = (torch.randn(N, LD, device=device, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.zeros(N, device=device)
    pre = torch.empty(M, N, device=device, dtype=torch.bfloat16)
    act = torch.empty(M, N, device=device, dtype=torch.bfloat16)
    stream = torch.cuda.current_stream()

    def launch():
        L.check(lib.fact_op_gemm_nt(L.EPI_BIAS_GELU, L.ptr(A), LD, L.ptr(B), LD, M, N, K, L.ptr(pre), N,
                                    L.ptr(act), N, L.ptr(bias), None, 0, None, 0, None, 0,
                                    L.cur_stream()))
    for _ in range(5):
        launch()
    iters = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        launch()
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * M * N * K
    achieved = flops / (ms * 1e-3) / 1e12
    traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/roofline_traffic.json)
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json")))["traffic_bytes"]
    except Exception:
        pass
    return {"bound": "mfma", "kernel": "gemm_nt_big_kernel<EPI_BIAS_GELU, 9> (288x256 tiles) M5760 N3072 K800",
            "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "avg_launch_us": round(ms * 1e3, 2),
            "flop_per_launch": flops, "traffic": traffic}


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


def cpu_baseline(budget_s=25.0):
    """Oracle (PyTorch-CPU fp32 restatement of the reference train step) on a bounded sample of the
    same workload: fact_v5 at batch 1.  The forward pass is timed first; the full step (forward,
    backward, Adam) is only run if it fits the time budget, otherwise the step time is the measured
    forward time x 3 (backward = 2 x forward FLOPs) and the sample says so."""
    from oracle import fact_oracle as O
    cores = min(usable_cores(), 64)
    torch.set_num_threads(cores)
    cfg = O.FACT_V5_CFG
    B = 1
    params = O.init_params(cfg, seed=0, dtype=torch.float32)
    batch = O.synthetic_batch(cfg, B, TARGET_LEN, seed=0, dtype=torch.float32)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.fact_forward(params, cfg, batch["motion_input"], batch["audio_input"])
        t_fwd = time.perf_counter() - t0
    if 3.5 * t_fwd <= budget_s:
        m = {k: torch.zeros_like(v) for k, v in params.items()}
        v = {k: torch.zeros_like(x) for k, x in params.items()}
        t0 = time.perf_counter()
        O.train_step(params, m, v, 0, cfg, batch, 1e-4)
        dt = time.perf_counter() - t0
        sample = "1 full train step of fact_v5 at batch 1, %.1f s" % dt
    else:
        dt = 3.0 * t_fwd
        sample = "forward pass of fact_v5 at batch 1 (%.1f s) x 3 (bwd = 2 x fwd FLOPs); full step over budget" % t_fwd
    return {"value": round(B * 120 / dt, 2), "unit": "motion frames/sec", "cores": cores, "kind": "port",
            "sample": sample + " (fp32 PyTorch-CPU oracle, %d threads)" % cores}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from mint_amd import configs, model_builder
    from mint_amd.learning_schedules import create_learning_rate
    from mint_amd.trainer import Adam, SingleTaskTrainer

    pipe = configs.fact_v5_deeper_t10_cm12()
    model = model_builder.build(pipe.multi_modal_model, True)
    B = args.batch
    gen = torch.Generator().manual_seed(1234 + rank)
    batch = {"motion_input": torch.randn(B, 120, 225, generator=gen).to(device),
             "audio_input": torch.randn(B, 240, 35, generator=gen).to(device),
             "target": torch.randn(B, TARGET_LEN, 225, generator=gen).to(device)}
    model.build(B, 225, 35)
    model.set_option("side_stream", args.side_stream)
    opt = Adam(create_learning_rate(pipe.train_config.learning_rate))

    class Repeat:
        def __iter__(self):
            return self

        def __next__(self):
            return batch

    trainer = SingleTaskTrainer(Repeat(), "target", model, optimizer=opt, fuse_optimizer=bool(args.fuse_optimizer))
    it = iter(Repeat())

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    trainer.train_loop_begin()
    for _ in range(args.warmup):
        trainer.train_step(it)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.train_step(it)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss)

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
                       "target_frames": TARGET_LEN, "parallelism": "dp%d" % world, "params": 120406977},
            "samples_per_sec": round(frames_per_s / 120, 2),
            "step_tflops": round(frames_per_s * FLOP_PER_FRAME / 1e12, 1),
            "step_mfma_frac": round(frames_per_s * FLOP_PER_FRAME / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
            "final_loss": round(final_loss, 5),
        }
        out["roofline"] = gemm_roofline(device)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
