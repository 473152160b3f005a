"""Cost of the data-parallel step structure on ONE GPU (nccl, world size 1: the collectives are local): bucket
callbacks, bf16 casts, optimizer placement (one Adam pass after the last all-reduce vs every bucket's Adam right behind
its all-reduce on the communication stream) and the number of streams in the process.  Every configuration runs in a
fresh process: which streams end up sharing one of HIP's 4 hardware queues depends on creation order."""
import os, subprocess, sys, time

CONFIGS = {
    "single replica, no process group": dict(dist=False, kw=dict(overlap_grad_allreduce=False)),
    "single replica, RCCL initialised": dict(dist=True, kw=dict(overlap_grad_allreduce=False)),
    "reducer fp32, Adam after the last all-reduce": dict(dist=True, kw=dict(overlap_grad_allreduce=True, dp_fused_adam=False)),
    "reducer bf16, Adam after the last all-reduce": dict(dist=True, kw=dict(overlap_grad_allreduce=True, bf16_grad_buckets=True, dp_fused_adam=False)),
    "reducer fp32, Adam behind each all-reduce": dict(dist=True, kw=dict(overlap_grad_allreduce=True, dp_fused_adam="force")),
    "reducer bf16, Adam behind each all-reduce": dict(dist=True, kw=dict(overlap_grad_allreduce=True, bf16_grad_buckets=True, dp_fused_adam="force")),
    "reducer bf16, Adam behind each, third engine stream kept": dict(dist=True, aux=1, kw=dict(overlap_grad_allreduce=True, bf16_grad_buckets=True, dp_fused_adam="force")),
}


def child(name):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from mint_amd import configs, model_builder
    from mint_amd.trainer import Adam, SingleTaskTrainer
    c = CONFIGS[name]
    if c["dist"]:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pipe = configs.fact_v5_deeper_t10_cm12()
    B = 16
    gen = torch.Generator().manual_seed(0)
    batch = {"motion_input": torch.randn(B, 120, 225, generator=gen).cuda(), "audio_input": torch.randn(B, 240, 35, generator=gen).cuda(),
             "target": torch.randn(B, 20, 225, generator=gen).cuda()}

    class Rep:
        def __iter__(self): return self
        def __next__(self): return batch
    model = model_builder.build(pipe.multi_modal_model, True)
    model.build(B, 225, 35)
    tr = SingleTaskTrainer(Rep(), "target", model, optimizer=Adam(1e-4), **c["kw"])
    it = iter(Rep())
    tr.train_step(it)
    if c.get("aux") is not None:
        model.set_option("aux_stream", c["aux"])
    for _ in range(6): tr.train_step(it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tr.train_step(it)
    torch.cuda.synchronize()
    print("%-58s %.3f ms/step" % (name, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
    if c["dist"]:
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for name in CONFIGS:
            subprocess.run([sys.executable, os.path.abspath(__file__), name], stderr=subprocess.DEVNULL)
