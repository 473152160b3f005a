"""De-risking the 8-GPU run on one GPU: the data-parallel step with a STAND-IN for the RCCL
all-reduce of every gradient bucket - a kernel that holds `ch` CUs (one 256-thread workgroup each, like a collective's
channel workgroups) for `us` microseconds on the communication stream, launched from the engine's bucket callback exactly
where the reducer launches the all-reduce.  World size 1, no torch.distributed.  Reports ms/step for (ch, us) pairs."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import _lib as L
from mint_amd import configs, model_builder
from mint_amd.trainer import Adam, SingleTaskTrainer


def main():
    torch.cuda.set_device(0)
    lib = L.lib()
    pipe = configs.fact_v5_deeper_t10_cm12()
    model = model_builder.build(pipe.multi_modal_model, True)
    B = 16
    g = torch.Generator().manual_seed(1)
    batch = {"motion_input": torch.randn(B, 120, 225, generator=g).cuda(), "audio_input": torch.randn(B, 240, 35, generator=g).cuda(),
             "target": torch.randn(B, 20, 225, generator=g).cuda()}
    model.build(B, 225, 35)
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        model.set_option(k, int(v))
    comm = torch.cuda.Stream()
    cfg = {"ch": 0, "us": 0}

    def on_bucket(bucket, off, cnt):
        if cfg["ch"] > 0:
            # duration scaled with the bucket size (the head bucket is tiny)
            us = max(5, int(cfg["us"] * cnt / 7.5e6))
            L.check(lib.fact_debug_cu_hog(cfg["ch"], us, C.c_void_p(comm.cuda_stream)))
        model.adam_bucket(bucket, comm)   # what the reducer does behind the all-reduce (dp_fused_adam)

    class Rep:
        def __iter__(self):
            return self

        def __next__(self):
            return batch
    model.set_grad_callback(on_bucket, comm)
    opt = Adam(1e-4)
    inp = {k: v for k, v in batch.items() if k != "target"}

    def step():
        opt.begin_fused(model)
        model.forward_backward(inp, batch["target"])
        torch.cuda.current_stream().wait_stream(comm)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    res = {}
    combos = [(0, 0), (8, 50), (8, 150), (16, 150), (32, 150), (16, 400), (64, 150)]
    for rnd in range(2):
        for ch, us in combos:
            cfg["ch"], cfg["us"] = ch, us
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                step()
            e1.record()
            e1.synchronize()
            res.setdefault((ch, us), []).append(e0.elapsed_time(e1) / 10)
    print("comm_standin %s queues=%s: " % (" ".join(sys.argv[1:]), os.environ.get("GPU_MAX_HW_QUEUES", "dflt"))
          + "  ".join("ch%d/%dus %s" % (k[0], k[1], "/".join("%.2f" % x for x in v)) for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
