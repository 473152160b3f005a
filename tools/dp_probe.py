"""Cost of the data-parallel step structure on ONE GPU (nccl, world size 1: the collectives are local copies): the
bucket callbacks, the bf16 casts and the optimizer placement - one Adam pass after the last all-reduce vs every
bucket's Adam right behind its all-reduce on the communication stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from mint_amd import configs, model_builder
from mint_amd.trainer import Adam, SingleTaskTrainer

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
if os.environ.get("NO_DIST") != "1":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pipe = configs.fact_v5_deeper_t10_cm12()
B = 16
gen = torch.Generator().manual_seed(0)
batch = {"motion_input": torch.randn(B, 120, 225, generator=gen).cuda(), "audio_input": torch.randn(B, 240, 35, generator=gen).cuda(),
         "target": torch.randn(B, 20, 225, generator=gen).cuda()}
class Rep:
    def __iter__(self): return self
    def __next__(self): return batch
ALL = (("single replica (in-backward Adam, no reducer)", dict(overlap_grad_allreduce=False)),
                 ("reducer fp32, Adam after the last all-reduce", dict(overlap_grad_allreduce=True, dp_fused_adam=False)),
                 ("reducer bf16, Adam after the last all-reduce", dict(overlap_grad_allreduce=True, bf16_grad_buckets=True, dp_fused_adam=False)),
                 ("reducer fp32, Adam behind each all-reduce", dict(overlap_grad_allreduce=True, dp_fused_adam="force")),
                 ("reducer bf16, Adam behind each all-reduce", dict(overlap_grad_allreduce=True, bf16_grad_buckets=True, dp_fused_adam="force")))
for name, kw in (ALL[:1] if os.environ.get("NO_DIST") == "1" else ALL):
    model = model_builder.build(pipe.multi_modal_model, True)
    model.build(B, 225, 35)
    tr = SingleTaskTrainer(Rep(), "target", model, optimizer=Adam(1e-4), **kw)
    it = iter(Rep())
    for _ in range(6): tr.train_step(it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tr.train_step(it)
    torch.cuda.synchronize()
    print("%-50s %.3f ms/step" % (name, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
    del tr, model
if dist.is_initialized():
    dist.destroy_process_group()
