"""Host enqueue time of one train step (no device sync inside the loop) vs device time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import configs, model_builder
from mint_amd.learning_schedules import create_learning_rate
from mint_amd.trainer import Adam, SingleTaskTrainer

pipe = configs.fact_v5_deeper_t10_cm12()
model = model_builder.build(pipe.multi_modal_model, True)
B = 16
gen = torch.Generator().manual_seed(0)
batch = {"motion_input": torch.randn(B, 120, 225, generator=gen).cuda(), "audio_input": torch.randn(B, 240, 35, generator=gen).cuda(),
         "target": torch.randn(B, 20, 225, generator=gen).cuda()}
model.build(B, 225, 35)
class Rep:
    def __iter__(self): return self
    def __next__(self): return batch
tr = SingleTaskTrainer(Rep(), "target", model, optimizer=Adam(1e-4))
it = iter(Rep())
for _ in range(5): tr.train_step(it)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
hs = []
for _ in range(N):
    a = time.perf_counter(); tr.train_step(it); hs.append(time.perf_counter() - a)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue per step: mean %.3f ms (min %.3f max %.3f); wall per step incl. device %.3f ms" % (
    t_host / N * 1e3, min(hs) * 1e3, max(hs) * 1e3, t_all / N * 1e3))
# pure C call
inp = {k: v for k, v in batch.items() if k != "target"}
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): model.forward_backward(inp, batch["target"])
th = time.perf_counter() - t0
torch.cuda.synchronize()
print("forward_backward only: host %.3f ms per call, wall %.3f" % (th / N * 1e3, (time.perf_counter() - t0) / N * 1e3))
