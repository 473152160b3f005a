"""Does replaying the whole train step as a HIP graph beat stream launches?  (timing probe: the captured optimizer
hyper-parameters are frozen, so this is NOT a training loop)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import configs, model_builder
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
for _ in range(8): tr.train_step(it)
torch.cuda.synchronize()
def timed(fn, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("stream launches: %.3f ms/step" % timed(lambda: tr.train_step(it)))
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        tr.train_step(it)
    for _ in range(3): g.replay()
    print("graph replay:    %.3f ms/step" % timed(g.replay))
except Exception as e:
    print("capture failed:", repr(e)[:300])
