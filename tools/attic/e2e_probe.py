"""Where does the input-fed train step lose time? (see tools/e2e_input_bench.py)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import configs, model_builder
from mint_amd.learning_schedules import create_learning_rate
from mint_amd.trainer import Adam, SingleTaskTrainer
dev = torch.device("cuda", 0)
pipe = configs.fact_v5_deeper_t10_cm12()
model = model_builder.build(pipe.multi_modal_model, True)
model.build(16, 225, 35)
g = torch.Generator().manual_seed(0)
host = [{"motion_input": torch.randn(16, 120, 225, generator=g), "audio_input": torch.randn(16, 240, 35, generator=g),
         "target": torch.randn(16, 20, 225, generator=g)} for _ in range(8)]
devb = [{k: v.to(dev) for k, v in b.items()} for b in host]
pinned = [{k: v.pin_memory() for k, v in b.items()} for b in host]


def loop(name, nxt, steps=60):
    class DS:
        def __iter__(self): return self
        def __next__(self): return nxt()
    tr = SingleTaskTrainer(DS(), "target", model, optimizer=Adam(1e-4))
    it = iter(DS())
    for _ in range(5): tr.train_step(it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): tr.train_step(it)
    torch.cuda.synchronize()
    print("%-62s %.3f ms/step" % (name, (time.perf_counter() - t0) / steps * 1e3), flush=True)

i = [0]
def rot(lst):
    i[0] += 1
    return lst[i[0] % len(lst)]
loop("one resident device batch", lambda: devb[0])
loop("8 resident device batches in rotation", lambda: rot(devb))
loop("pageable host batch .to(dev) per step", lambda: {k: v.to(dev) for k, v in rot(host).items()})
loop("pre-pinned host batch .to(dev, non_blocking) per step", lambda: {k: v.to(dev, non_blocking=True) for k, v in rot(pinned).items()})
loop("pin_memory() + .to(dev, non_blocking) per step", lambda: {k: v.pin_memory().to(dev, non_blocking=True) for k, v in rot(host).items()})
cs = torch.cuda.Stream()
def on_copy_stream():
    with torch.cuda.stream(cs):
        b = {k: v.to(dev, non_blocking=True) for k, v in rot(pinned).items()}
        ev = torch.cuda.Event(); ev.record(cs)
    torch.cuda.current_stream().wait_event(ev)
    return b
loop("pre-pinned, copy on its own stream + event", on_copy_stream)
t0 = time.perf_counter()
for _ in range(50):
    x = {k: v.pin_memory() for k, v in host[0].items()}
print("pin_memory() of one batch: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
