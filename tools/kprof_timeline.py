"""In-step timeline from the engine's own event recorder (no rocprof overhead): one profiled train step."""
import os as _os; _os.environ.setdefault("FACT_DEBUG_ABI", "1")  # these tools drive the test / bench surface (mint_amd/_lib.py)
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import configs, model_builder, _lib as L
from mint_amd.trainer import Adam, SingleTaskTrainer
pipe = configs.fact_v5_deeper_t10_cm12()
model = model_builder.build(pipe.multi_modal_model, True)
B = 16
gen = torch.Generator().manual_seed(0)
batch = {"motion_input": torch.randn(B, 120, 225, generator=gen).cuda(), "audio_input": torch.randn(B, 240, 35, generator=gen).cuda(),
         "target": torch.randn(B, 20, 225, generator=gen).cuda()}
model.build(B, 225, 35)
for kv in sys.argv[2:]:
    k, v = kv.split("="); model.debug_option(k, int(v))
class Rep:
    def __iter__(self): return self
    def __next__(self): return batch
tr = SingleTaskTrainer(Rep(), "target", model, optimizer=Adam(1e-4))
it = iter(Rep())
for _ in range(6): tr.train_step(it)
model.kernel_profile(True)
for _ in range(3): tr.train_step(it)   # host runs ahead; the LAST step is the one to read
torch.cuda.synchronize()
L.check(L.lib().fact_kprof_dump(model._h, sys.argv[1].encode()))
rows = [l.strip().split(",") for l in open(sys.argv[1])]
rows = [(r[0], int(r[1]), float(r[2]), float(r[3])) for r in rows]
# last step = records after the last 'ln_fwd' burst start: split by count
n = len(rows) // 3
last = rows[2 * n:]
t0 = min(r[2] for r in last)
print("records per step", n, " step span %.0f us" % (max(r[3] for r in last) - t0))
for r in sorted(last, key=lambda r: r[2]):
    print("s%d %8.1f -> %8.1f (%6.1f)  %s" % (r[1], r[2] - t0, r[3] - t0, r[3] - r[2], r[0]))
