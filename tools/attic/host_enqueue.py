"""How long does the HOST take to enqueue one train step (no GPU sync inside)?  Compare with the GPU step time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import configs, model_builder

pipe = configs.fact_v5_deeper_t10_cm12()
model = model_builder.build(pipe.multi_modal_model, True)
B = 16
g = torch.Generator().manual_seed(0)
batch = {"motion_input": torch.randn(B, 120, 225, generator=g).cuda(), "audio_input": torch.randn(B, 240, 35, generator=g).cuda()}
tgt = torch.randn(B, 20, 225, generator=g).cuda()
model.build(B, 225, 35)
for _ in range(3):
    model.forward_backward(batch, tgt); model.apply_adam(1e-4)
torch.cuda.synchronize()
for name, fn in (("forward_backward", lambda: model.forward_backward(batch, tgt)), ("apply_adam", lambda: model.apply_adam(1e-4)),
                 ("forward", lambda: model(batch))):
    host, total = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); total.append(t2 - t0)
    print("%-18s host enqueue %.3f ms   until GPU done %.3f ms" % (name, 1e3 * sorted(host)[5], 1e3 * sorted(total)[5]))
