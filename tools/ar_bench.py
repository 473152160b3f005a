"""Auto-regressive generation throughput (BASELINE.json configs[3] shape per GPU: 32 sequences)."""
import os as _os; _os.environ.setdefault("FACT_DEBUG_ABI", "1")  # these tools drive the test / bench surface (mint_amd/_lib.py)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import configs, model_builder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
model = model_builder.build(configs.fact_v5_deeper_t10_cm12().multi_modal_model, False)
g = torch.Generator().manual_seed(0)
inp = {"motion_input": torch.randn(B, 120, 225, generator=g).cuda(),
       "audio_input": torch.randn(B, 240 + steps - 1, 35, generator=g).cuda()}
model.infer_auto_regressive(inp, steps=3)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = model.infer_auto_regressive(inp, steps=steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert out.shape == (B, steps, 225) and torch.isfinite(out).all()
print("AR generation: B=%d steps=%d  %.2f ms/step  %.0f generated frames/s  (fwd %.1f TFLOP/s)" % (
    B, steps, dt / steps * 1e3, B * steps / dt, B * 80.97e9 * steps / dt / 1e12))
