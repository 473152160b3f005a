"""Dependent-launch boundary of the engine's own kernels, measured where it matters: the single-stream forward of the
headline configuration (no co-runner).  Run twice:
  python tools/fwd_boundary.py                       -> wall time per forward (HIP events, no profiler attached)
  rocprofv3 --kernel-trace --stats -d DIR -- python tools/fwd_boundary.py   -> summed kernel durations per forward
boundary = (wall - sum of kernel durations) / (launches - 1).  A kernel trace itself cannot give it: with the profiler
attached every dispatch is followed by ~10 us of idle queue (profiles/r05_launch_boundary.txt, first part)."""
import os as _os; _os.environ.setdefault("FACT_DEBUG_ABI", "1")
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import configs, model_builder

N = int(os.environ.get("ITERS", "20"))
pipe = configs.fact_v5_deeper_t10_cm12()
model = model_builder.build(pipe.multi_modal_model, True)
g = torch.Generator().manual_seed(3)
inp = {"motion_input": torch.randn(16, 120, 225, generator=g).cuda(), "audio_input": torch.randn(16, 240, 35, generator=g).cuda()}
model.build(16, 225, 35)
model.set_option("side_stream", int(os.environ.get("SIDE", "0")))  # 0: both encoders on the caller's stream too - ONE chain
for _ in range(3):
    model(inp)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(N):
    model(inp)
b.record()
torch.cuda.synchronize()
print("FWD_WALL_US %.2f per forward over %d forwards (side_stream=%s)" % (a.elapsed_time(b) * 1e3 / N, N, os.environ.get("SIDE", "0")))
