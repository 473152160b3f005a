import os, sys
os.environ.setdefault("FACT_DEBUG_ABI","1")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mint_amd import configs, model_builder
from mint_amd.trainer import Adam, SingleTaskTrainer
from oracle import fact_oracle as O
cfg = O.FACT_V5_CFG
batch = {k: v.float().cuda() for k, v in O.synthetic_batch(cfg, 16, 20, seed=21, dtype=torch.float32).items()}
def run(steps, opts=()):
    model = model_builder.build(configs.fact_v5_deeper_t10_cm12().multi_modal_model, True)
    model.build(16, 225, 35)
    for k, v in opts: model.debug_option(k, v)
    p0 = torch.cat([v.flatten() for v in model.trainable_variables]).double().sum().item()
    tr = SingleTaskTrainer([batch] * steps, "target", model, optimizer=Adam(1e-4))
    tr.train_loop_begin()
    it = iter([batch] * steps)
    losses = torch.stack([tr.train_step(it).detach().float().reshape(()) for _ in range(steps)])
    torch.cuda.synchronize()
    return p0, losses.cpu().double()
for name, opts in [("default", ()), ("default", ()), ("side_stream0", (("side_stream", 0),)), ("side_stream0", (("side_stream", 0),))]:
    p0, a = run(240, opts)
    print(name, "init checksum %.6f" % p0, "finite", bool(torch.isfinite(a).all()), " ".join("%.4f" % x for x in a[::12].tolist()), "max", float(a.max()), "argmax", int(a.argmax()))
