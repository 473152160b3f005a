import sys, os
sys.path.insert(0, os.getcwd())
import torch
from mint_amd import model_builder, _lib as L
from oracle import fact_oracle as O
sys.path.insert(0, "tests")
from test_gpu_model import make_config, oracle_params, rel
cfg = O.TINY_CFG
g = torch.Generator().manual_seed(5)
motion = torch.randn(2, 32, 225, generator=g, dtype=torch.float64)
audio = torch.randn(2, 64 + 5, 35, generator=g, dtype=torch.float64)
for side, tiled in ((1, 0), (0, 0), (1, 1), (0, 1)):
    L.lib().fact_debug_attn_force_tiled(tiled)
    model = model_builder.build(make_config(cfg), False)
    model.build(2, 225, 35)
    model.set_option("side_stream", side)
    out = model.infer_auto_regressive({"motion_input": motion.float().cuda(), "audio_input": audio.float().cuda()}, steps=10)
    params = oracle_params(model)
    ref = O.infer_auto_regressive(params, cfg, motion, audio, steps=10)
    per = [rel(out[:, i], ref[:, i]) for i in range(out.shape[1])]
    fwd = model({"motion_input": motion.float().cuda(), "audio_input": audio[:, :64].float().cuda()})
    reff = O.fact_forward(params, cfg, motion, audio[:, :64])
    print("side", side, "tiled", tiled, "AR rel per step", ["%.3f" % p for p in per], "plain fwd rel %.4f" % rel(fwd, reff))
