"""What do the two encoder stacks cost inside the train step?  Times the fused train step (same loop as bench.py) for
layer-count variants of fact_v5 at B = 16: (enc layers, cross layers) in {(2,12), (0,12), (2,10), (1,12)}.
encoder cost = t(2,12) - t(0,12);  two cross layers = t(2,12) - t(2,10).  The encoders do 2 cross-layer equivalents of
GEMM work (1920 + 3840 = 5760 rows per layer pair)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import configs, model_builder
from mint_amd.trainer import Adam, SingleTaskTrainer


def run(enc, cross, B=16, steps=30, warm=6, opts=()):
    mm = configs.fact_config(motion=(120, 225, 800, enc, 10, 3072), audio=(240, 35, 800, enc, 10, 3072),
                             cross=(800, cross, 10, 3072))
    model = model_builder.build(mm, True)
    g = torch.Generator().manual_seed(1)
    batch = {"motion_input": torch.randn(B, 120, 225, generator=g).cuda(),
             "audio_input": torch.randn(B, 240, 35, generator=g).cuda(),
             "target": torch.randn(B, 20, 225, generator=g).cuda()}
    model.build(B, 225, 35)
    for kv in opts:
        k, v = kv.split("=")
        model.set_option(k, int(v))

    class Rep:
        def __iter__(self):
            return self

        def __next__(self):
            return batch
    tr = SingleTaskTrainer(Rep(), "target", model, optimizer=Adam(1e-4))
    it = iter(Rep())
    for _ in range(warm):
        tr.train_step(it)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        tr.train_step(it)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / steps
    del tr, model
    torch.cuda.empty_cache()
    return ms


if __name__ == "__main__":
    opts = sys.argv[1:]
    res = {}
    for rnd in range(2):
        for cfg in ((2, 12), (0, 12), (2, 10), (1, 12), (2, 0)):
            res.setdefault(cfg, []).append(run(*cfg, opts=opts))
    for cfg, v in res.items():
        print("enc %d cross %2d: %s ms" % (cfg[0], cfg[1], " / ".join("%.3f" % x for x in v)))
    t = {k: min(v) for k, v in res.items()}
    print("encoder stacks (2+2 layers): %.3f ms;  two cross layers: %.3f ms;  one encoder layer pair: %.3f ms"
          % (t[(2, 12)] - t[(0, 12)], t[(2, 12)] - t[(2, 10)], t[(2, 12)] - t[(1, 12)]))
