"""End-to-end train-step rate WITH the host input pipeline (row f2) in the loop: TFRecord files -> create_input (decoded-track
cache, prefetch thread, pinned host-to-device copies) -> SingleTaskTrainer on the engine, fact_v5 at B = 16, against the
same trainer fed one resident synthetic batch (what bench.py times).  python tools/e2e_input_bench.py [steps]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mint_amd import configs, inputs, model_builder, protos, tfrecord  # noqa: E402
from mint_amd.learning_schedules import create_learning_rate  # noqa: E402
from mint_amd.trainer import Adam, SingleTaskTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
d = protos.Dataset()
d.name = "train"
d.input_length_sec, d.target_length_sec, d.target_shift_sec = 120.0, 20, 120
for name, dim, rate in (("motion", 219, 1), ("audio", 35, 2)):
    g = d.modality.add().general_modality
    g.feature_name, g.dimension, g.sample_rate = name, dim, rate
d.data_augmentation_options.add().fact_preprocessor.CopyFrom(protos.FACTPreprocessor())
tmp = tempfile.mkdtemp()
rng = np.random.RandomState(0)
for f in range(4):
    recs = []
    for i in range(25):
        n = 1500 + rng.randint(0, 1500)
        m, a = rng.randn(n, 219).astype(np.float32), rng.randn(2 * n, 35).astype(np.float32)
        recs.append(tfrecord.make_example({
            "motion_name": "m%d" % i, "motion_sequence": m.flatten(), "motion_sequence_shape": np.array(m.shape),
            "audio_name": "a%d" % i, "audio_sequence": a.flatten(), "audio_sequence_shape": np.array(a.shape)}))
    tfrecord.write_records(os.path.join(tmp, "aist_tfrecord-train-%d" % f), recs)
d.data_files = os.path.join(tmp, "*_tfrecord-train*")
tc = protos.TrainConfig()
tc.batch_size = 16
pipe = configs.fact_v5_deeper_t10_cm12()
dev = torch.device("cuda", 0)


def run(dataset, label):
    model = model_builder.build(pipe.multi_modal_model, True)
    model.build(16, 225, 35)
    tr = SingleTaskTrainer(dataset, "target", model, optimizer=Adam(create_learning_rate(pipe.train_config.learning_rate)))
    it = iter(dataset)
    tr.train_loop_begin()
    for _ in range(10):
        tr.train_step(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.train_step(it)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%-46s %.3f ms/step  %.0f motion frames/s  (loss %.4f)" % (label, dt * 1e3, 16 * 120 / dt, float(loss)))
    return dt


class Resident:
    def __init__(self):
        g = torch.Generator().manual_seed(0)
        self.b = {"motion_input": torch.randn(16, 120, 225, generator=g).to(dev),
                  "audio_input": torch.randn(16, 240, 35, generator=g).to(dev),
                  "target": torch.randn(16, 20, 225, generator=g).to(dev)}

    def __iter__(self):
        return self

    def __next__(self):
        return self.b


class Pipeline:
    def __init__(self, **kw):
        self.kw = kw

    def __iter__(self):
        gen = inputs.create_input(tc, d, is_training=True, seed=0, device=dev, **self.kw)
        return ({k: v for k, v in b.items() if not k.endswith("_name")} for b in gen)


a = run(Resident(), "resident synthetic batch (bench.py's loop)")
b = run(Pipeline(), "TFRecord pipeline (cache + prefetch thread)")
c = run(Pipeline(prefetch_batches=0), "TFRecord pipeline, no prefetch thread")
e = run(Pipeline(cache_decoded_bytes=0), "TFRecord pipeline, re-parse every epoch")
print("input-inclusive / resident: %.3f (prefetch), %.3f (synchronous), %.3f (no cache)" % (a / b, a / c, a / e))
