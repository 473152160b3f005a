"""Host input pipeline throughput (row f2): TFRecord -> Example -> FACT windows -> batches of 16, synthetic AIST++-sized
tracks (1 500-3 000 frames).  One MI355X consumes ~2 000 windows/s (bench.py)."""
import time, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mint_amd import inputs, protos, tfrecord
d = protos.Dataset()
d.name = "train"
d.input_length_sec, d.target_length_sec, d.target_shift_sec = 120.0, 20, 120
for name, dim, rate in (("motion", 219, 1), ("audio", 35, 2)):
    g = d.modality.add().general_modality
    g.feature_name, g.dimension, g.sample_rate = name, dim, rate
d.data_augmentation_options.add().fact_preprocessor.CopyFrom(protos.FACTPreprocessor())
tmp = tempfile.mkdtemp()
rng = np.random.RandomState(0)
t0 = time.time()
for f in range(4):
    recs = []
    for i in range(25):
        n = 1500 + rng.randint(0, 1500)
        m = rng.randn(n, 219).astype(np.float32); a = rng.randn(2 * n, 35).astype(np.float32)
        recs.append(tfrecord.make_example({"motion_name": "m%d" % i, "motion_sequence": m.flatten(), "motion_sequence_shape": np.array(m.shape),
                     "audio_name": "a%d" % i, "audio_sequence": a.flatten(), "audio_sequence_shape": np.array(a.shape)}))
    tfrecord.write_records(os.path.join(tmp, "aist_tfrecord-train-%d" % f), recs)
print("wrote 100 records in %.1fs" % (time.time() - t0))
d.data_files = os.path.join(tmp, "*_tfrecord-train*")
tc = protos.TrainConfig(); tc.batch_size = 16
for pf, cache in ((0, 0), (0, 8 << 30), (2, 8 << 30)):
    it = inputs.create_input(tc, d, is_training=True, seed=0, prefetch_batches=pf, cache_decoded_bytes=cache)
    next(it)
    t0 = time.time(); n = 0
    while time.time() - t0 < 8:
        next(it); n += 16
    print("prefetch %d, decoded-track cache %s: %.0f samples/s" % (pf, "on" if cache else "off", n / (time.time() - t0)))
