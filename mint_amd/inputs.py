"""Batched input without tf.data: mirror of mint/core/inputs.py `create_input` (row f2).

Yields dicts of torch tensors (`motion_input`, `audio_input`, `target`) plus the `motion_name` /
`audio_name` string lists, batched with the GLOBAL train batch size per replica exactly like the
reference (which drops the input_context, trainer.py:102-103).  CPU-side plumbing only; the
benchmark uses synthetic tensors of the same shapes."""
import glob as _glob
import queue as _queue
import random
import threading

import numpy as np
import torch

from mint_amd import inputs_util, tfrecord


def _decode(payload, modalities):
    ex = tfrecord.parse_example(payload)
    out = {}
    for m in modalities:
        shape = tuple(int(x) for x in ex["%s_sequence_shape" % m])
        out["%s_sequence" % m] = np.asarray(ex["%s_sequence" % m], dtype=np.float32).reshape(shape)
        out["%s_name" % m] = ex["%s_name" % m][0].decode("utf-8")
    return out


def prefetch(gen, depth=2):
    """`ds.prefetch(...)` (inputs.py:118-123): a daemon thread runs the producer `gen` (file reads, Example
    decoding, FACT windowing, collation, pinned host-to-device copies) up to `depth` batches ahead of the consumer,
    so the next batch is staged while the GPU runs the current step.  Order is preserved, an exception in the producer
    re-raises in the consumer, and closing / dropping the returned generator stops the thread."""
    q = _queue.Queue(maxsize=max(1, int(depth)))
    stop = threading.Event()
    END, ERR = object(), object()

    def put(item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except _queue.Full:
                continue
        return False

    def work():
        try:
            for item in gen:
                if not put(item):
                    return
            put(END)
        except BaseException as e:  # handed to the consumer
            put((ERR, e))

    t = threading.Thread(target=work, name="mint_amd-input-prefetch", daemon=True)
    t.start()

    def consume():
        try:
            while True:
                item = q.get()
                if item is END:
                    return
                if isinstance(item, tuple) and len(item) == 2 and item[0] is ERR:
                    raise item[1]
                yield item
        finally:
            stop.set()

    return consume()


_EVENT = "__copy_done__"


class _DeviceStager:
    """Host-to-device staging of batches: a ring of PRE-PINNED host buffers and a dedicated copy stream; the consumer's
    stream waits for the batch's event (`_join_copies`).  Measured on MI355X (tools/attic/e2e_probe.py, fact_v5 B = 16 train
    step): resident batch 8.01 ms; this scheme 8.02 ms; `pin_memory().to(device, non_blocking=True)` per batch - the
    round-1 code - 16.7 ms (the freshly pinned temporary is released while its copy is in flight); pageable `.to(device)`
    8.19 ms."""

    def __init__(self, device, slots):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.ring = [dict() for _ in range(max(2, slots))]
        self.events = [None] * len(self.ring)
        self.i = 0

    def stage(self, arrays):
        """{key: list of equally shaped numpy arrays} -> ({key: device tensor}, event recorded behind the copies)"""
        i, self.i = self.i, (self.i + 1) % len(self.ring)
        if self.events[i] is not None:
            self.events[i].synchronize()  # the copies that last read this slot's pinned buffers are done
        out = {}
        with torch.cuda.stream(self.stream):
            for k, parts in arrays.items():
                shape = (len(parts),) + tuple(parts[0].shape)
                dtype = torch.from_numpy(np.empty(0, parts[0].dtype)).dtype
                pinned = self.ring[i].get(k)
                if pinned is None or tuple(pinned.shape) != shape or pinned.dtype != dtype:
                    pinned = self.ring[i][k] = torch.empty(shape, dtype=dtype, pin_memory=True)
                np.stack(parts, out=pinned.numpy())
                out[k] = pinned.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.events[i] = ev
        return out, ev


def _join_copies(gen):
    """Consumer side of `_DeviceStager`: the consumer's current stream waits for the batch's copies, and the device tensors
    (allocated on the copy stream) are marked as used by that stream."""
    for batch in gen:
        ev = batch.pop(_EVENT, None)
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for v in batch.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)
        yield batch


def create_input(train_eval_config, dataset_config, num_cpu_threads=2, is_training=True, use_tpu=False,
                 device=None, seed=None, prefetch_batches=2, cache_decoded_bytes=8 << 30):
    """Generator of feature dicts (inputs.py:20-123). Training: shuffle(100), repeat forever,
    drop the remainder; eval: one pass in file order, remainder kept.  `prefetch_batches` > 0 stages that many
    batches ahead on a background thread (the reference's `ds.prefetch`); 0 = synchronous.

    `cache_decoded_bytes`: training repeats the files forever and cuts ONE 360-frame window out of a multi-thousand
    frame track per visit, so the decoded tracks of a file are kept (up to this many bytes in total; 0 = re-read and
    re-parse every epoch like tf.data does).  One MI355X consumes ~2 000 windows/s (bench.py); parsing a 2.3 MB
    Example per window in Python delivers ~1 300/s, windows cut from cached tracks ~9 000/s (tools/attic/input_bench.py).
    Order, shuffling and the random windows are unchanged by the cache."""
    batch_size = train_eval_config.batch_size
    files = sorted(_glob.glob(dataset_config.data_files))
    if not files:
        # tf.data.Dataset.list_files raises on an empty match; a training generator would otherwise spin forever
        raise ValueError("data_files pattern %r matched no files" % dataset_config.data_files)
    params = inputs_util.get_modality_to_param_dict(dataset_config)
    use_fact = any(o.WhichOneof("preprocessor") == "fact_preprocessor"
                   for o in dataset_config.data_augmentation_options)
    rng = np.random.RandomState(seed)
    pyrng = random.Random(seed)

    cache, cached = {}, [0]

    def decoded(path):
        """decoded records of one file, from the cache when the file has been seen (training only)"""
        if path in cache:
            yield from cache[path]
            return
        keep = [] if (is_training and cache_decoded_bytes > 0) else None
        for payload in tfrecord.read_records(path):
            ex = _decode(payload, list(params))
            if keep is not None:
                keep.append(ex)
            yield ex
        if keep is not None:
            size = sum(v.nbytes for ex in keep for v in ex.values() if isinstance(v, np.ndarray))
            if cached[0] + size <= cache_decoded_bytes:
                cache[path] = keep
                cached[0] += size

    def examples():
        while True:
            order = list(files)
            if is_training:
                pyrng.shuffle(order)
            buf = []
            for path in order:
                for ex in decoded(path):
                    if use_fact:
                        ex = inputs_util.fact_preprocessing(ex, params, is_training, rng)
                    if not is_training:
                        yield ex
                        continue
                    buf.append(ex)
                    if len(buf) >= 100:  # ds.shuffle(100)
                        yield buf.pop(pyrng.randrange(len(buf)))
            while buf:
                yield buf.pop(pyrng.randrange(len(buf)))
            if not is_training:
                return

    def collate(batch):
        out, arrays = {}, {}
        for k in batch[0]:
            if isinstance(batch[0][k], str):
                out[k] = [b[k] for b in batch]
            else:
                arrays[k] = [b[k] for b in batch]
        if stager is not None:
            tensors, ev = stager.stage(arrays)
            out.update(tensors)
            out[_EVENT] = ev
        else:
            out.update({k: torch.from_numpy(np.stack(v)) for k, v in arrays.items()})
        return out

    def batches():
        batch = []
        for ex in examples():
            batch.append(ex)
            if len(batch) == batch_size:
                yield collate(batch)
                batch = []
        if batch and not is_training:
            yield collate(batch)

    depth = prefetch_batches if prefetch_batches and prefetch_batches > 0 else 0
    stager = _DeviceStager(device, depth + 2) if (device is not None and torch.cuda.is_available()) else None
    gen = prefetch(batches(), depth) if depth else batches()
    return _join_copies(gen) if stager is not None else gen
