"""Train-step host logic: mirror of mint/ctl/single_task_trainer.py (SingleTaskTrainer,
IdentityMetric) and of the Keras-Adam / Orbit pieces trainer.py wires around it, on top of the
HIP engine.  Data parallelism is one process per GPU with torch.distributed (backend "nccl" =
RCCL over xGMI on ROCm, "gloo" in the CPU tests): per-replica loss is divided by the replica count
and gradients are SUMMED across replicas (single_task_trainer.py:157-158, 186-187).
"""
import torch

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None


class IdentityMetric:
    """Metric that reports the last value assigned (single_task_trainer.py:21-47)."""

    def __init__(self, name, aggregation="sum"):
        self.name = name
        self.aggregation = aggregation
        self.value = 0.0

    def update_state(self, current_value):
        self.value = current_value

    def reset_states(self):
        self.value = 0.0

    def result(self):
        v = self.value
        return float(v.item()) if torch.is_tensor(v) else float(v)


class Adam:
    """tf.keras.optimizers.Adam as trainer.py:150 builds it: Keras defaults beta_1=0.9,
    beta_2=0.999, epsilon=1e-7 (epsilon outside the bias correction), `learning_rate` a float or a
    schedule callable evaluated at `iterations`."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self._lr = learning_rate
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.iterations = 0

    def learning_rate(self, step=None):
        step = self.iterations if step is None else step
        return float(self._lr(step)) if callable(self._lr) else float(self._lr)

    def apply_gradients(self, model, clip_norm=0.0):
        lr = self.learning_rate(self.iterations)
        model.apply_adam(lr, self.beta_1, self.beta_2, self.epsilon, clip_norm)
        self.iterations += 1
        return lr

    def begin_fused(self, model):
        """Arm the optimizer step that the engine applies inside the coming backward pass."""
        lr = self.learning_rate(self.iterations)
        model.begin_fused_adam(lr, self.beta_1, self.beta_2, self.epsilon)
        self.iterations += 1
        return lr


def replica_count():
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


def allreduce_gradients(grad_arena, bucket_bytes=64 << 20, async_op=False):
    """Sum the flat fp32 gradient arena across replicas in fixed-size buckets (one RCCL all-reduce
    per bucket, so the first buckets are on the wire while later ones are still being queued).
    Returns the list of work handles when async_op is set."""
    if replica_count() == 1:
        return []
    n = grad_arena.numel()
    step = max(1, bucket_bytes // grad_arena.element_size())
    works = []
    for lo in range(0, n, step):
        w = dist.all_reduce(grad_arena[lo:min(n, lo + step)], op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            works.append(w)
    return works


class OverlappedGradReducer:
    """Sum-all-reduces gradient buckets on a dedicated communication stream WHILE the backward pass is still
    running: the engine reports each contiguous bucket as soon as its producers are enqueued (head, cross
    layers L-1..0, audio stack, motion stack; ~30 MB fp32 per cross layer).  RCCL runs on xGMI beside the
    remaining dgrad / wgrad kernels; `finish()` makes the compute stream wait for the collectives before the
    optimizer step.

    `bf16_buckets=True` halves the payload (240.8 MB instead of 481.6 MB per step): a bucket is
    cast to bf16 into a communication buffer, all-reduced there, and written back into the fp32 arena the
    optimizer reads (the casts are the model's `cast_bucket_*` methods: HIP kernels on the engine).  The sum
    then carries one bf16 rounding per replica contribution.

    The protocol only needs `set_grad_callback(fn, stream)`, `grad_arena` and (for bf16) the two cast methods
    from the model, so tests/test_trainer_dist.py drives THIS class on CPU (gloo, world 2) with a stand-in
    that reports its buckets the way the engine does."""

    def __init__(self, model, bf16_buckets=False, keep_streams_low=False):
        self.model = model
        self.bf16 = bool(bf16_buckets)
        self.comm = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.works = []       # (work handle, bucket offset, count, bf16 buffer or None)
        self.fused_adam = False  # set per step by the trainer
        self._buf16 = None
        self.buckets_seen = []   # (bucket, offset, count) of the step in flight, in arrival order
        # diagnostics (bench.py --gpus N): with `profile` on, every bucket's all-reduce is bracketed by events on the
        # communication stream and finish() by events on the compute stream; read with `comm_report()`
        self.profile = False
        self._prof = []          # (bucket, count, event before, event after) of profiled steps
        self._prof_finish = []   # (event before finish, event after) per profiled step
        model.set_grad_callback(self._on_bucket, self.comm)
        # This stream and RCCL's own join the engine's three: more streams than the 4 hardware queues HIP uses by
        # default, and streams that share a queue serialise (one-GPU dry run with RCCL initialised, tools/attic/dp_probe.py:
        # 11.4 ms per step at GPU_MAX_HW_QUEUES = 4, 8.6 at 6, 8.2 at 7 - the single-replica speed - and 16.5 at 8).
        # Launchers should export GPU_MAX_HW_QUEUES=7 before the HIP runtime starts (bench.py does for N > 1).
        # `keep_streams_low` folds the engine's third stream instead (no measurable help at 4 queues: 11.4 vs 11.0).
        if keep_streams_low and hasattr(model, "set_option"):
            model.set_option("aux_stream", 0)

    def _stream(self):
        import contextlib
        return torch.cuda.stream(self.comm) if self.comm is not None else contextlib.nullcontext()

    def _on_bucket(self, bucket, offset, count):
        self.buckets_seen.append((bucket, offset, count))
        arena = self.model.grad_arena
        with self._stream():
            if self.bf16:
                if self._buf16 is None or self._buf16.numel() != arena.numel():
                    self._buf16 = torch.empty(arena.numel(), dtype=torch.bfloat16, device=arena.device)
                buf = self._buf16[offset:offset + count]
                self.model.cast_bucket_to_bf16(arena[offset:offset + count], buf, self.comm)
                e0 = self._mark()
                w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
            else:
                buf = None
                e0 = self._mark()
                w = dist.all_reduce(arena[offset:offset + count], op=dist.ReduceOp.SUM, async_op=True)
            if e0 is not None:
                w.wait()  # profiling: the communication stream waits here, so the closing event times the collective
                self._prof.append((bucket, count, e0, self._mark()))
            if self.fused_adam:
                # optimizer step of this bucket right behind its all-reduce, on the communication stream: the
                # HBM-bound update overlaps the rest of backward instead of following it as one serial pass
                w.wait()  # the communication stream waits for the collective
                if buf is not None:  # Adam reads the reduced bf16 bucket itself: no cast back into the arena
                    self.model.adam_bucket(bucket, self.comm, grads_bf16=self._buf16)
                else:
                    self.model.adam_bucket(bucket, self.comm)
            else:
                self.works.append((w, offset, count, buf))

    def _mark(self):
        if not (self.profile and self.comm is not None):
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(self.comm)
        return e

    def comm_report(self):
        """Per-bucket all-reduce time and the time the compute stream spent in finish() (= communication that backward
        did not hide, plus the bf16 cast-back when Adam is not fused), averaged over the profiled steps."""
        if not self._prof:
            return None
        torch.cuda.synchronize()
        per, steps = {}, max(1, len(self._prof_finish))
        for bucket, count, e0, e1 in self._prof:
            d = per.setdefault(bucket, {"bucket": bucket, "MB": round(count * (2 if self.bf16 else 4) / 1e6, 2), "ms": 0.0, "n": 0})
            d["ms"] += e0.elapsed_time(e1)
            d["n"] += 1
        rows = [{"bucket": d["bucket"], "MB": d["MB"], "allreduce_ms": round(d["ms"] / d["n"], 3)} for d in per.values()]
        exposed = sum(a.elapsed_time(b) for a, b in self._prof_finish) / steps
        return {"buckets": rows, "allreduce_ms_per_step": round(sum(r["allreduce_ms"] for r in rows), 3),
                "exposed_ms_per_step": round(exposed, 3), "profiled_steps": steps,
                "payload": "bf16" if self.bf16 else "fp32"}

    def finish(self):
        if self.profile and torch.cuda.is_available():
            a = torch.cuda.Event(enable_timing=True)
            a.record()
        else:
            a = None
        self._finish()
        if a is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self._prof_finish.append((a, b))

    def _finish(self):
        arena = self.model.grad_arena
        for w, offset, count, buf in self.works:
            if buf is None:
                w.wait()  # current (compute) stream waits for the collective
            else:
                # the cast back runs on the COMMUNICATION stream: that stream has to wait for the collective (RCCL runs
                # it on its own stream); the compute stream joins the communication stream below
                with self._stream():
                    w.wait()
                    self.model.cast_bucket_from_bf16(buf, arena[offset:offset + count], self.comm)
        self.works = []
        self.buckets_seen = []
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)


class SingleTaskTrainer:
    """Trains a single-output model on a given dataset (single_task_trainer.py:50-211).

    `train_dataset` is any iterable of feature dicts containing `label_key`; `model` is a
    mint_amd FACTModel; `loss_fn` is accepted for signature parity (the engine computes the
    reference's loss, fact_model.py:143-148, fused with the backward pass)."""

    def __init__(self, train_dataset, label_key, model, loss_fn=None, optimizer=None, metrics=None,
                 trainer_options=None, summary_fn=None, grad_clip_norm=0.0, overlap_grad_allreduce=None,
                 fuse_optimizer=True, bf16_grad_buckets=False, dp_fused_adam=True):
        self.train_dataset = train_dataset
        self.label_key = label_key
        self.model = model
        self.loss_fn = loss_fn
        self.optimizer = optimizer if optimizer is not None else Adam()
        self.summary_fn = summary_fn
        self.grad_clip_norm = grad_clip_norm
        self.num_replicas_in_sync = replica_count()
        self.train_loss = IdentityMetric("training_loss", "sum")
        self.task_loss = IdentityMetric("task_loss", "sum")
        self.regularization_loss = IdentityMetric("regularization_loss", "sum")
        self.learning_rate = IdentityMetric("learning_rate", "only_first_replica")
        if metrics is None:
            self.metrics = []
        elif isinstance(metrics, list):
            self.metrics = metrics
        else:
            self.metrics = [metrics]
        self._iter = None
        self._grad_overwrite_set = False
        self._reducer = None
        self._bf16_buckets = bool(bf16_grad_buckets)
        if overlap_grad_allreduce is None:
            overlap_grad_allreduce = (self.num_replicas_in_sync > 1 and hasattr(model, "set_grad_callback")
                                      and dist.get_backend() == "nccl")
        # per-replica clipping needs the whole local gradient before anything is summed: no bucket overlap then
        self._overlap = bool(overlap_grad_allreduce) and not (grad_clip_norm > 0.)  # reducer is created lazily
        # Optimizer step inside backward (engine API, no global-norm clipping).  Single replica: the engine holds the
        # head + cross-modal buckets back until the cross-modal backward is done and updates them beside the two small
        # encoder stacks' backward.  Data parallel: see `dp_fused_adam` below.
        self._fuse = bool(fuse_optimizer) and hasattr(model, "begin_fused_adam") and not (grad_clip_norm > 0.)
        # Data parallel: Adam of a bucket runs behind that bucket's all-reduce (communication stream) instead of as
        # one pass after the last all-reduce - that pass is 0.8-0.9 ms of an 8.2 ms step that nothing overlaps.
        self._dp_fused_adam = dp_fused_adam

    def train_loop_begin(self):
        self.train_loss.reset_states()
        self.task_loss.reset_states()
        self.regularization_loss.reset_states()
        self.learning_rate.reset_states()
        for metric in self.metrics:
            metric.reset_states()

    def train_step(self, iterator=None):
        """One optimizer step (train_fn, single_task_trainer.py:141-196)."""
        if iterator is None:
            if self._iter is None:
                self._iter = iter(self.train_dataset)
            iterator = self._iter
        inputs = dict(next(iterator))
        target = inputs.pop(self.label_key)  # the model never sees it
        R = self.num_replicas_in_sync
        if (self._overlap or self._fuse) and hasattr(self.model, "ensure_built"):
            self.model.ensure_built(inputs)
        if not self._grad_overwrite_set and hasattr(self.model, "set_option"):
            # one forward_backward per optimizer step: the layer weight gradients can be WRITTEN by the wgrad launches
            # instead of accumulated, and the optimizer pass need not zero them (engine option grad_overwrite)
            if hasattr(self.model, "ensure_built"):
                self.model.ensure_built(inputs)
            import os
            # not swallowed if it fails: a trainer that believes the option is on while the engine accumulates (or the
            # reverse) corrupts a step silently.  FACTModel remembers the option and re-applies it to re-created handles.
            self.model.set_option("grad_overwrite", int(os.environ.get("FACT_GRAD_OVERWRITE", "1")))
            self._grad_overwrite_set = True
        if self._overlap and self._reducer is None:
            self._reducer = OverlappedGradReducer(self.model, bf16_buckets=self._bf16_buckets)
        # fused path: single replica without a gradient callback (the engine updates the buckets itself), or - with
        # the overlapped reducer - every bucket's update right behind its all-reduce on the communication stream
        dp_fused = (self._fuse and self._reducer is not None and bool(self._dp_fused_adam)
                    and (R > 1 or self._dp_fused_adam == "force")  # "force": single-process test of this very path
                    and hasattr(self.model, "adam_bucket"))
        fused = (self._fuse and R == 1 and self._reducer is None) or dp_fused
        step = self.optimizer.iterations  # summaries are written at the PRE-update step (:172-173)
        lr_used = None
        output = None
        if self.metrics:
            # user metrics see (target, output) of the PRE-update weights, like the reference's `output` of the taped
            # forward (:151,194-196): run that forward before anything can update the parameters (one extra pass;
            # FACT's get_metrics() is empty, so the hot path never pays it)
            output = self.model(inputs, training=True)
        if self._reducer is not None:
            self._reducer.fused_adam = dp_fused
        if fused:
            snap = (self.optimizer.iterations, self.model.global_step)
            lr_used = self.optimizer.begin_fused(self.model)
        try:
            raw_loss = self.model.forward_backward(inputs, target, loss_scale=1.0 / R)
        except Exception:
            if fused:  # disarm the in-backward optimizer and roll the counters back
                self.optimizer.iterations, self.model.global_step = snap
                if hasattr(self.model, "cancel_fused_adam"):
                    self.model.cancel_fused_adam(snap[1])
            raise
        loss = raw_loss / R
        regularization_loss = 0.0  # model.losses is empty: no regularisers
        total_loss = loss + regularization_loss
        if self.summary_fn:
            self.summary_fn({"total_loss": total_loss, "loss:": loss, "reg_loss": regularization_loss}, step)
        clip_in_adam = self.grad_clip_norm
        if R > 1 and self.grad_clip_norm > 0.:
            # tf.clip_by_global_norm runs on each replica's OWN gradient before apply_gradients sums
            # them (:180-187): scale the local arena first, then reduce; nothing is left for Adam to clip
            # (engine kernels: fact_clip_gradients - sum of squares + in-place scale on the device, no host sync)
            self.model.clip_gradients(self.grad_clip_norm)
            clip_in_adam = 0.0
        if self._reducer is not None:
            self._reducer.finish()
        elif not fused:
            allreduce_gradients(self.model.grad_arena)
        if not fused:
            lr_used = self.optimizer.apply_gradients(self.model, clip_norm=clip_in_adam)
        self.train_loss.update_state(total_loss)
        self.task_loss.update_state(loss)
        self.regularization_loss.update_state(regularization_loss)
        # the reference reports the schedule at the POST-increment iteration count (:192-193)
        self.learning_rate.update_state(self.optimizer.learning_rate(self.optimizer.iterations))
        if self.metrics:
            for metric in self.metrics:
                metric.update_state(target, output)
        return total_loss

    def train_loop_end(self):
        metrics = {m.name: m.result() for m in self.metrics}
        # the loss metrics aggregate with SUM across replicas (:110-114)
        vals = []
        for m in (self.train_loss, self.task_loss, self.regularization_loss):
            v = m.value
            vals.append(v.detach().float().reshape(1) if torch.is_tensor(v) else None)
        if self.num_replicas_in_sync > 1 and all(v is not None for v in vals[:2]):
            t = torch.cat([vals[0], vals[1]])
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            metrics[self.train_loss.name] = float(t[0].item())
            metrics[self.task_loss.name] = float(t[1].item())
        else:
            metrics[self.train_loss.name] = self.train_loss.result()
            metrics[self.task_loss.name] = self.task_loss.result()
        metrics[self.regularization_loss.name] = self.regularization_loss.result()
        metrics[self.learning_rate.name] = self.learning_rate.result()
        return metrics


def train(trainer, steps, steps_per_loop=10, on_loop_end=None):
    """Minimal stand-in for orbit.Controller.train (trainer.py:164-178): loops of
    `steps_per_loop` train steps bracketed by train_loop_begin / train_loop_end."""
    done, history = 0, []
    while done < steps:
        k = min(steps_per_loop, steps - done)
        trainer.train_loop_begin()
        for _ in range(k):
            trainer.train_step()
        metrics = trainer.train_loop_end()
        done += k
        history.append((done, metrics))
        if on_loop_end:
            on_loop_end(done, metrics)
    return history
