"""The main FACT model — host-side mirror of mint/core/fact_model.py:26-148 on the HIP engine.

Same constructor and methods as the reference class (`__call__(inputs)`, `loss(target, pred)`,
`infer_auto_regressive(inputs, steps)`, `get_metrics`, `trainable_variables`, `losses`,
`global_step`), same error behaviour for bad configs.  Tensors are torch CUDA tensors; all math
runs in libfact_hip.so (no torch ops on the hot path, no CPU fallback).
"""
import copy
import ctypes as C
import math

import torch

from mint_amd import _lib as L
from mint_amd import protos


def build_modalities_model(modality_config):
    """Process the parameters in the modality config (mint/core/multi_modal_model_util.py:24-56)."""
    feature_to_model, feature_to_params, feature_to_preprocessor = {}, {}, {}
    for modality in modality_config:
        name = modality.feature_name
        feature_to_params[name] = {"sequence_length": modality.sequence_length,
                                   "feature_dim": modality.feature_dim}
        feature_to_model[name] = {}
        feature_to_preprocessor[name] = [None for _ in modality.preprocessor]
        for model in modality.model:
            if model.WhichOneof("model") == "transformer":
                feature_to_model[name]["transformer_layer"] = model.transformer
    return feature_to_model, feature_to_params, feature_to_preprocessor


def _trunc_normal_(t, std, gen):
    """tf.keras.initializers.TruncatedNormal: resample outside +-2 sigma."""
    t.normal_(0.0, std, generator=gen)
    for _ in range(64):
        bad = t.abs() > 2 * std
        if not bad.any():
            break
        t[bad] = torch.empty(int(bad.sum()), dtype=t.dtype).normal_(0.0, std, generator=gen)
    return t


class FACTModel:
    """Audio Motion Multi-Modal model (FACT) on MI355X."""

    def __init__(self, config, is_training, seed=0, device=None):
        self.config = copy.deepcopy(config)
        self.is_training = is_training
        (self.feature_to_model, self.feature_to_params,
         self.feature_to_preprocessor) = build_modalities_model(self.config.modality)
        # the reference indexes these names directly (fact_model.py:45-48) -> KeyError if absent
        self._motion_cfg = self.feature_to_model["motion"]["transformer_layer"]
        self._audio_cfg = self.feature_to_model["audio"]["transformer_layer"]
        cross = self.config.cross_modal_model
        self._cross_cfg = cross.transformer
        self._out_dim = cross.output_layer.out_dim
        self._out_init_range = cross.output_layer.initializer_range
        self.global_step = 0
        self.losses = []  # no regularisers (single_task_trainer.py:163-164 reads this)
        self._seed = seed
        dev = torch.device(device if device is not None else "cuda")
        if dev.type == "cuda" and dev.index is None and torch.cuda.is_available():
            dev = torch.device("cuda", torch.cuda.current_device())
        self._device = dev
        self._h = None
        self._max_batch = 0
        self._arena = None  # dict of torch tensors: params, grads, adam_m, adam_v
        self._table = None
        self._feat = {"motion": self.feature_to_params["motion"]["feature_dim"],
                      "audio": self.feature_to_params["audio"]["feature_dim"]}
        self._loss_buf = None
        self._grad_cb = None
        self._grad_cb_args = None
        self._options = {}  # engine options set through set_option / debug_option: key -> (value, debug); re-applied by build()

    # ------------------------------------------------------------------------------------------
    def _cfg_struct(self):
        def stack(t, seq, feat):
            return L.FactStackCfg(seq_len=seq, feature_dim=feat, hidden=t.hidden_size,
                                  layers=t.num_hidden_layers, heads=t.num_attention_heads,
                                  ff=t.intermediate_size)
        p = self.feature_to_params
        return L.FactConfig(
            motion=stack(self._motion_cfg, p["motion"]["sequence_length"], self._feat["motion"]),
            audio=stack(self._audio_cfg, p["audio"]["sequence_length"], self._feat["audio"]),
            cross=stack(self._cross_cfg, 0, 0), out_dim=self._out_dim, ln_eps=1e-5)

    def _check_config(self, motion_width, audio_width):
        cross = self.config.cross_modal_model
        # base_models.py:184-189
        if self._motion_cfg.hidden_size != self._audio_cfg.hidden_size:
            raise ValueError("The modal_a hidden size (%d) should be the same with the modal_b "
                             "hidden size (%d)" % (self._motion_cfg.hidden_size, self._audio_cfg.hidden_size))
        # base_models.py:190-196
        if cross.cross_modal_concat_dim != protos.CrossModalModel.CrossModalConcatDim.SEQUENCE_WISE:
            raise NotImplementedError("cross_modal_concat_dim %s is not supported." % cross.cross_modal_concat_dim)

    def build(self, max_batch, motion_feature_dim=None, audio_feature_dim=None):
        """Create (or re-create for a larger batch) the engine handle. Variables are created on
        first call like Keras does; input widths unset in the config (audio `feature_dim`, Q10)
        are taken from the first batch."""
        lib = L.lib()
        if motion_feature_dim and not self._feat["motion"]:
            self._feat["motion"] = int(motion_feature_dim)
        if audio_feature_dim and not self._feat["audio"]:
            self._feat["audio"] = int(audio_feature_dim)
        self._check_config(self._feat["motion"], self._feat["audio"])
        if self._h is not None and max_batch <= self._max_batch:
            return
        if not torch.cuda.is_available():  # the product path has no CPU fallback: say so instead of a torch device error
            raise RuntimeError("mint_amd runs on an AMD GPU (gfx950) through libfact_hip.so; torch.cuda.is_available() is "
                               "False on this host and there is no CPU path")
        torch.cuda.set_device(self._device)
        cfg = self._cfg_struct()
        n_floats, n_tensors = C.c_size_t(0), C.c_int(0)
        L.check(lib.fact_arena_size(C.byref(cfg), C.byref(n_floats), C.byref(n_tensors)))
        first = self._arena is None
        if first:
            names = ["params"] + (["grads", "adam_m", "adam_v"] if self.is_training else [])
            self._arena = {k: torch.zeros(n_floats.value, dtype=torch.float32, device=self._device)
                           for k in names}
        if self._h is not None:
            torch.cuda.synchronize()
            lib.fact_destroy(self._h)
            self._h = None
        ar = L.FactArenas(params=self._arena["params"].data_ptr(),
                          grads=self._arena["grads"].data_ptr() if self.is_training else None,
                          adam_m=self._arena["adam_m"].data_ptr() if self.is_training else None,
                          adam_v=self._arena["adam_v"].data_ptr() if self.is_training else None)
        h = C.c_void_p()
        L.check(lib.fact_create(C.byref(cfg), int(max_batch), 1 if self.is_training else 0, C.byref(ar),
                                C.byref(h)))
        self._h = h
        self._max_batch = int(max_batch)
        tab, n = C.POINTER(L.FactParamDesc)(), C.c_int(0)
        L.check(lib.fact_param_table(self._h, C.byref(tab), C.byref(n)))
        self._table = [(tab[i].name.decode(), int(tab[i].offset), int(tab[i].rows), int(tab[i].cols),
                        int(tab[i].kind)) for i in range(n.value)]
        if first:
            self._init_parameters()
        L.check(lib.fact_set_step(self._h, int(self.global_step)))
        L.check(lib.fact_refresh_weights(self._h, L.cur_stream()))
        self._loss_buf = torch.zeros(1, dtype=torch.float32, device=self._device)
        if self._grad_cb_args is not None:  # handle was re-created: re-register the bucket callback
            self.set_grad_callback(*self._grad_cb_args)
        for key, (value, debug) in self._options.items():  # options are per handle: a re-created handle gets them again
            self._apply_option(key, value, debug)
        if getattr(self, "_pending_state", None) is not None:
            state, self._pending_state = self._pending_state, None
            self.load_state_dict(state)

    def _init_parameters(self):
        """Reference initialisers: glorot_uniform Dense kernels and zero biases
        everywhere (Transformer ignores initializer_range, base_models.py:94-107), LN gamma=1 /
        beta=0, TruncatedNormal(0.02) position tables (:147) and output-head kernel (:176-180)."""
        gen = torch.Generator().manual_seed(self._seed)
        host = torch.zeros(self._arena["params"].numel(), dtype=torch.float32)
        for name, off, rows, cols, kind in self._table:
            view = host[off:off + rows * cols]
            if kind == 0:
                if name == "cross_modal_layer/output/kernel":
                    _trunc_normal_(view, self._out_init_range, gen)
                else:
                    limit = math.sqrt(6.0 / (rows + cols))
                    view.uniform_(-limit, limit, generator=gen)
            elif kind == 2:
                view.fill_(1.0)
            elif kind == 4:
                _trunc_normal_(view, 0.02, gen)
        self._arena["params"].copy_(host)

    # ------------------------------------------------------------------------------------------
    def _prep(self, x):
        if not torch.is_tensor(x):
            x = torch.as_tensor(x)
        return x.to(device=self._device, dtype=torch.float32).contiguous()

    def _inputs(self, inputs, ar=False):
        """Validate and stage the two modal inputs.  Shapes are checked BEFORE raw pointers reach the
        C ABI (the engine indexes with the configured strides): the reference fails at the position
        embedding add (base_models.py:148-156) for any other sequence length / feature width.
        `ar=True` (auto-regressive sampling) admits an audio track longer than the window."""
        motion = self._prep(inputs["motion_input"])
        audio = self._prep(inputs["audio_input"])
        if motion.dim() != 3 or audio.dim() != 3 or motion.shape[0] != audio.shape[0]:
            raise ValueError("motion_input/audio_input must be [batch, seq, feature] with equal batch")
        self.build(motion.shape[0], motion.shape[2], audio.shape[2])
        p = self.feature_to_params
        if motion.shape[1] != p["motion"]["sequence_length"] or motion.shape[2] != self._feat["motion"]:
            raise ValueError("motion_input shape %s incompatible with [*, %d, %d]" % (
                tuple(motion.shape), p["motion"]["sequence_length"], self._feat["motion"]))
        n_a = p["audio"]["sequence_length"]
        bad_len = (audio.shape[1] < n_a) if ar else (audio.shape[1] != n_a)
        if bad_len or audio.shape[2] != self._feat["audio"]:
            raise ValueError("audio_input shape %s incompatible with [*, %s%d, %d]" % (
                tuple(audio.shape), ">=" if ar else "", n_a, self._feat["audio"]))
        return motion, audio

    def __call__(self, inputs, training=True):
        return self.call(inputs)

    def call(self, inputs):
        """Single forward pass (fact_model.py:72-101). Extra dict keys are ignored (Q12).
        Returns [batch, motion_seq + audio_seq, out_dim]; only the first N frames are supervised."""
        motion, audio = self._inputs(inputs)
        p = self.feature_to_params
        B = motion.shape[0]
        n = p["motion"]["sequence_length"] + p["audio"]["sequence_length"]
        out = torch.empty(B, n, self._out_dim, dtype=torch.float32, device=self._device)
        L.check(L.lib().fact_forward(self._h, L.ptr(motion), L.ptr(audio), B, L.ptr(out), L.cur_stream()))
        return out

    def infer_auto_regressive(self, inputs, steps=1200):
        """Auto-regressive generation (fact_model.py:103-132): keep frame 0 of each forward, shift
        the motion window by one, slide the audio window by one; stops early when the audio runs
        out. Returns [batch, steps_done, out_dim]."""
        motion, audio = self._inputs(inputs, ar=True)
        if self._out_dim != self._feat["motion"]:
            raise ValueError("auto-regressive inference feeds outputs back as motion frames: out_dim %d != "
                             "motion feature_dim %d" % (self._out_dim, self._feat["motion"]))
        B, audio_len = motion.shape[0], audio.shape[1]
        out = torch.empty(B, max(steps, 1), self._out_dim, dtype=torch.float32, device=self._device)
        done = C.c_int(0)
        L.check(L.lib().fact_infer_ar(self._h, L.ptr(motion), L.ptr(audio), B, audio_len, int(steps),
                                      L.ptr(out), C.byref(done), L.cur_stream()))
        return out[:, :done.value]

    def loss(self, target, pred):
        """Motion generation loss, argument order (target, pred) (fact_model.py:134-148):
        mean((target - pred[:, :target_len])^2)."""
        target, pred = self._prep(target), self._prep(pred)
        B, T, D = target.shape
        n = pred.shape[1]
        loss = torch.zeros(1, dtype=torch.float32, device=self._device)
        L.check(L.lib().fact_loss(L.ptr(target), L.ptr(pred), B, n, T, D, L.ptr(loss), L.cur_stream()))
        return loss[0]

    def compute_motion_generation_loss(self, pred_tensors, target_tensors):
        return self.loss(target_tensors, pred_tensors)

    def get_metrics(self, eval_config):
        """Off-line metrics only (fact_model.py:138-141)."""
        return []

    # ---- training hot path -----------------------------------------------------------------------
    def forward_backward(self, inputs, target, loss_scale=1.0):
        """Tape section of train_fn (single_task_trainer.py:141-178) fused: forward, loss, backward.
        Accumulates d(loss*loss_scale)/dparams into the grad arena; returns the unscaled loss as a
        0-dim CUDA tensor (no host sync)."""
        if not self.is_training:
            raise RuntimeError("model was built with is_training=False")
        motion, audio = self._inputs(inputs)
        target = self._prep(target)
        n_tot = self.feature_to_params["motion"]["sequence_length"] + self.feature_to_params["audio"]["sequence_length"]
        if (target.dim() != 3 or target.shape[0] != motion.shape[0] or target.shape[2] != self._out_dim
                or not (0 < target.shape[1] <= n_tot)):
            raise ValueError("target shape %s incompatible with [%d, 1..%d, %d]" % (
                tuple(target.shape), motion.shape[0], n_tot, self._out_dim))
        B, T = target.shape[0], target.shape[1]
        L.check(L.lib().fact_forward_backward(self._h, L.ptr(motion), L.ptr(audio), L.ptr(target), B, T,
                                              float(loss_scale), L.ptr(self._loss_buf), L.cur_stream()))
        return self._loss_buf[0]

    def ensure_built(self, inputs):
        """Create the engine for this batch if it does not exist yet (Keras builds on first call)."""
        self._inputs(inputs)

    def begin_fused_adam(self, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        """The next forward_backward also applies the optimizer step, bucket by bucket, overlapped
        with the rest of the backward pass (see include/fact_hip.h fact_adam_begin)."""
        self._require_built()
        L.check(L.lib().fact_adam_begin(self._h, float(lr), float(beta_1), float(beta_2), float(epsilon)))
        self.global_step += 1

    def cast_bucket_to_bf16(self, src_f32, dst_bf16, stream=None):
        """fp32 gradient range -> bf16 communication buffer (HIP cast kernel on `stream`)."""
        st = C.c_void_p(stream.cuda_stream) if stream is not None else L.cur_stream()
        L.check(L.lib().fact_cast_f32_bf16(L.ptr(src_f32), L.ptr(dst_bf16), src_f32.numel(), st))

    def cast_bucket_from_bf16(self, src_bf16, dst_f32, stream=None):
        """all-reduced bf16 bucket -> fp32 gradient arena range."""
        st = C.c_void_p(stream.cuda_stream) if stream is not None else L.cur_stream()
        L.check(L.lib().fact_cast_bf16_f32(L.ptr(src_bf16), L.ptr(dst_f32), src_bf16.numel(), st))

    def cancel_fused_adam(self, global_step):
        """Disarm a begin_fused_adam whose forward_backward never ran (it raised before reaching the engine)."""
        L.check(L.lib().fact_adam_cancel(self._h))
        self.global_step = int(global_step)

    def adam_bucket(self, bucket, stream, grads_bf16=None):
        """Optimizer step of one gradient bucket on `stream` (after begin_fused_adam).  `grads_bf16`: a bf16 tensor
        indexed like the gradient arena (the all-reduced communication buffer) to read the gradients from."""
        if grads_bf16 is None:
            L.check(L.lib().fact_adam_bucket(self._h, int(bucket), C.c_void_p(stream.cuda_stream)))
        else:
            L.check(L.lib().fact_adam_bucket_bf16(self._h, int(bucket), L.ptr(grads_bf16),
                                                  C.c_void_p(stream.cuda_stream)))

    def clip_gradients(self, clip_norm):
        """tf.clip_by_global_norm on this replica's OWN gradient arena, in place (single_task_trainer.py:180-183): the
        data-parallel step clips before the gradients are summed (:180-187), i.e. outside apply_adam.  Engine kernels
        (fact_clip_gradients), no host synchronisation."""
        self._require_built()
        L.check(L.lib().fact_clip_gradients(self._h, float(clip_norm), None, L.cur_stream()))

    def apply_adam(self, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clip_norm=0.0):
        L.check(L.lib().fact_adam_step(self._h, float(lr), float(beta_1), float(beta_2), float(epsilon),
                                       float(clip_norm), L.cur_stream()))
        self.global_step += 1

    # ---- variables -------------------------------------------------------------------------------
    def _require_built(self):
        if self._h is None:
            raise RuntimeError("variables are created on first call (call the model or .build() first)")

    @property
    def variable_names(self):
        self._require_built()
        return [t[0] for t in self._table]

    def _views(self, arena):
        self._require_built()
        a = self._arena[arena]
        return [a[off:off + r * c].view(r, c) if kind in (0, 4) else a[off:off + r * c]
                for _, off, r, c, kind in self._table]

    @property
    def trainable_variables(self):
        """Views into the fp32 master arena in Keras trainable_variables order (Dense kernels
        [in, out]). After writing to them call `sync_weights()`."""
        return self._views("params")

    @property
    def gradients(self):
        return self._views("grads")

    @property
    def grad_arena(self):
        self._require_built()
        return self._arena["grads"]

    def sync_weights(self):
        self._require_built()
        L.check(L.lib().fact_refresh_weights(self._h, L.cur_stream()))

    def set_grad_callback(self, fn, comm_stream):
        """Register `fn(bucket, offset, count)` to be called during forward_backward whenever the
        gradient range grad_arena[offset:offset+count] is final on `comm_stream` (a torch.cuda.Stream);
        fn=None disables.  See include/fact_hip.h fact_set_grad_callback."""
        self._require_built()
        self._grad_cb_args = None if fn is None else (fn, comm_stream)
        if fn is None:
            self._grad_cb = L.GRAD_CB(0)
            L.check(L.lib().fact_set_grad_callback(self._h, self._grad_cb, None, None))
            return
        self._grad_cb = L.GRAD_CB(lambda user, bucket, off, cnt: fn(int(bucket), int(off), int(cnt)))
        L.check(L.lib().fact_set_grad_callback(self._h, self._grad_cb, None,
                                               C.c_void_p(comm_stream.cuda_stream)))

    def kernel_profile(self, on=None):
        """In-step kernel-class timing (fact_kprof): `kernel_profile(True)` arms it, `kernel_profile()` reads the
        records as a list of dicts {name, launches, total_ms, flops, bytes} (device synchronised)."""
        self._require_built()
        L.need_debug_abi("kernel_profile()")
        lib = L.lib()
        if on is not None:
            L.check(lib.fact_kprof(self._h, 1 if on else 0))
            return None
        n, cap = C.c_int(0), 32
        names = (C.c_char_p * cap)()
        arrs = [(C.c_double * cap)() for _ in range(4)]
        L.check(lib.fact_kprof_read(self._h, cap, C.byref(n), names, *arrs))
        out = []
        buf = C.create_string_buffer(16384)
        for i in range(n.value):
            # the kernels the recorder saw behind this class: symbol (as rocprofv3 prints it), grid, block, LDS, occupancy
            L.check(lib.fact_kprof_kernels(self._h, i, buf, len(buf)))
            kernels = []
            for line in buf.value.decode(errors="replace").splitlines():
                f = line.split("\t", 5)
                if len(f) == 6:
                    kernels.append({"count": int(f[0]), "grid": int(f[1]), "block": int(f[2]), "lds_bytes": int(f[3]),
                                    "workgroups_per_cu": int(f[4]), "name": f[5]})
            out.append({"name": names[i].decode(), "launches": arrs[0][i], "total_ms": arrs[1][i], "flops": arrs[2][i],
                        "bytes": arrs[3][i], "kernels": kernels})
        return out

    def _apply_option(self, key, value, debug):
        if debug:
            L.need_debug_abi("debug_option(%r)" % key)
        fn = L.lib().fact_debug_set_option if debug else L.lib().fact_set_option
        L.check(fn(self._h, key.encode(), int(value)))

    def set_option(self, key, value):
        """Engine option of the production surface (include/fact_hip.h: sr_rows, grad_overwrite, adam_in_wgrad,
        side_stream, aux_stream).  Remembered and re-applied when build() re-creates the handle for a larger batch - options are
        per handle, and e.g. a trainer that set grad_overwrite must not silently fall back to accumulation."""
        if key not in L.PUBLIC_OPTIONS:
            raise ValueError("%r is not a production option %s; test / bench knobs: debug_option()" % (
                key, list(L.PUBLIC_OPTIONS)))
        self._require_built()
        self._apply_option(key, value, False)
        self._options[key] = (int(value), False)

    def debug_option(self, key, value):
        """Test / bench knob (include/fact_hip_debug.h, fact_debug_set_option): kernel-selection and scheduling A/B
        switches, the timing-only ablation mask.  Not for production hosts."""
        self._require_built()
        self._apply_option(key, value, True)
        self._options[key] = (int(value), key not in L.PUBLIC_OPTIONS)

    def state_dict(self):
        self._require_built()
        # the gradient arena is transient (zero between optimizer steps; under grad_overwrite partly stale by design): it is
        # neither saved nor restored
        d = {k: v.detach().cpu().clone() for k, v in self._arena.items() if k != "grads"}
        d["global_step"] = int(self.global_step)
        d["variable_names"] = self.variable_names
        return d

    def load_state_dict(self, state):
        if self._h is None:  # deferred restore: applied by build() right after the variables exist
            self._pending_state = state
            self.global_step = int(state.get("global_step", 0))
            return
        for k in self._arena:
            if k == "grads":
                self._arena[k].zero_()
            elif k in state:
                self._arena[k].copy_(state[k])
        self.global_step = int(state.get("global_step", 0))
        L.check(L.lib().fact_set_step(self._h, int(self.global_step)))
        self.sync_weights()

    def __del__(self):
        try:
            if self._h is not None:
                torch.cuda.synchronize()
                L.lib().fact_destroy(self._h)
                self._h = None
        except Exception:
            pass
