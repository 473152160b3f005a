/* fact_hip.h — C ABI of the MI355X-native FACT engine (libfact_hip.so).
 *
 * The reference (google-research/mint) is pure Python/TF and has no FFI; this ABI is the boundary
 * a maintainer binds (ctypes) to replace what `model_builder.build()` returns
 * (mint/core/model_builder.py:29-33) and what `SingleTaskTrainer.train_step.train_fn` executes
 * (mint/ctl/single_task_trainer.py:141-196).  Each entry point names the reference code it replaces.
 *
 * Conventions: plain pointers and sizes only (no torch types). All tensor arguments are DEVICE
 * pointers, row-major, float32 unless stated. All work is enqueued on the caller's `stream`
 * (a hipStream_t passed as void*); no hidden synchronisation except where documented.
 * Return value: 0 = ok, negative = error (fact_last_error() gives the message).  One handle per
 * GPU per process; a handle is not thread-safe.
 */
#ifndef FACT_HIP_H_
#define FACT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libfact_hip.so is built with -fvisibility=hidden: the entry points declared in this header are the library's whole
 * exported surface (tests/test_cabi.py compares `nm -D` with this file). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* 2 (rounds 4-5): the A/B and ablation option keys, the recorder, single-op and probe entry points moved to
 * fact_hip_debug.h (fact_set_option rejects them); fact_create zeroes a caller-provided gradient arena; fact_loss added;
 * option "adam_in_wgrad" added (when set, a fused step under "grad_overwrite" no longer writes the gradient-arena ranges
 * of the transformer-layer Dense kernels).  A host built against version 1 must be re-read against this header. */
#define FACT_ABI_VERSION 3

/* One transformer stack (mint/core/base_models.py:91-110) plus the modality it embeds. */
typedef struct FactStackCfg {
  int seq_len;      /* Modality.sequence_length (model.proto:37-44) */
  int feature_dim;  /* width of the raw input features (225 motion / 35 audio) */
  int hidden;       /* Transformer.hidden_size */
  int layers;       /* Transformer.num_hidden_layers */
  int heads;        /* Transformer.num_attention_heads */
  int ff;           /* Transformer.intermediate_size (proto default 3072) */
} FactStackCfg;

/* FACTModel hyper-parameters (mint/core/fact_model.py:29-70). cross.seq_len / feature_dim unused. */
typedef struct FactConfig {
  FactStackCfg motion;
  FactStackCfg audio;
  FactStackCfg cross;
  int out_dim;   /* CrossModalModel.output_layer.out_dim (225) */
  float ln_eps;  /* 1e-5, base_models.py:27 */
} FactConfig;

typedef struct FactHandle FactHandle;

/* One trainable tensor, in Keras `trainable_variables` order; `offset` in floats into the arenas.
 * Dense kernels are [in, out] (rows=in, cols=out) exactly like Keras. */
typedef struct FactParamDesc {
  char name[96];
  size_t offset;
  int rows;
  int cols;
  int kind; /* 0 = dense kernel, 1 = bias, 2 = LN gamma, 3 = LN beta, 4 = position table */
} FactParamDesc;

/* Caller-owned arenas (optional): param/grad/adam_m/adam_v, each `arena_floats` floats. */
typedef struct FactArenas {
  float* params;
  float* grads;
  float* adam_m;
  float* adam_v;
} FactArenas;

int fact_abi_version(void);
/* Host utility (no GPU work): CRC-32C (Castagnoli) of `n` bytes continuing from `crc` (0 to start) - the checksum of
 * TFRecord frames (mint/core/inputs.py reads them through tf.data) and of TensorFlow tensor-bundle entries
 * (trainer.py:168-173 checkpoints); mint_amd/tfrecord.py and tf_checkpoint.py use it for large payloads. */
unsigned int fact_crc32c(const void* data, size_t n, unsigned int crc);
const char* fact_last_error(void);

/* Number of floats in each arena (padded, 16-byte aligned tensors) and number of tensors. */
int fact_arena_size(const FactConfig* cfg, size_t* arena_floats, int* n_tensors);

/* Replaces FACTModel.__init__ (fact_model.py:29-70). `arenas` may be NULL (library allocates).
 * Validates the config the way the reference does: equal hidden sizes for the cross-modal concat
 * (base_models.py:184-189) -> -2.  `max_batch` sizes the workspaces; `training` != 0 keeps
 * per-layer activations for the backward pass. */
int fact_create(const FactConfig* cfg, int max_batch, int training, const FactArenas* arenas,
                FactHandle** out);
int fact_destroy(FactHandle* h);

/* Parameter table / arena pointers (device). */
int fact_param_table(FactHandle* h, const FactParamDesc** table, int* n);
int fact_arenas(FactHandle* h, FactArenas* out, size_t* arena_floats);

/* Rebuild the bf16 weight shadows from the fp32 master parameters (call after writing params). */
int fact_refresh_weights(FactHandle* h, void* stream);

/* Replaces FACTModel.call (fact_model.py:72-101): motion (B, n_m, F_m), audio (B, n_a, F_a)
 * -> out (B, n_m + n_a, out_dim). */
int fact_forward(FactHandle* h, const float* motion, const float* audio, int B, float* out,
                 void* stream);

/* Replaces FACTModel.loss / compute_motion_generation_loss (fact_model.py:134-148; argument order (target, pred) as
 * the reference calls it, single_task_trainer.py:157): *loss_out (device float[1]) = mean((target - pred[:, :T])^2) with
 * target (B, T, D) and pred (B, n, D), T <= n.  The training step does not call it: fact_forward_backward fuses the
 * same kernel with the gradient of the loss. */
int fact_loss(const float* target, const float* pred, int B, int n, int T, int D, float* loss_out, void* stream);

/* Replaces the tape section of train_fn (single_task_trainer.py:141-178): forward, loss
 * (fact_model.py:143-148; target (B, T, out_dim)), backward.  Gradients of (loss * loss_scale)
 * are ACCUMULATED into the grad arena (loss_scale = 1/num_replicas, single_task_trainer.py:158).
 * `loss_out` (device float[1]) receives the unscaled mean-squared error. */
int fact_forward_backward(FactHandle* h, const float* motion, const float* audio,
                          const float* target, int B, int T, float loss_scale, float* loss_out,
                          void* stream);

/* Replaces optimizer.apply_gradients (single_task_trainer.py:186-187) with Keras-Adam semantics
 * (epsilon outside the bias correction), optional clip_by_global_norm (:180-183; clip_norm <= 0
 * disables; enabling it synchronises the stream once). Zeroes the grad arena, advances the step
 * counter and rewrites the bf16 weight shadows - one pass over p / m / v / g (36 bytes per parameter). */
int fact_adam_step(FactHandle* h, float lr, float beta1, float beta2, float eps, float clip_norm,
                   void* stream);
/* tf.clip_by_global_norm (single_task_trainer.py:180-183) on the gradient arena IN PLACE, as its own step: under data
 * parallelism every replica clips its OWN gradient before apply_gradients sums them (:180-187), i.e. before the
 * all-reduce and therefore outside fact_adam_step.  grads *= clip_norm / max(||grads||_2, clip_norm); two kernels, no host
 * synchronisation; `norm_out` (device float[1], may be NULL) receives the global norm.  (ABI 3) */
int fact_clip_gradients(FactHandle* h, float clip_norm, float* norm_out, void* stream);
/* Optimizer step inside the backward pass (no gradient clipping): call fact_adam_begin before
 * fact_forward_backward.  Without a gradient callback the engine updates the buckets itself on an internal
 * optimizer stream (Keras Adam + grad zeroing + bf16 shadows; joined before fact_forward_backward's work
 * completes on the caller's stream): the head and cross-modal buckets together once the last of them is
 * final - i.e. beside the backward of the two small encoder stacks, not beside the dense cross-modal
 * backward, which an HBM-bound update slows by more than it hides - then the encoder buckets.  With a
 * callback the host may call fact_adam_bucket(bucket, comm_stream) from its fact_grad_cb after that
 * bucket's all-reduce.  Same arithmetic as fact_adam_step. */
int fact_adam_begin(FactHandle* h, float lr, float beta1, float beta2, float eps);
int fact_adam_bucket(FactHandle* h, int bucket, void* stream);
/* Same, reading the bucket's gradients from `grads_bf16` - a bf16 array indexed like the fp32 arenas (the all-reduced
 * communication buffer of the data-parallel path with bf16 buckets) - instead of from the fp32 gradient arena, which
 * is still zeroed.  Saves the cast back into the arena (6 bytes per parameter). */
int fact_adam_bucket_bf16(FactHandle* h, int bucket, const void* grads_bf16, void* stream);
/* Disarm a fact_adam_begin whose fact_forward_backward never ran (host-side error): step counter restored. */
int fact_adam_cancel(FactHandle* h);
int fact_num_buckets(FactHandle* h, int* n);

/* Gradient-bucket casts for bf16 all-reduce payloads (data-parallel path): n elements, n % 4 == 0
 * (bucket ranges of the arenas are 64-float aligned).  Replaces nothing in the reference (TF all-reduces fp32). */
int fact_cast_f32_bf16(const float* src, void* dst_bf16, size_t n, void* stream);
int fact_cast_bf16_f32(const void* src_bf16, float* dst, size_t n, void* stream);

int fact_get_step(FactHandle* h, int64_t* step);
int fact_set_step(FactHandle* h, int64_t step);

/* Replaces FACTModel.infer_auto_regressive (fact_model.py:103-132): motion seed (B, n_m, F_m),
 * audio (B, audio_len, F_a) -> out (B, steps_done, out_dim) with row stride `steps` frames;
 * steps_done = min(steps, audio_len - n_a + 1) is returned through *steps_done.  Everything stays on the device
 * between frames (no host synchronisation); only row 0 of every step's output is computed past the last layer's
 * key / value projections (option "sr_rows"). */
int fact_infer_ar(FactHandle* h, const float* motion_seed, const float* audio, int B, int audio_len,
                  int steps, float* out, int* steps_done, void* stream);

/* Data-parallel overlap.  During fact_forward_backward the engine calls `cb(user, bucket, offset,
 * count)` (from the calling thread, while it is still enqueueing work) each time a contiguous
 * range grads[offset, offset+count) (floats) has all of its producers enqueued; `comm_stream` has
 * already been made to wait for them, so the host enqueues its sum-all-reduce of that range on
 * `comm_stream` (RCCL) and it overlaps the remaining backward kernels.  This is the explicit form
 * of the implicit all-reduce inside optimizer.apply_gradients under MirroredStrategy
 * (single_task_trainer.py:186-187).  Buckets: head, cross layers L-1..0, audio stack, motion stack.
 * The caller must make its compute stream wait for `comm_stream` before fact_adam_step.
 * cb == NULL disables. */
typedef void (*fact_grad_cb)(void* user, int bucket, size_t offset_floats, size_t count_floats);
int fact_set_grad_callback(FactHandle* h, fact_grad_cb cb, void* user, void* comm_stream);

/* Engine options (production surface; unknown keys return an error - the kernel-selection / scheduling A/B switches and
 * the timing-only ablation mask of the test and bench builds live behind fact_debug_set_option in fact_hip_debug.h and
 * are NOT accepted here):
 *   "sr_rows"        (default 1) 1 = the last cross-modal layer + head run on the rows that are kept: the B*T supervised
 *                        rows of a train step (fact_model.py:143-148) and the one generated row per sequence of
 *                        fact_infer_ar (fact_model.py:128), whose GEMMs of <= 512 rows are then cut along K (fp32 atomics:
 *                        equal up to summation order, not bit for bit from run to run); 0 = every layer on all rows.
 *                        fact_forward is never affected.
 *   "grad_overwrite" (default 0) 1 = fact_forward_backward WRITES the gradient of every transformer-layer Dense kernel
 *                        (plain stores of the whole-K grouped wgrad launch) instead of adding to it, and the optimizer
 *                        pass leaves those ranges as they are instead of zeroing them: 32 instead of 36 bytes per
 *                        parameter in the optimizer pass and no read-modify-write in the wgrad epilogue.  For hosts that
 *                        call fact_forward_backward exactly once per optimizer step (mint_amd/trainer.py sets it; the
 *                        reference's train_fn has no gradient accumulation either, single_task_trainer.py:141-196);
 *                        with 0 gradients accumulate across calls until fact_adam_step zeroes them.  Per handle: a new
 *                        handle starts at 0 with a zeroed gradient arena (caller-provided arenas included), and setting
 *                        it back to 0 clears the ranges that were being overwritten, so accumulation never starts on
 *                        a stale gradient.
 *   "adam_in_wgrad"  (default 0) 1 = in the engine-owned fused step (fact_adam_begin + fact_forward_backward, no gradient
 *                        callback) with "grad_overwrite" on, the Adam update and bf16-shadow refresh of every
 *                        transformer-layer Dense kernel run in the epilogue of the whole-K grouped wgrad launch that
 *                        holds its finished gradient in registers (28 bytes per parameter instead of 4 + 32; the
 *                        optimizer pass shrinks to biases, LayerNorm, position tables, embedding / head kernels).  The
 *                        gradient-arena ranges of those kernels are then NOT written by the step.  Same arithmetic as
 *                        the optimizer pass (Keras Adam, single_task_trainer.py:186-187 / trainer.py:150); takes effect
 *                        only when every stack runs the grouped launch (B * seq_len a multiple of 32 and >= 512, widths
 *                        multiples of 16), otherwise the step silently keeps the bucket path.  0 = always the bucket path.
 *                        At fact_v5 / batch 16 the wgrad launches are given 95 of 256 CUs beside the dgrad chain and the
 *                        update streams at half the rate of the 256-CU optimizer kernel: measured slower.
 *   "side_stream"    (default 1) 1 = the weight-gradient batches / the audio encoder run on the handle's second stream
 *   "aux_stream"     (default 1) 1 = the motion encoder's backward chain runs on the handle's third stream; hosts that
 *                        add a communication stream (data parallelism) set 0 to stay within the hardware queues */
int fact_set_option(FactHandle* h, const char* key, int value);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FACT_HIP_H_ */
