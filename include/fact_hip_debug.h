/* fact_hip_debug.h - test / bench surface of libfact_hip.so.  NOT part of the drop-in boundary (include/fact_hip.h is):
 * single-op entry points the parity tests drive the kernels through, layout probes, the in-step kernel-class recorder of
 * bench.py, process-wide kernel-selection switches and the string-keyed A/B options of the engine - one of which ("skip")
 * produces wrong results by design.  A host that binds the engine for training or inference needs none of this. */
#ifndef FACT_HIP_DEBUG_H_
#define FACT_HIP_DEBUG_H_

#include "fact_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
/* Exported only by the test / bench build of the library (mint_amd/lib/libfact_hip_dbg.so: engine.hip and probe.hip
 * compiled with -DFACT_DEBUG_ABI).  In the production libfact_hip.so these functions are compiled with hidden visibility:
 * `dlsym(lib, "fact_debug_set_option")` finds nothing there. */
#if defined(__GNUC__) || defined(__clang__)
#ifdef FACT_DEBUG_ABI
#pragma GCC visibility push(default)
#else
#pragma GCC visibility push(hidden)
#endif
#endif

/* In-step kernel-class timing.  fact_kprof(h, 1) arms it (and clears earlier records): every instrumented launch
 * site of the following forward / backward calls is bracketed by HIP events recorded on the stream it launches
 * on, with all the stream overlap of a normal step.  fact_kprof_read synchronises the device and returns, per
 * kernel class, the number of launches, the summed event time (ms), the algorithmic FLOPs and bytes.  Measurement
 * aid of bench.py, not part of the reference surface. */
int fact_kprof(FactHandle* h, int on);
int fact_kprof_dump(FactHandle* h, const char* path); /* CSV timeline: class, stream, start_us, end_us */
int fact_kprof_read(FactHandle* h, int max_classes, int* n_classes, const char** names, double* launches,
                    double* total_ms, double* flops, double* bytes);
/* The kernels behind class `cls` (index into fact_kprof_read's arrays) as the recorder saw them launched: text lines
 * "count\\tgrid\\tblock\\tlds_bytes\\tworkgroups_per_cu\\tkernel name" (most frequent first; the name is what rocprofv3 prints for
 * the same dispatch, workgroups_per_cu the runtime's occupancy answer for that launch shape) - bench.py names the symbol
 * that really ran and the CUs its grid can hold from this instead of from a hand-kept table. */
int fact_kprof_kernels(FactHandle* h, int cls, char* buf, int cap);

/* Test / bench knobs of one handle (several are process-wide, as noted).  Results are unchanged by every key except
 * "skip"; the production keys of fact_set_option are accepted too.
 *   "wgrad_tr"       1 = wgrad GEMM builds its fragments with the LDS transpose read, 0 = explicit transposes
 *   "wgrad_slab"     1 = split-K partials as plain stores + a streaming reduce, 0 = fp32 atomics
 *   "fuse_adam_cast" 1 = Adam writes the bf16 weight shadows itself, 0 = Adam, then a cast/transpose pass
 *   "bias_in_wgrad"  1 (default) = the dense_1 / dense_2 / to_out bias gradients are operand column sums inside the grouped
 *                    wgrad launch (gemm.h TnProblem::csum), 0 = col_tasks_kernel re-reads dpre and the residual gradients
 *   "wgrad_parts", "wgrad_defer", "wgrad_big", "ln_split", "ln_cs", "bwd_splitk", "adam_hold", "ln_fuse", "lite_stream", "keep_pre":
 *                    scheduling / fusion switches of the A/B runs documented in DESIGN.md sections 3 and 6
 *   "tn_loop", "attn_variant", "adam_variant", "big_impl", "tile192", "tile128x160", "k64" (bit 0: 256x160 GEMMs, bit 1: 288x256 / 256x256 GEMMs on
 *                    64-deep ring slots; default 3): PROCESS-WIDE kernel selection
 *   "skip":          TIMING-ONLY ablation mask: results are WRONG while it is set */
int fact_debug_set_option(FactHandle* h, const char* key, int value);

/* ---- single-op entry points (used by the parity tests; same kernels as the model path) ---- */
/* C = A(MxK) * B^T(NxK) ; epi selects the fused epilogue (see gemm.h); bf16 operands. */
int fact_op_gemm_nt(int epi, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                    void* out0, int ldo0, void* out1, int ldo1, const float* bias, const float* pos,
                    int seq, const float* resid, int ldr, const void* pre, int ldp, void* stream);
/* C(MoxNo) += A^T B with A [K][Mo], B [K][No] bf16 (wgrad form), f32 atomic accumulate. */
int fact_op_gemm_tn(const void* A, int lda, const void* B, int ldb, int Mo, int No, int K,
                    float* out, int ldo, int splitk, int use_tr, void* scratch, void* stream);
/* Grouped whole-K weight-gradient GEMM (one launch for the 1..4 wgrads of a transformer layer, 160x256 tiles,
 * no split-K): out_i[Mo_i][No_i] += A_i^T B_i with A_i bf16 [K][lda_i], B_i bf16 [K][ldb_i]; trans[i] = 1
 * stores out_i as [No_i][Mo_i].  K % 32 == 0, Mo / No / ldo % 4 == 0.  Replaces the tape's Dense-kernel
 * gradients (single_task_trainer.py:175-178 through base_models.py:51-53,68-69). */
int fact_op_gemm_tn_group(int n, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                          float* const* out, const int* ldo, const int* Mo, const int* No, const int* trans, int K,
                          void* stream);
/* The same launch that also leaves operand column sums (gemm.h TnProblem::csum; engine option "bias_in_wgrad"): csum[i] != NULL
 * adds sum_k B_i[k][n] (n < No_i) when trans[i] = 0, sum_k A_i[k][m] (m < Mo_i) when trans[i] = 1, with fp32 atomics - the
 * bias gradient of the Dense layer whose incoming gradient is that operand (single_task_trainer.py:175-178 through
 * base_models.py:51-53,69).  Staggered main loop only (fact_debug_gemm_tn_cfg low byte 0). */
int fact_op_gemm_tn_group_cs(int n, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                             float* const* out, const int* ldo, const int* Mo, const int* No, const int* trans,
                             float* const* csum, int K, void* stream);
/* The same launch with the optimizer in its epilogue (engine option "adam_in_wgrad", gemm.h TnGroup::adam): problem i
 * updates p_i / m_i / v_i (fp32, indexed like out_i would be: [Mo_i][No_i], or [No_i][Mo_i] when trans[i]) with Keras Adam
 * on the gradient A_i^T B_i and writes the bf16 shadows s_i (same orientation, row pitch lds_i) and t_i (transposed, row
 * pitch ldt_i).  Mo, No % 16 == 0.  Replaces, for the layer kernels, the tape's gradient + optimizer.apply_gradients
 * (single_task_trainer.py:175-187). */
int fact_op_gemm_tn_group_adam(int n, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                               float* const* p, float* const* m, float* const* v, void* const* s, const int* lds,
                               void* const* t, const int* ldt, const int* Mo, const int* No, const int* trans, int K,
                               float lr_t, float beta1, float beta2, float eps, void* stream);
int fact_op_ln_fwd(const float* x, const float* gamma, const float* beta, void* h, float* mean,
                   float* rstd, int M, int C, float eps, void* stream);
int fact_op_ln_bwd(const void* dh, const float* x, const float* mean, const float* rstd,
                   const float* gamma, const float* dres, float* dx, void* dx_bf16, float* dgamma,
                   float* dbeta, float* dbias_prev, int M, int C, void* stream);
/* qkv: bf16 [B*n][3*hid] in (qkv h d) column order -> out bf16 [B*n][hid]; if dout != NULL also
 * runs the backward and writes dqkv bf16 [B*n][3*hid].  `scratch` >= fact_op_attention_scratch(). */
size_t fact_op_attention_scratch(int B, int H, int n, int dh);
int fact_op_attention(const void* qkv, int B, int H, int n, int dh, float scale, void* out,
                      const void* dout, void* dqkv, void* scratch, void* stream);
int fact_op_adam(float* p, float* m, float* v, float* g, size_t n, float lr_t, float b1, float b2,
                 float eps, void* stream);
int fact_op_mse(const float* pred, const float* target, float* loss, void* dpred, int B, int n, int T,
                int D, int ldp, float gscale, void* stream);
/* MFMA / LDS-transpose-read layout probes (diagnostics). a_regs/b_regs: f32[64*8] per-lane operand
 * registers (rounded to bf16), d_regs: f32[64*4].  lds_vals: n<=4096 values placed in LDS as bf16,
 * byte_addrs: int[64] per-lane LDS byte address, out: f32[64*4] = what each lane received. */
int fact_probe_mfma(const float* a_regs, const float* b_regs, float* d_regs, void* stream);
int fact_probe_tr(const float* lds_vals, int n, const int* byte_addrs, float* out, void* stream);
/* Test knob: route every GEMM through the register-staged generic kernels (process-global). */
int fact_debug_force_generic_gemm(int on);
/* Test knob: 1 = the tiled reference attention kernels (standard online softmax) for every shape. */
int fact_debug_attn_force_tiled(int on);
/* Test/bench knob: attention kernel family. 5 (default) = streaming forward + lean LDS-resident backward where the head fits
 * (streaming backward otherwise); 2 = streaming 4-wave kernels everywhere; any other value selects the default. */
int fact_debug_attn_variant(int v);
int fact_debug_attn_variant_get(void); /* the current family (tests restore it) */
/* bench only: `nwg` one-per-CU workgroups that spin for ~`micros` microseconds on `stream` (CU-availability probe) */
int fact_debug_cu_hog(int nwg, int micros, void* stream);
/* Test/bench knob: NT GEMM kernel choice (0 auto, 1 = 128x128, 10.. = the tile configs of gemm_big.hip, see gemm.hip launch_nt_t). */
int fact_debug_gemm_splitk_max(int v); /* in-kernel split-K slices of the N = 800 GEMMs (1 = off, default 4) */
int fact_debug_gemm_tn_cfg(int v); /* grouped wgrad tile: 0 = 160x256, 1 = 160x384 */
int fact_debug_gemm_big_impl(int v); /* 1 = gemm_big.hip family (default), 2 = without the 256x128 pairs, 0 = 128x128 kernel only */
int fact_debug_gemm_nt_variant(int v);
/* Test/bench knob: NT GEMM tile band height (tile order inside an XCD; 1 = row-major, default 8). */
int fact_debug_gemm_nt_band(int band);
/* Test/bench knob of fact_op_ln_bwd: LayerNorm-backward rows per workgroup of the round-1 fused kernel (multiple of 4,
 * >= 8); use_ws: 0 = round-2 split (column-sum kernel + row-wise dx kernel), 1 / 2 = round-1 fused kernel with the
 * partial-sum workspace / with atomics, 3 / 4 = round-4 engine form (dx kernel leaving per-workgroup column-sum
 * partials, 4 / 2 rows per wave, + reduce), 5 = the row-wise dx kernel alone (bench). */
int fact_debug_ln_bwd(int rows_per_block, int use_ws);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FACT_HIP_DEBUG_H_ */
