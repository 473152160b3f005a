// bf16 MFMA GEMM engine for gfx950: C[M][N] = sum_k A(m,k) * B(n,k), fp32 accumulate,
// pluggable fused epilogues.  Two operand-layout families:
//   NT : A[m*lda + k], B[n*ldb + k]            (both contraction-contiguous)   fwd / dgrad
//   TN : A[k*lda + m], B[k*ldb + n]            (both contraction-strided)      wgrad
// The TN kernel stages [k][m] tiles in LDS and builds MFMA fragments with the gfx950 LDS
// transpose read (ds_read_b64_tr_b16), so no transposed activation copies ever hit HBM.
#pragma once
#include "common.h"

enum EpiKind : int {
  EPI_BF16 = 0,           // out0 bf16 = acc
  EPI_F32_BIAS = 1,       // out0 f32  = acc + bias[n]
  EPI_F32_BIAS_POS = 2,   // out0 f32  = acc + bias[n] + pos[m % seq][n]
  EPI_F32_BIAS_RESID = 3, // out0 f32  = acc + bias[n] + resid[m][n]
  EPI_BIAS_GELU = 4,      // out0 bf16 = acc + bias[n] (pre-activation; not stored when out0 == nullptr), out1 bf16 = gelu(pre)
  EPI_GELU_BWD = 5,       // out0 bf16 = acc * gelu'(pre[m][n])
  EPI_HEADS = 6,          // scatter to per-head token-major q/k/v style buffers [B*H][n_pad][dhp]
  EPI_ATOMIC_F32 = 7,     // atomicAdd(out0 f32, alpha * acc)   (split-K wgrad)
  EPI_F32_BF16 = 8,       // out0 f32 = acc, out1 bf16 = acc
  EPI_F32_SLAB = 9,       // split-K partials: slab z at out0 + z*slab_stride, plain float4 stores
};

struct EpiParams {
  void* out0;
  int ldo0;
  void* out1;
  int ldo1;
  const float* bias;
  const float* pos;
  int seq;
  const float* resid;
  int ldr;
  const bf16_t* pre;
  int ldp;
  // EPI_HEADS: column c -> (which = c / hid, h = (c % hid) / dh, d = c % dh); row -> (b, t)
  bf16_t* hrow[3];  // [B*H][n_pad][dhp]
  int n_tok, n_pad, heads, dh, dhp, hid;
  float alpha;
  size_t slab_stride;  // EPI_F32_SLAB: floats between consecutive k-split slabs
  int z;               // (device) k-split index of this block, filled in by the kernel
  unsigned mg_hid, mg_dh, mg_ntok;  // EPI_HEADS in the big-tile kernels: 2^32 / d + 1 (filled by the launcher)
  // Optional LayerNorm of the output row fused into the skinny-M epilogue pass (EPI_F32_BIAS_RESID only, N <= 1024; ask
  // gemm_nt_takes_skinny first - other paths ignore it): ln_h bf16 [M][ln_ldh] = LN(out0 row) * ln_g + ln_b, and the row
  // statistics.  The next LayerNorm of the forward then costs no launch of its own.
  const float* ln_g;
  const float* ln_b;
  bf16_t* ln_h;
  int ln_ldh;
  float* ln_mean;
  float* ln_rstd;
  float ln_eps;
};

struct GemmParams {
  const bf16_t* A;
  int lda;
  const bf16_t* B;
  int ldb;
  int M, N, K;
  int splitk;
  int force_generic;  // tests: use the register-staged fallback kernel
  int band;           // NT tile order: band height in m-tiles (0 = default, 1 = row-major)
  // In-kernel split-K workspace of the big-tile NT kernels (optional): fp32 partial slabs for up to 256
  // workgroups of one launch and one arrival counter per tile (zero between launches).  One workspace per
  // stream: launches that may run concurrently must not share it.
  float* sk_slab;
  unsigned* sk_cnt;
  // Skinny-M path (optional): a ZERO-FILLED fp32 accumulator of >= M * round4(N) floats.  When present and
  // M <= 512 the GEMM runs as split-K 128x128 tiles with fp32 atomics into it (a 320-row GEMM has only 21
  // N = 800 tiles: the K loop is cut so that ~256 workgroups share it), followed by one element-wise kernel that
  // applies the fused epilogue and returns the accumulator to zero.  One stream at a time.
  float* skinny_acc;
  size_t skinny_floats;
  EpiParams ep;
};
constexpr size_t kSplitKSlabBytes = (size_t)256 * 288 * 256 * 4;  // 256 workgroups x the largest tile, fp32
constexpr int kSplitKCounters = 1024;

// ---- big-tile family (gemm_big.hip): one 8-wave workgroup per CU, 4-deep LDS-DMA ring -------------------
enum BigCfgId : int { BIG_288x256 = 0, BIG_256x256 = 1, BIG_256x160 = 2, BIG_160x256 = 3, BIG_256x128 = 4,
                      BIG_256x160_K64 = 5, BIG_288x256_K64 = 6, BIG_256x256_K64 = 7 /* 64-deep ring slots (whole-line DMA pieces) */,
                      BIG_192x160_K64 = 8 /* round 4: 150 instead of 115 tiles for the whole-K N = 800 dgrads at M = 5760 */,
                      BIG_128x160 = 9 /* round 5: short-K N = 800 GEMMs as 225 tiles, two workgroups per CU */,
                      BIG_256x256_M32 = 10, BIG_384x192_M32 = 11 /* round 6: v_mfma_f32_32x32x16_bf16 tiles on 64-deep slots (wave tile 128x64 / 96x96) */ };
int big_tile_dims(int cfg, int* bm, int* bn);
// NT GEMM on a given tile config (K % 32 == 0, splitk == 1); fused epilogues as above.
int launch_big_nt(int cfg, int epi, const GemmParams& p, hipStream_t stream);

// Grouped whole-K TN GEMM (weight gradients of one layer in one launch, 160x256 tiles):
//   out[m][n] (+)= sum_k A[k*lda + m] * B[k*ldb + n]           (trans_out = 0, out row pitch ldo >= N)
//   out[n][m] (+)= ...                                          (trans_out = 1, out row pitch ldo >= M)
constexpr int TN_GROUP_MAX = 4;
struct TnProblem {
  const bf16_t* A;
  const bf16_t* B;
  float* out;
  int lda, ldb, ldo;
  int M, N;        // A columns (the 160-tiled side), B columns (the 256-tiled side)
  int trans_out;
  int tiles_m, tile_begin;  // filled by the launcher
  // Fused optimizer step (TnGroup::adam; round 5).  The whole-K launch owns every element of `out` exactly once, so the
  // product IS the final gradient of that weight: instead of storing it, the epilogue applies the Keras-Adam update to the
  // fp32 master weight and both moments (same element index as `out`) and writes the two bf16 weight shadows of the
  // tensor - 28 bytes per parameter where "store gradient + optimizer pass" moves 4 + 32, and no optimizer pass left to
  // expose at the end of the step.  `out` is not written.
  float* p;      // fp32 master weights     [rows of out][ldo]
  float* m1;     // Adam first moment
  float* v;      // Adam second moment
  bf16_t* sd;    // bf16 shadow in out's orientation   [rows of out][ldsd]
  bf16_t* st;    // bf16 shadow, transposed            [cols of out][ldst]
  int ldsd, ldst;
  // Column sums of an operand the launch holds anyway (round 6; staggered main loop only, big_tn_group_has_colsum()): the
  // bias gradient of a Dense layer is the column sum over tokens of the gradient entering it, and that matrix IS an operand
  // of one of the layer's weight gradients (dpre -> dense_1 bias, the bf16 residual gradients -> dense_2 / to_out bias).
  //   trans_out = 0: csum[n] += sum_k B[k][n]  (n < N), by the workgroups of m-tile 0
  //   trans_out = 1: csum[m] += sum_k A[k][m]  (m < M), by the workgroups of n-tile 0
  // taken from the MFMA pipe (a constant ones fragment as the other operand: one extra 16-wide tile row per K step in one
  // wave row / column of those workgroups), left with fp32 atomics.  nullptr = off.
  float* csum;
};
struct TnGroup {
  TnProblem p[TN_GROUP_MAX];
  int n;
  int K;
  int tile0;  // first logical tile of this launch (filled by the launcher)
  int overwrite;  // 1: out = product (plain stores; every output element belongs to exactly one tile), 0: out += product
  int adam;       // 1: fused optimizer epilogue (TnProblem::p ...); every M, N a multiple of 16, overwrite semantics
  float lr_t, b1, b2, eps;  // Keras Adam: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t), epsilon outside the bias correction
};
// `parts` > 1 cuts the group's tiles into that many launches (same stream, in order) of about equal size: each
// then occupies only ~tiles/parts CUs, which leaves room for the CU-exclusive kernels of another stream.
int launch_big_tn_group(TnGroup g, hipStream_t stream, int parts = 1);
bool big_tn_group_has_colsum();  // does the selected main loop (gemm_set_tn_cfg) honour TnProblem::csum?
void gemm_set_tn_cfg(int v);  // 0 = 160x256 tiles (default), 1 = 160x384 tiles

// skinny-M epilogue pass (gemm_big.hip): out = epilogue(acc[M][ldacc]), acc <- 0
int launch_skinny_epilogue(int epi, float* acc, int ldacc, const GemmParams& p, hipStream_t stream);
// true when launch_gemm_nt(epi, p, .) will run the skinny-M path (split-K atomics + epilogue pass)
bool gemm_nt_takes_skinny(int epi, const GemmParams& p);

// Launchers. Return 0 on success, negative on invalid arguments.
int launch_gemm_nt(int epi, const GemmParams& p, hipStream_t stream);
int launch_gemm_tn(int epi, const GemmParams& p, hipStream_t stream);
void gemm_set_nt_variant(int v);  // 0 auto, 1 = 128x128, 10 / 11 / 12 / 14 / 17-23 = gemm_big.hip configs (gemm.hip launch_nt_t)
void gemm_set_tile192(int v);     // auto mode: 1 = 192x160 tiles for the whole-K N = 800 dgrads (BIG_192x160_K64)
void gemm_set_k64(int v);         // auto mode: 1 = 256x160 GEMMs run on 64-deep ring slots (BIG_256x160_K64)
void gemm_set_big_impl(int v);    // auto mode: 1 = gemm_big.hip family (default), 2 = without the 256x128 pairs, 0 = 128x128 kernel only
void gemm_set_tile128x160(int v);  // auto mode: 1 (default) = 128x160 tiles (two workgroups per CU) for the short-K N = 800 GEMMs, 0 = 256x128
void gemm_set_splitk_max(int v);  // in-kernel split-K of the 256x160 tile: max slices (default 4, 1 = off)
void gemm_set_nt_band(int band);   // NT tile band height (1 = row-major)
