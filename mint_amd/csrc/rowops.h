// HBM-bound row/elementwise kernels of the FACT train step (gfx950).
#pragma once
#include "common.h"

// LayerNorm forward (eps inside the rsqrt, biased variance; mint/core/base_models.py:27).
// x f32 [M][C] -> h bf16 [M][ldh]; saves mean/rstd f32 [M].  C % 4 == 0, C <= 2048.
// (bf16 activations use a row pitch that is a multiple of 64 elements so that every 128-byte K-step
//  chunk a GEMM stages is exactly one cache line.)
int launch_ln_fwd(const float* x, const float* gamma, const float* beta, bf16_t* h, int ldh, float* mean,
                  float* rstd, int M, int C, float eps, hipStream_t s);

// LayerNorm backward fused with the residual-gradient add:
//   dx = dres + LNbwd(dh);  dgamma += sum_rows dh*xhat;  dbeta += sum_rows dh;
//   dbias_prev += sum_rows dres (bias gradient of the GEMM whose output fed this residual add).
// dx may alias dres.  dx_bf16 / dbias_prev / dres may be null.  dh and dx_bf16 are bf16 with row
// pitch ld16; everything else is dense f32 [M][C].
// `ws` (>= ln_bwd_ws_floats(M, C) floats, or null) holds per-block partial column sums that a second
// tiny kernel reduces; with ws == null the partials are accumulated with atomics instead.
size_t ln_bwd_ws_floats(int M, int C);
void ln_set_bwd_rows(int rows_per_block);  // tuning knob (multiple of 4, >= 8 keeps ws sizing valid)
int launch_ln_bwd(const bf16_t* dh, const float* x, const float* mean, const float* rstd,
                  const float* gamma, const float* dres, float* dx, bf16_t* dx_bf16, float* dgamma,
                  float* dbeta, float* dbias_prev, float* ws, int M, int C, int ld16, hipStream_t s);

// The same backward split by what the dgrad chain waits for (round 2):
//   launch_ln_bwd_dx        dx = dres + LNbwd(dh) (+ bf16 copy), row-wise, nothing else;
//   launch_ln_param_grads   dgamma += sum_rows dh*xhat, dbeta += sum_rows dh, dbias += sum_rows dy (dy = the
//                           gradient that entered the residual add, bf16 [M][ldy] or f32; may be null) - column
//                           sums only the optimizer needs: the engine queues them on the wgrad stream.
int launch_ln_bwd_dx(const bf16_t* dh, const float* x, const float* mean, const float* rstd, const float* gamma,
                     const float* dres, float* dx, bf16_t* dx_bf16, int M, int C, int ld16, hipStream_t s);
// Round 4: the row-wise dx kernel that also leaves the three column sums as per-workgroup partial rows
// part[ln_cs_blocks(M)][3][C] (dh * xhat, dh, dres); launch_colreduce adds them into dgamma / dbeta / dbias later, off the
// dgrad chain.  C % 4 == 0, C <= 1024; `part` >= ln_cs_part_floats(M, C) floats.
int ln_cs_blocks(int M);
size_t ln_cs_part_floats(int M, int C);
void ln_set_cs_rows(int rows_per_wave);  // 2 or 4 (default)
int launch_ln_bwd_dx_cs(const bf16_t* dh, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* dres, float* dx, bf16_t* dx_bf16, float* part, int M, int C, int ld16, hipStream_t s);
int launch_colreduce(const float* part, int nblk, int C, float* o0, float* o1, float* o2, hipStream_t s);
int launch_ln_param_grads(const bf16_t* dh, int ld16, const float* x, const float* mean, const float* rstd,
                          const void* dy, int ldy, int dy_is_f32, float* dgamma, float* dbeta, float* dbias, int M,
                          int C, hipStream_t s);

// Up to 3 column-sum tasks over the same M rows in one launch: x != null -> LayerNorm parameter gradients of
// (dh, x, mean, rstd) plus dbias += colsum(dy) when dy is given; x == null -> dbias += colsum(dy) only.
constexpr int COL_TASKS_MAX = 3;
struct ColTask {
  const bf16_t* dh;
  int ld16;
  const float* x;
  const float* mean;
  const float* rstd;
  const bf16_t* dy;
  int ldy;
  float* dgamma;
  float* dbeta;
  float* dbias;
  int C;
  int cg_begin;  // filled by the launcher
  int cgw;       // filled by the launcher: columns per column group (<= 256, multiple of 4): C = 800 -> 4 groups of 200
};
struct ColTasks {
  ColTask t[COL_TASKS_MAX];
  int n, M, rpb;
};
int launch_col_tasks(ColTasks ts, hipStream_t s);

// out[c] += sum_m in[m][c]   (bf16 or f32 input), C % 8 == 0 for bf16, % 4 for f32
// only columns < Cout are accumulated into out
int launch_colsum_bf16(const bf16_t* in, int ld, float* out, int M, int C, int Cout, hipStream_t s);
int launch_colsum_f32(const float* in, int ld, float* out, int M, int C, int Cout, hipStream_t s);

// dpos[t][c] += sum_b dx[b][t][c]
int launch_possum(const float* dx, float* dpos, int B, int n, int C, hipStream_t s);

// MSE loss on the first T tokens (mint/core/fact_model.py:143-148):
//   loss_sum[0] += sum((target - pred[:, :T])^2) / (B*T*D)
//   dpred bf16 [B*n][ldp] = gscale * 2*(pred-target)/(B*T*D) on rows t<T, else 0 (pads too)
int launch_mse_loss(const float* pred, const float* target, float* loss_sum, bf16_t* dpred, int B,
                    int n, int T, int D, int ldp, float gscale, hipStream_t s);

// Keras Adam (epsilon outside the bias correction), fused with gradient zeroing:
//   m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= lr_t * m / (sqrt(v) + eps); g = 0
int launch_adam(float* p, float* m, float* v, float* g, size_t n, float lr_t, float b1, float b2,
                float eps, float gscale, hipStream_t s);

// Keras Adam FUSED with the bf16 weight-shadow refresh: one workgroup per AdamBlock.  A dense block is
// one 64x64 tile of a Dense kernel [R][C] at arena offset `off`: p/m/v/g are updated in place (g zeroed)
// and the new weights are written as bf16 to s [R][lds] and, through an LDS transpose, to t [C][ldt] -
// the master weights are not re-read by a separate cast pass.  A flat block (R == 0) is `C` floats
// (multiple of 4, <= 4096) of non-Dense parameters (biases, LayerNorm, position tables, arena padding).
// `blocks` lives in device memory; every block is independent.
struct AdamBlock {
  unsigned long long off;  // float offset into the four arenas (dense: of element [0][0] of the tensor)
  bf16_t* s;
  bf16_t* t;
  int R, C, lds, ldt;
  int r0, c0;
  int pad[4];
};
// `g16` != nullptr: gradients are read from that bf16 buffer (indexed like the arenas; the all-reduced bf16 bucket of
// the data-parallel path) instead of from `g`; `g` is still zeroed.
int launch_adam_fused(const AdamBlock* blocks, int nblocks, float* p, float* m, float* v, float* g,
                      float lr_t, float b1, float b2, float eps, float gscale, hipStream_t s,
                      const bf16_t* g16 = nullptr, int keep_overwritten = 0);
// `keep_overwritten`: blocks flagged AdamBlock::pad[0] & 1 (Dense kernels whose gradient the next step's wgrad launch
// overwrites) are not zeroed.

void adam_set_variant(int v);  // 0 round-2 kernel, 1 round-3 kernel, 2 round-3 kernel with non-temporal accesses
int adam_tile_width();         // 64 or 128 (FACT_ADAM_TW): column width of the dense AdamBlock tiles

// f32 [R][C] -> bf16 dst [R][ldd] and bf16 dstT [C][ldt] (either may be null)
int launch_cast_transpose(const float* src, int R, int C, bf16_t* dst, int ldd, bf16_t* dstT, int ldt,
                          hipStream_t s);
// One launch for a whole table of weight matrices: f32 [R][C] -> bf16 s [R][lds] and t [C][ldt].
// `descs` lives in device memory; tile_begin is the running sum of 64x64 tile counts.
struct CastDesc {
  const float* src;
  bf16_t* s;
  bf16_t* t;
  int R, C, lds, ldt, tiles_x, tile_begin;
};
int launch_multi_cast_transpose(const CastDesc* descs, int n, int total_tiles, hipStream_t s);
// bf16 [R][C] (ld) -> bf16 [C][R] (ldt)
int launch_transpose_bf16(const bf16_t* src, int ld, int R, int C, bf16_t* dstT, int ldt,
                          hipStream_t s);
// f32 rows (b, t) at src + b*batch_stride + t*F (M = B*n rows) -> bf16 [M][Fp] zero-padded
int launch_pad_cast(const float* src, int n, size_t batch_stride, int M, int F, bf16_t* dst, int Fp,
                    hipStream_t s);
// f32 -> bf16 flat, bf16 -> f32 flat (n % 4 == 0)
int launch_cast_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t s);
int launch_cast_f32(const bf16_t* src, float* dst, size_t n, hipStream_t s);
// out (B, na+nb, C) = concat along the sequence axis of a (B, na, C) and b (B, nb, C)
int launch_concat_seq(const float* a, const float* b, int B, int na, int nb, int C, float* out, hipStream_t s);
// split the cross-modal gradient (B, na+nb, C) into the two encoder gradients (f32 + bf16 copies)
int launch_split_grad(const float* dx, int B, int na, int nb, int C, float* da, bf16_t* da16,
                      float* db, bf16_t* db16, int ld16, hipStream_t s);
// Supervised-rows shortcut (last cross-modal layer): rows (b, t < T) of (B, n, .) tensors <-> compact (B*T, .):
//   gather : xc f32 [B*T][C] <- x f32 [B*n][C];  ac bf16 [B*T][ld16] <- a bf16 [B*n][ld16]  (either pair may be null)
//   scatter: dx f32 [B*n][C] <- dxc f32 [B*T][C] on rows t < T, zero elsewhere (all of dx is written)
int launch_gather_rows(const float* x, const bf16_t* a, int B, int n, int T, int C, int ld16, float* xc, bf16_t* ac,
                       hipStream_t s);
int launch_scatter_rows_zero(const float* dxc, int B, int n, int T, int C, float* dx, hipStream_t s);
// out[i] += sum_z slabs[z*stride + i], i < n (split-K partials -> gradient); stride % 4 == 0
int launch_slab_reduce(const float* slabs, size_t stride, int nslab, float* out, size_t n, hipStream_t s);
// sum of squares of a flat f32 buffer -> out[0] (atomicAdd), and flat scale
int launch_sumsq(const float* g, size_t n, float* out, hipStream_t s);
// g *= clip / max(sqrt(*sumsq), clip), *sumsq read on the device; norm_out (may be null) <- sqrt(*sumsq)
int launch_clip_scale(float* g, size_t n, const float* sumsq, float clip, float* norm_out, hipStream_t s);
