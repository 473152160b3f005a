// Fused (flash-style) full self-attention for FACT on gfx950: non-causal, unmasked,
// softmax scale = hidden_size^-0.5 (mint/core/base_models.py:66,82-85), MFMA 16x16x32 bf16.
//
// All score tiles are computed "transposed" (S^T = K Q^T) so that one query row lives in one
// lane column: row max / row sum are per-lane scalars plus two wave shuffles, and the bf16 P^T
// tile is directly the B operand of the O^T = V^T P^T MFMA (no LDS round trip for P).
//
// Per-head operand buffers (written by the QKV GEMM epilogue, EPI_HEADS):
//   *row : [B*H][NP][DHP]  token-major, head dim zero-padded to a multiple of 32,
//          NP = round_up(n, 128) with zero rows beyond n.
// Products that contract over tokens (P V, dS K, dS^T Q, P^T dO) need token-contiguous operand
// fragments; they are built from the same token-major LDS tiles with the gfx950 LDS transpose
// read (ds_read_b64_tr_b16), so no transposed copy of q/k/v/dO ever exists in HBM.
#pragma once
#include "common.h"

struct AttnParams {
  const bf16_t* qrow;
  const bf16_t* krow;
  const bf16_t* vrow;
  const bf16_t* dorow;
  bf16_t* out;        // fwd: attention output [B*n][hid] (b n (h d))
  const bf16_t* o;    // bwd: same tensor, read-only
  float* lse2;        // [B*H][NP] base-2 log-sum-exp of the scaled scores
  float* dsum;        // [B*H][NP] rowsum(dO * O)
  bf16_t* dqkv;       // bwd output [B*n][3*hid] in (qkv h d) column order
  int B, H, n, NP, hid, dh;
  int ldo, ldq;       // row pitches (elements) of out/o and of dqkv (>= hid, >= 3*hid)
  float scale;
  int nq;             // > 0: only query rows t < nq matter (supervised-rows shortcut of the last layer): forward computes
                      // those rows only; backward treats dO rows >= nq as zero (the LDS-resident kernels skip the work,
                      // the other families rely on the caller having zeroed those dO rows)
  int qblocks;        // filled by the launcher: 128-query blocks per head of the streaming forward's 1-D grid
};

int launch_attn_fwd(const AttnParams& p, hipStream_t s);
int launch_attn_bwd(const AttnParams& p, hipStream_t s);  // prep + dQ + dKdV
// kernel families (round 6: pruned): 5 (default) = streaming forward + lean LDS-resident backward where the head fits, streaming
// backward otherwise; 2 = streaming kernels everywhere; any other value selects 5.  attn_set_force_tiled(1) = the tiled reference path.
void attn_set_variant(int v);
int attn_get_variant();
void attn_set_force_tiled(int on);
