// HBM-bound row/elementwise kernels (see rowops.h).  One wave (64 lanes) per token row for the
// LayerNorm kernels: float4 loads, wave-shuffle reductions, no LDS in the row statistics.
#include "rowops.h"

#include <cstdlib>

namespace {

constexpr int LN_MAXV = 8;  // float4 vectors per lane -> C <= 2048

// MAXV float4 per lane: C <= 256 * MAXV.  (The d = 800 model runs MAXV = 4: with 8 guarded trips the same kernel took
// 14.2 instead of 10.8 us at 11 520 rows - the AR sampler's batch of 32.)
template <int MAXV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     bf16_t* __restrict__ h, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, int C, int ldh, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = C >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * C);
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      v[i] = xr[idx];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  bf16x4* hr = reinterpret_cast<bf16x4*>(h + (size_t)row * ldh);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      const float4 g = g4[idx], b = b4[idx];
      bf16x4 o = {(bf16_t)((v[i].x - mu) * rs * g.x + b.x), (bf16_t)((v[i].y - mu) * rs * g.y + b.y),
                  (bf16_t)((v[i].z - mu) * rs * g.z + b.z), (bf16_t)((v[i].w - mu) * rs * g.w + b.w)};
      hr[idx] = o;
    }
  }
}

// Each block handles LN_BWD_ROWS rows (4 waves x LN_BWD_ROWS/4 rows); per-lane column partials
// of dgamma/dbeta/dbias are reduced across the 4 waves in LDS, then one atomicAdd per column.
int g_ln_bwd_rows = 8;  // rows per block (multiple of 4); fact_debug_ln_bwd_rows

template <int MAXV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(
    const bf16_t* __restrict__ dh, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* dres, float* dx,
    bf16_t* __restrict__ dx16, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dbias_prev, float* __restrict__ part, int M, int C, int ld16, int rpb) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [3][4 waves][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = C >> 2;
  const float invC = 1.0f / (float)C;
  float4 ag[MAXV], ab[MAXV], ar[MAXV], gm[MAXV];
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = ag[i];
    ar[i] = ag[i];
    const int idx = lane + i * 64;
    gm[i] = (idx < nv) ? g4[idx] : ag[i];
  }
  const int row_beg = blockIdx.x * rpb;
  for (int rr = wave; rr < rpb; rr += 4) {
    const int row = row_beg + rr;
    if (row >= M) break;
    const float mu = mean[row], rs = rstd[row];
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * C);
    const bf16x4* dhr = reinterpret_cast<const bf16x4*>(dh + (size_t)row * ld16);
    const float4* rr4 = dres ? reinterpret_cast<const float4*>(dres + (size_t)row * C) : nullptr;
    // all three input rows are requested before the first reduction: the residual-gradient row is not
    // needed until after the two wave reductions, so its latency hides behind them
    float4 xv[MAXV], rv[MAXV];
    bf16x4 dv[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int idx = lane + i * 64;
      if (idx < nv) {
        xv[i] = xr[idx];
        dv[i] = dhr[idx];
        rv[i] = rr4 ? rr4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 xh[MAXV], dy[MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int idx = lane + i * 64;
      if (idx < nv) {
        const float d0 = (float)dv[i][0], d1 = (float)dv[i][1], d2 = (float)dv[i][2], d3 = (float)dv[i][3];
        xh[i] = make_float4((xv[i].x - mu) * rs, (xv[i].y - mu) * rs, (xv[i].z - mu) * rs, (xv[i].w - mu) * rs);
        dy[i] = make_float4(d0 * gm[i].x, d1 * gm[i].y, d2 * gm[i].z, d3 * gm[i].w);
        s1 += (dy[i].x + dy[i].y) + (dy[i].z + dy[i].w);
        s2 += (dy[i].x * xh[i].x + dy[i].y * xh[i].y) + (dy[i].z * xh[i].z + dy[i].w * xh[i].w);
        ag[i].x += d0 * xh[i].x; ag[i].y += d1 * xh[i].y; ag[i].z += d2 * xh[i].z; ag[i].w += d3 * xh[i].w;
        ab[i].x += d0; ab[i].y += d1; ab[i].z += d2; ab[i].w += d3;
      }
    }
    s1 = wave_sum(s1) * invC;
    s2 = wave_sum(s2) * invC;
    float4* dxr = reinterpret_cast<float4*>(dx + (size_t)row * C);
    bf16x4* dx16r = dx16 ? reinterpret_cast<bf16x4*>(dx16 + (size_t)row * ld16) : nullptr;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int idx = lane + i * 64;
      if (idx < nv) {
        float4 o = make_float4(rs * (dy[i].x - s1 - xh[i].x * s2), rs * (dy[i].y - s1 - xh[i].y * s2),
                               rs * (dy[i].z - s1 - xh[i].z * s2), rs * (dy[i].w - s1 - xh[i].w * s2));
        if (rr4) {
          const float4 r = rv[i];
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
          ar[i].x += r.x; ar[i].y += r.y; ar[i].z += r.z; ar[i].w += r.w;
        }
        dxr[idx] = o;
        if (dx16r) {
          bf16x4 o16 = {(bf16_t)o.x, (bf16_t)o.y, (bf16_t)o.z, (bf16_t)o.w};
          dx16r[idx] = o16;
        }
      }
    }
  }
  // cross-wave reduction
  float4* r4 = reinterpret_cast<float4*>(red);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      r4[(0 * 4 + wave) * nv + idx] = ag[i];
      r4[(1 * 4 + wave) * nv + idx] = ab[i];
      r4[(2 * 4 + wave) * nv + idx] = ar[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f, r = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[(0 * 4 + w) * C + c];
      b += red[(1 * 4 + w) * C + c];
      r += red[(2 * 4 + w) * C + c];
    }
    if (part) {  // per-block partial sums, reduced by colreduce_kernel (no same-address atomics)
      float* pb = part + (size_t)blockIdx.x * 3 * C;
      pb[c] = a;
      pb[C + c] = b;
      pb[2 * C + c] = r;
    } else {
      atomicAdd(dgamma + c, a);
      atomicAdd(dbeta + c, b);
      if (dbias_prev && dres) atomicAdd(dbias_prev + c, r);
    }
  }
}

// ---- round 2: LayerNorm backward split by what is on the critical path ----------------------------------
// dx (the residual-stream gradient) is the only output the next kernel of the backward chain waits for; the
// parameter gradients (dgamma, dbeta, the previous bias) are column sums that only the optimizer reads.
//   ln_bwd_dx_kernel      row-wise, one wave per row, no column accumulators / LDS / second pass: ~60 VGPRs, so
//                         its waves fit beside the 184-VGPR wgrad workgroups that run on the same CUs (round-2
//                         timeline: the fused kernel took 53 us beside the wgrad launch, 20 us alone);
//   ln_param_grads_kernel column-wise (a lane owns 4 columns, a wave 256), runs on the wgrad stream.
template <int MAXV>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const bf16_t* __restrict__ dh, const float* __restrict__ x,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* dres, float* dx,
                                                        bf16_t* __restrict__ dx16, int M, int C, int ld16) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = C >> 2;
  const float invC = 1.0f / (float)C;
  const float mu = mean[row], rs = rstd[row];
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * C);
  const bf16x4* dhr = reinterpret_cast<const bf16x4*>(dh + (size_t)row * ld16);
  const float4* rr4 = dres ? reinterpret_cast<const float4*>(dres + (size_t)row * C) : nullptr;
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  float4 xv[MAXV], rv[MAXV], gm[MAXV];
  bf16x4 dv[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      xv[i] = xr[idx];
      dv[i] = dhr[idx];
      rv[i] = rr4 ? rr4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
      gm[i] = g4[idx];
    }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      // xv <- xhat, gm <- dy = dh * gamma
      xv[i] = make_float4((xv[i].x - mu) * rs, (xv[i].y - mu) * rs, (xv[i].z - mu) * rs, (xv[i].w - mu) * rs);
      gm[i] = make_float4((float)dv[i][0] * gm[i].x, (float)dv[i][1] * gm[i].y, (float)dv[i][2] * gm[i].z,
                          (float)dv[i][3] * gm[i].w);
      s1 += (gm[i].x + gm[i].y) + (gm[i].z + gm[i].w);
      s2 += (gm[i].x * xv[i].x + gm[i].y * xv[i].y) + (gm[i].z * xv[i].z + gm[i].w * xv[i].w);
    }
  }
  s1 = wave_sum(s1) * invC;
  s2 = wave_sum(s2) * invC;
  float4* dxr = reinterpret_cast<float4*>(dx + (size_t)row * C);
  bf16x4* dx16r = dx16 ? reinterpret_cast<bf16x4*>(dx16 + (size_t)row * ld16) : nullptr;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      const float4 o = make_float4(rv[i].x + rs * (gm[i].x - s1 - xv[i].x * s2), rv[i].y + rs * (gm[i].y - s1 - xv[i].y * s2),
                                   rv[i].z + rs * (gm[i].z - s1 - xv[i].z * s2), rv[i].w + rs * (gm[i].w - s1 - xv[i].w * s2));
      dxr[idx] = o;
      if (dx16r) {
        bf16x4 o16 = {(bf16_t)o.x, (bf16_t)o.y, (bf16_t)o.z, (bf16_t)o.w};
        dx16r[idx] = o16;
      }
    }
  }
}

// ---- round 4: the row-wise dx kernel ALSO produces the column sums only the optimizer reads, as per-workgroup
// partials.  The split of round 2 kept the dgrad chain short (the old fused kernel above runs 37 us in the step against
// 22 us: two rows per wave behind each other, 38 KiB of LDS, 2400 atomics per workgroup or a reduce launch ON the chain)
// but its column-sum pass re-reads dh, x and the entering gradient - 74 MB per layer of the 109 MB col_tasks_kernel
// moves, 46-50 us per layer on the wgrad stream and HBM traffic beside the GELU' dgrad.  Here a workgroup owns 4 * RPW
// consecutive rows (a wave RPW of them, two in flight at a time), every lane keeps the three column accumulators of its
// 4 * MAXV columns in registers while it walks its rows, and at the end the four waves' accumulators meet in LDS and leave
// as ONE plain-store row part[block][3][C] (no atomics, nothing waits for them).  A tiny reduce over the blocks
// (colreduce_kernel) runs later on the wgrad stream: 3.5 MB per LayerNorm instead of 37 MB.
//   part[b][0][c] = sum_rows dh * xhat   part[b][1][c] = sum_rows dh   part[b][2][c] = sum_rows dres (0 without dres)
template <int MAXV, int RPW>
__global__ __launch_bounds__(256) void ln_bwd_dx_cs_kernel(const bf16_t* __restrict__ dh, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* dres, float* dx,
                                                           bf16_t* __restrict__ dx16, float* __restrict__ part, int M, int C,
                                                           int ld16) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][3][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = C >> 2;
  const float invC = 1.0f / (float)C;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ag[MAXV], ab[MAXV], ar[MAXV], gmm[MAXV];
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    ag[i] = z4; ab[i] = z4; ar[i] = z4;
    const int idx = lane + i * 64;
    gmm[i] = (idx < nv) ? g4[idx] : z4;
  }
  const int row0 = blockIdx.x * (4 * RPW) + wave * RPW;
  static_assert(RPW % 2 == 0, "rows go through the wave two at a time");
#pragma unroll 1
  for (int k0 = 0; k0 < RPW; k0 += 2) {
    float4 xv[2][MAXV], rv[2][MAXV];
    bf16x4 dv[2][MAXV];
    float mu[2], rs[2];
    // both rows' loads are requested before the first use (a clamped duplicate row past the end is loaded, not used)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = min(row0 + k0 + k, M - 1);
      mu[k] = mean[row];
      rs[k] = rstd[row];
      const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * C);
      const bf16x4* dhr = reinterpret_cast<const bf16x4*>(dh + (size_t)row * ld16);
      const float4* rr4 = dres ? reinterpret_cast<const float4*>(dres + (size_t)row * C) : nullptr;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        if (idx < nv) {
          xv[k][i] = xr[idx];
          dv[k][i] = dhr[idx];
          rv[k][i] = rr4 ? rr4[idx] : z4;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = row0 + k0 + k;
      if (row >= M) break;  // wave-uniform
      float4 dy[MAXV];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        if (idx < nv) {
          const float d0 = (float)dv[k][i][0], d1 = (float)dv[k][i][1], d2 = (float)dv[k][i][2], d3 = (float)dv[k][i][3];
          float4& xh = xv[k][i];
          xh = make_float4((xh.x - mu[k]) * rs[k], (xh.y - mu[k]) * rs[k], (xh.z - mu[k]) * rs[k], (xh.w - mu[k]) * rs[k]);
          dy[i] = make_float4(d0 * gmm[i].x, d1 * gmm[i].y, d2 * gmm[i].z, d3 * gmm[i].w);
          s1 += (dy[i].x + dy[i].y) + (dy[i].z + dy[i].w);
          s2 += (dy[i].x * xh.x + dy[i].y * xh.y) + (dy[i].z * xh.z + dy[i].w * xh.w);
          ag[i].x += d0 * xh.x; ag[i].y += d1 * xh.y; ag[i].z += d2 * xh.z; ag[i].w += d3 * xh.w;
          ab[i].x += d0; ab[i].y += d1; ab[i].z += d2; ab[i].w += d3;
          const float4 r = rv[k][i];
          ar[i].x += r.x; ar[i].y += r.y; ar[i].z += r.z; ar[i].w += r.w;
        }
      }
      s1 = wave_sum(s1) * invC;
      s2 = wave_sum(s2) * invC;
      float4* dxr = reinterpret_cast<float4*>(dx + (size_t)row * C);
      bf16x4* dx16r = dx16 ? reinterpret_cast<bf16x4*>(dx16 + (size_t)row * ld16) : nullptr;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        if (idx < nv) {
          const float4 xh = xv[k][i], r = rv[k][i];
          const float4 o = make_float4(r.x + rs[k] * (dy[i].x - s1 - xh.x * s2), r.y + rs[k] * (dy[i].y - s1 - xh.y * s2),
                                       r.z + rs[k] * (dy[i].z - s1 - xh.z * s2), r.w + rs[k] * (dy[i].w - s1 - xh.w * s2));
          dxr[idx] = o;
          if (dx16r) {
            bf16x4 o16 = {(bf16_t)o.x, (bf16_t)o.y, (bf16_t)o.z, (bf16_t)o.w};
            dx16r[idx] = o16;
          }
        }
      }
    }
  }
  // the four waves' column accumulators -> one partial row of this block
  float4* r4 = reinterpret_cast<float4*>(red);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      r4[(wave * 3 + 0) * nv + idx] = ag[i];
      r4[(wave * 3 + 1) * nv + idx] = ab[i];
      r4[(wave * 3 + 2) * nv + idx] = ar[i];
    }
  }
  __syncthreads();
  float4* p4 = reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 3 * C);
  const int n4 = 3 * nv;
  for (int j = threadIdx.x; j < n4; j += 256) {
    const float4 a = r4[j], b = r4[n4 + j], c = r4[2 * n4 + j], d = r4[3 * n4 + j];
    p4[j] = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z),
                        (a.w + b.w) + (c.w + d.w));
  }
}

// dgamma[c] += sum_r dh[r][c] * xhat[r][c];  dbeta[c] += sum_r dh[r][c];  dbias[c] += sum_r dy[r][c]
// grid (ceil(C / 256), ceil(M / rows_per_block)), block 256 = 4 waves; a lane owns 4 consecutive columns, the
// waves of a block interleave the rows of its chunk; one LDS reduction and 3 atomics per column and block.
template <typename TY>
__global__ __launch_bounds__(256) void ln_param_grads_kernel(const bf16_t* __restrict__ dh, int ld16,
                                                             const float* __restrict__ x,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const TY* __restrict__ dy,
                                                             int ldy, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ dbias,
                                                             int M, int C, int rpb) {
  __shared__ float red[3][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 256 + lane * 4;
  const int r_beg = blockIdx.y * rpb, r_end = min(M, r_beg + rpb);
  float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f}, ar[4] = {0.f, 0.f, 0.f, 0.f};
  if (c0 < C) {
    // 8 rows of this wave per trip, every load of the trip requested before the first use (the kernel has
    // < 3 waves per CU: bytes in flight per wave, not occupancy, have to cover the HBM latency)
    constexpr int U = 8;
    for (int r0 = r_beg + wave; r0 < r_end; r0 += 4 * U) {
      bf16x4 dv[U];
      float4 xv[U];
      float mu[U], rs[U];
      bf16x4 yb[U];
      float4 yf[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = min(r0 + 4 * u, M - 1);
        dv[u] = *reinterpret_cast<const bf16x4*>(dh + (size_t)r * ld16 + c0);
        xv[u] = *reinterpret_cast<const float4*>(x + (size_t)r * C + c0);
        mu[u] = mean[r];
        rs[u] = rstd[r];
        if (dy) {
          if constexpr (sizeof(TY) == 2) yb[u] = *reinterpret_cast<const bf16x4*>(dy + (size_t)r * ldy + c0);
          else yf[u] = *reinterpret_cast<const float4*>(dy + (size_t)r * ldy + c0);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (r0 + 4 * u >= r_end) break;
        const float d0 = (float)dv[u][0], d1 = (float)dv[u][1], d2 = (float)dv[u][2], d3 = (float)dv[u][3];
        ag[0] += d0 * ((xv[u].x - mu[u]) * rs[u]); ag[1] += d1 * ((xv[u].y - mu[u]) * rs[u]);
        ag[2] += d2 * ((xv[u].z - mu[u]) * rs[u]); ag[3] += d3 * ((xv[u].w - mu[u]) * rs[u]);
        ab[0] += d0; ab[1] += d1; ab[2] += d2; ab[3] += d3;
        if (dy) {
          if constexpr (sizeof(TY) == 2) {
            ar[0] += (float)yb[u][0]; ar[1] += (float)yb[u][1]; ar[2] += (float)yb[u][2]; ar[3] += (float)yb[u][3];
          } else {
            ar[0] += yf[u].x; ar[1] += yf[u].y; ar[2] += yf[u].z; ar[3] += yf[u].w;
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][wave][lane * 4 + j] = ag[j];
    red[1][wave][lane * 4 + j] = ab[j];
    red[2][wave][lane * 4 + j] = ar[j];
  }
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    const int t = threadIdx.x;
    atomicAdd(dgamma + c, (red[0][0][t] + red[0][1][t]) + (red[0][2][t] + red[0][3][t]));
    atomicAdd(dbeta + c, (red[1][0][t] + red[1][1][t]) + (red[1][2][t] + red[1][3][t]));
    if (dy && dbias) atomicAdd(dbias + c, (red[2][0][t] + red[2][1][t]) + (red[2][2][t] + red[2][3][t]));
  }
}

// All column sums of one layer's optimizer-only batch in ONE launch (ColTasks): per task either the LayerNorm
// parameter gradients (+ the bias gradient from dy) as above, or a plain column sum of dy (x == nullptr).
// grid (total column groups of 256, ceil(M / rpb)).
__global__ __launch_bounds__(256) void col_tasks_kernel(const ColTasks ts) {
  __shared__ float red[3][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int ti = 0;
#pragma unroll
  for (int i = 1; i < COL_TASKS_MAX; ++i)
    if (i < ts.n && (int)blockIdx.x >= ts.t[i].cg_begin) ti = i;
  ColTask t = ts.t[0];
#pragma unroll
  for (int i = 1; i < COL_TASKS_MAX; ++i)
    if (ti == i) t = ts.t[i];
  const int cg = blockIdx.x - t.cg_begin;
  const int c0 = cg * t.cgw + lane * 4;
  const int r_beg = blockIdx.y * ts.rpb, r_end = min(ts.M, r_beg + ts.rpb);
  const bool ln = t.x != nullptr;
  float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f}, ar[4] = {0.f, 0.f, 0.f, 0.f};
  if (lane * 4 < t.cgw && c0 < t.C) {
    constexpr int U = 8;
    for (int r0 = r_beg + wave; r0 < r_end; r0 += 4 * U) {
      bf16x4 dv[U], yb[U];
      float4 xv[U];
      float mu[U], rs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = min(r0 + 4 * u, ts.M - 1);
        if (ln) {
          dv[u] = *reinterpret_cast<const bf16x4*>(t.dh + (size_t)r * t.ld16 + c0);
          xv[u] = *reinterpret_cast<const float4*>(t.x + (size_t)r * t.C + c0);
          mu[u] = t.mean[r];
          rs[u] = t.rstd[r];
        }
        if (t.dy) yb[u] = *reinterpret_cast<const bf16x4*>(t.dy + (size_t)r * t.ldy + c0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (r0 + 4 * u >= r_end) break;
        if (ln) {
          const float d0 = (float)dv[u][0], d1 = (float)dv[u][1], d2 = (float)dv[u][2], d3 = (float)dv[u][3];
          ag[0] += d0 * ((xv[u].x - mu[u]) * rs[u]); ag[1] += d1 * ((xv[u].y - mu[u]) * rs[u]);
          ag[2] += d2 * ((xv[u].z - mu[u]) * rs[u]); ag[3] += d3 * ((xv[u].w - mu[u]) * rs[u]);
          ab[0] += d0; ab[1] += d1; ab[2] += d2; ab[3] += d3;
        }
        if (t.dy) {
          ar[0] += (float)yb[u][0]; ar[1] += (float)yb[u][1]; ar[2] += (float)yb[u][2]; ar[3] += (float)yb[u][3];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][wave][lane * 4 + j] = ag[j];
    red[1][wave][lane * 4 + j] = ab[j];
    red[2][wave][lane * 4 + j] = ar[j];
  }
  __syncthreads();
  const int c = cg * t.cgw + threadIdx.x;
  if ((int)threadIdx.x < t.cgw && c < t.C) {
    const int k = threadIdx.x;
    if (ln) {
      atomicAdd(t.dgamma + c, (red[0][0][k] + red[0][1][k]) + (red[0][2][k] + red[0][3][k]));
      atomicAdd(t.dbeta + c, (red[1][0][k] + red[1][1][k]) + (red[1][2][k] + red[1][3][k]));
    }
    if (t.dy && t.dbias) atomicAdd(t.dbias + c, (red[2][0][k] + red[2][1][k]) + (red[2][2][k] + red[2][3][k]));
  }
}

// out_k[c] += sum over blocks of part[b][k][c], k = 0..2; grid (ceil(3C/256), ceil(nblk/32))
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ part, int nblk, int C,
                                                        float* __restrict__ o0, float* __restrict__ o1,
                                                        float* __restrict__ o2) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= 3 * C) return;
  const int b0 = blockIdx.y * 32, b1 = min(nblk, b0 + 32);
  float s = 0.f;
#pragma unroll 8
  for (int b = b0; b < b1; ++b) s += part[(size_t)b * 3 * C + j];
  const int k = j / C, c = j - k * C;
  float* o = (k == 0) ? o0 : (k == 1) ? o1 : o2;
  if (o) atomicAdd(o + c, s);
}

// column sums: block = 32 column-chunks x 8 row lanes; each thread owns VEC columns
template <typename T, int VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ in, int ld,
                                                     float* __restrict__ out, int M, int C,
                                                     int Cout, int rows_per_block) {
  __shared__ float red[8][32 * VEC + 1];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c0 = (blockIdx.x * 32 + cx) * VEC;
  const int r_beg = blockIdx.y * rows_per_block;
  const int r_end = min(M, r_beg + rows_per_block);
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (c0 < C) {
    for (int r = r_beg + ry; r < r_end; r += 8) {
      const T* p = in + (size_t)r * ld + c0;
      if constexpr (sizeof(T) == 2) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += (float)v[j];
      } else {
        const float4 v = *reinterpret_cast<const float4*>(p);
        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[ry][cx * VEC + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < 32 * VEC; c += 256) {
    const int gc = blockIdx.x * 32 * VEC + c;
    if (gc < Cout) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += red[w][c];
      atomicAdd(out + gc, s);
    }
  }
}

__global__ void possum_kernel(const float* __restrict__ dx, float* __restrict__ dpos, int B,
                              size_t nC) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nC) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += dx[(size_t)b * nC + i];
  dpos[i] += s;
}

__global__ __launch_bounds__(256) void mse_loss_kernel(const float* __restrict__ pred,
                                                       const float* __restrict__ target,
                                                       float* __restrict__ loss_sum,
                                                       bf16_t* __restrict__ dpred, int B, int n, int T,
                                                       int D, int ldp, float inv_count, float gscale) {
  // one block per (b, t) row of dpred
  const int row = blockIdx.x;
  const int b = row / n, t = row - b * n;
  float part = 0.f;
  for (int c = threadIdx.x; c < ldp; c += blockDim.x) {
    float g = 0.f;
    if (t < T && c < D) {
      const float diff = pred[(size_t)row * D + c] - target[((size_t)b * T + t) * D + c];
      part += diff * diff;
      g = 2.0f * diff * inv_count * gscale;
    }
    if (dpred) dpred[(size_t)row * ldp + c] = (bf16_t)g;
  }
  if (t < T) {
    __shared__ float red[4];
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
      atomicAdd(loss_sum, s * inv_count);
    }
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, float4* __restrict__ m,
                                                   float4* __restrict__ v, float4* __restrict__ g,
                                                   size_t n4, float lr_t, float b1, float b2, float eps,
                                                   float gscale) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 gv = g[i], mv = m[i], vv = v[i], pv = p[i];
    gv.x *= gscale; gv.y *= gscale; gv.z *= gscale; gv.w *= gscale;
    mv.x = b1 * mv.x + (1.f - b1) * gv.x; mv.y = b1 * mv.y + (1.f - b1) * gv.y;
    mv.z = b1 * mv.z + (1.f - b1) * gv.z; mv.w = b1 * mv.w + (1.f - b1) * gv.w;
    vv.x = b2 * vv.x + (1.f - b2) * gv.x * gv.x; vv.y = b2 * vv.y + (1.f - b2) * gv.y * gv.y;
    vv.z = b2 * vv.z + (1.f - b2) * gv.z * gv.z; vv.w = b2 * vv.w + (1.f - b2) * gv.w * gv.w;
    pv.x -= lr_t * mv.x / (sqrtf(vv.x) + eps); pv.y -= lr_t * mv.y / (sqrtf(vv.y) + eps);
    pv.z -= lr_t * mv.z / (sqrtf(vv.z) + eps); pv.w -= lr_t * mv.w / (sqrtf(vv.w) + eps);
    p[i] = pv; m[i] = mv; v[i] = vv; g[i] = z;
  }
}

// adam1: common.h (shared with the fused optimizer epilogue of the grouped wgrad kernel, gemm_big.hip)

// G16 != nullptr (data parallel with bf16 gradient buckets): the gradient is read from the all-reduced bf16 buffer
// (indexed like the fp32 arena) instead of from G - no cast back into the arena; G is still zeroed for the next step.
DEVINL float4 adam_grad4(const float* G, const bf16_t* G16, size_t o) {
  if (G16) {
    const bf16x4 g = *reinterpret_cast<const bf16x4*>(G16 + o);
    return make_float4((float)g[0], (float)g[1], (float)g[2], (float)g[3]);
  }
  return *reinterpret_cast<const float4*>(G + o);
}

__global__ __launch_bounds__(256) void adam_fused_kernel(const AdamBlock* __restrict__ blocks,
                                                         float* __restrict__ P, float* __restrict__ Mm,
                                                         float* __restrict__ V, float* __restrict__ G,
                                                         const bf16_t* __restrict__ G16, float lr_t, float b1,
                                                         float b2, float eps, float gscale) {
  __shared__ float tile[64][65];
  const AdamBlock d = blocks[blockIdx.x];
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d.R == 0) {  // flat segment
    for (int i = threadIdx.x * 4; i < d.C; i += 1024) {
      const size_t o = d.off + i;
      float4 gv = adam_grad4(G, G16, o), mv = *reinterpret_cast<const float4*>(Mm + o);
      float4 vv = *reinterpret_cast<const float4*>(V + o), pv = *reinterpret_cast<const float4*>(P + o);
      adam1(pv.x, mv.x, vv.x, gv.x * gscale, lr_t, b1, b2, eps);
      adam1(pv.y, mv.y, vv.y, gv.y * gscale, lr_t, b1, b2, eps);
      adam1(pv.z, mv.z, vv.z, gv.z * gscale, lr_t, b1, b2, eps);
      adam1(pv.w, mv.w, vv.w, gv.w * gscale, lr_t, b1, b2, eps);
      *reinterpret_cast<float4*>(P + o) = pv;
      *reinterpret_cast<float4*>(Mm + o) = mv;
      *reinterpret_cast<float4*>(V + o) = vv;
      *reinterpret_cast<float4*>(G + o) = z;
    }
    return;
  }
  const int cx = (threadIdx.x & 15) * 4, ry = threadIdx.x >> 4;  // 16 column-quads x 16 rows
  const bool vec_ok = ((d.C & 3) == 0);
#pragma unroll
  for (int rr = ry; rr < 64; rr += 16) {
    const int r = d.r0 + rr, c = d.c0 + cx;
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < d.R) {
      const size_t o = d.off + (size_t)r * d.C + c;
      if (vec_ok && c + 3 < d.C) {
        // (a tensor whose offset + row pitch is not 8-byte aligned in bf16 terms cannot occur: offsets are 64-float
        //  aligned and this branch requires C % 4 == 0)
        float4 gv = adam_grad4(G, G16, o), mv = *reinterpret_cast<const float4*>(Mm + o);
        float4 vv = *reinterpret_cast<const float4*>(V + o), pv = *reinterpret_cast<const float4*>(P + o);
        adam1(pv.x, mv.x, vv.x, gv.x * gscale, lr_t, b1, b2, eps);
        adam1(pv.y, mv.y, vv.y, gv.y * gscale, lr_t, b1, b2, eps);
        adam1(pv.z, mv.z, vv.z, gv.z * gscale, lr_t, b1, b2, eps);
        adam1(pv.w, mv.w, vv.w, gv.w * gscale, lr_t, b1, b2, eps);
        *reinterpret_cast<float4*>(P + o) = pv;
        *reinterpret_cast<float4*>(Mm + o) = mv;
        *reinterpret_cast<float4*>(V + o) = vv;
        *reinterpret_cast<float4*>(G + o) = z;
        w[0] = pv.x; w[1] = pv.y; w[2] = pv.z; w[3] = pv.w;
        bf16x4 o16 = {(bf16_t)w[0], (bf16_t)w[1], (bf16_t)w[2], (bf16_t)w[3]};
        *reinterpret_cast<bf16x4*>(d.s + (size_t)r * d.lds + c) = o16;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < d.C) {
            float pv = P[o + j], mv = Mm[o + j], vv = V[o + j];
            adam1(pv, mv, vv, (G16 ? (float)G16[o + j] : G[o + j]) * gscale, lr_t, b1, b2, eps);
            P[o + j] = pv; Mm[o + j] = mv; V[o + j] = vv; G[o + j] = 0.f;
            w[j] = pv;
            d.s[(size_t)r * d.lds + c + j] = (bf16_t)pv;
          }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[rr][cx + j] = w[j];
  }
  __syncthreads();
  // transposed shadow: thread -> column cc, 4 consecutive rows
  const int rq = (threadIdx.x & 15) * 4, cy = threadIdx.x >> 4;
#pragma unroll
  for (int cc = cy; cc < 64; cc += 16) {
    const int c = d.c0 + cc, r = d.r0 + rq;
    if (c < d.C) {
      if (r + 3 < d.R) {
        bf16x4 o = {(bf16_t)tile[rq][cc], (bf16_t)tile[rq + 1][cc], (bf16_t)tile[rq + 2][cc],
                    (bf16_t)tile[rq + 3][cc]};
        *reinterpret_cast<bf16x4*>(d.t + (size_t)c * d.ldt + r) = o;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (r + j < d.R) d.t[(size_t)c * d.ldt + r + j] = (bf16_t)tile[rq + j][cc];
      }
    }
  }
}

// Round 3: the same pass with (a) non-temporal loads / stores on the four fp32 streams (each byte is touched once per
// step), (b) every load of a group of row passes issued before the first use, (c) 64 x TW tiles (TW = 64 or 128: 256- or
// 512-byte row runs) and (d) no zero write-back of the gradient where the block says its producer overwrites it
// (AdamBlock::pad[0] & 1: Dense kernels whose weight gradient comes from the whole-K grouped wgrad launch, engine option
// grad_overwrite).  tools/attic/adam_probe.hip: a flat kernel with today's access mix streams 5.85 TB/s, 6.3 with non-temporal
// accesses, and the 32 B/param mix (no zeroing) finishes in 0.63-0.64 ms against 0.74 for 36 B/param.
DEVINL f32x4 ld4(const float* p, bool nt) {
  return nt ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)) : *reinterpret_cast<const f32x4*>(p);
}
DEVINL void st4(float* p, f32x4 v, bool nt) {
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<f32x4*>(p) = v;
}
template <int TW, bool NT>
__global__ __launch_bounds__(256) void adam_fused2_kernel(const AdamBlock* __restrict__ blocks, float* __restrict__ P,
                                                          float* __restrict__ Mm, float* __restrict__ V,
                                                          float* __restrict__ G, const bf16_t* __restrict__ G16,
                                                          float lr_t, float b1, float b2, float eps, float gscale,
                                                          int honor_keep) {
  __shared__ float tile[64][TW + 1];
  const AdamBlock d = blocks[blockIdx.x];
  const bool keep_g = honor_keep && (d.pad[0] & 1) != 0;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  auto grad4 = [&](size_t o) -> f32x4 {
    if (G16) {
      const bf16x4 g = *reinterpret_cast<const bf16x4*>(G16 + o);
      return f32x4{(float)g[0], (float)g[1], (float)g[2], (float)g[3]};
    }
    return ld4(G + o, NT);
  };
  auto upd4 = [&](f32x4& pv, f32x4& mv, f32x4& vv, f32x4 gv) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float pj = pv[j], mj = mv[j], vj = vv[j];
      adam1(pj, mj, vj, gv[j] * gscale, lr_t, b1, b2, eps);
      pv[j] = pj; mv[j] = mj; vv[j] = vj;
    }
  };
  if (d.R == 0) {  // flat segment (<= 4096 floats): biases, LayerNorm, tables, padding - always zeroed (atomics feed them)
    for (int i = threadIdx.x * 4; i < d.C; i += 1024) {
      const size_t o = d.off + i;
      f32x4 gv = grad4(o), mv = ld4(Mm + o, NT), vv = ld4(V + o, NT), pv = ld4(P + o, NT);
      upd4(pv, mv, vv, gv);
      st4(P + o, pv, NT); st4(Mm + o, mv, NT); st4(V + o, vv, NT); st4(G + o, z, NT);
    }
    return;
  }
  constexpr int CQ = TW / 4, RP = 256 / CQ, NPASS = 64 / RP, GRP = 4;  // column quads per row, rows per pass, passes
  static_assert(NPASS % GRP == 0, "pass groups");
  const int cx = (threadIdx.x % CQ) * 4, ry = threadIdx.x / CQ;
  const bool vec_ok = ((d.C & 3) == 0);
  if (vec_ok) {
#pragma unroll
    for (int g0 = 0; g0 < NPASS; g0 += GRP) {
      f32x4 gv[GRP], mv[GRP], vv[GRP], pv[GRP];
      bool ok[GRP];
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
        const int r = d.r0 + ry + (g0 + u) * RP, c = d.c0 + cx;
        ok[u] = r < d.R && c + 3 < d.C;
        if (ok[u]) {
          const size_t o = d.off + (size_t)r * d.C + c;
          gv[u] = grad4(o); mv[u] = ld4(Mm + o, NT); vv[u] = ld4(V + o, NT); pv[u] = ld4(P + o, NT);
        }
      }
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
        const int rr = ry + (g0 + u) * RP;
        f32x4 w = z;
        if (ok[u]) {
          const int r = d.r0 + rr, c = d.c0 + cx;
          const size_t o = d.off + (size_t)r * d.C + c;
          upd4(pv[u], mv[u], vv[u], gv[u]);
          st4(P + o, pv[u], NT); st4(Mm + o, mv[u], NT); st4(V + o, vv[u], NT);
          if (!keep_g) st4(G + o, z, NT);
          w = pv[u];
          bf16x4 o16 = {(bf16_t)w[0], (bf16_t)w[1], (bf16_t)w[2], (bf16_t)w[3]};
          *reinterpret_cast<bf16x4*>(d.s + (size_t)r * d.lds + c) = o16;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[rr][cx + j] = w[j];
      }
    }
  } else {
    for (int rr = ry; rr < 64; rr += RP) {
      const int r = d.r0 + rr, c = d.c0 + cx;
      float w[4] = {0.f, 0.f, 0.f, 0.f};
      if (r < d.R) {
        const size_t o = d.off + (size_t)r * d.C + c;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < d.C) {
            float pv = P[o + j], mv = Mm[o + j], vv = V[o + j];
            adam1(pv, mv, vv, (G16 ? (float)G16[o + j] : G[o + j]) * gscale, lr_t, b1, b2, eps);
            P[o + j] = pv; Mm[o + j] = mv; V[o + j] = vv;
            if (!keep_g) G[o + j] = 0.f;
            w[j] = pv;
            d.s[(size_t)r * d.lds + c + j] = (bf16_t)pv;
          }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[rr][cx + j] = w[j];
    }
  }
  __syncthreads();
  // transposed shadow: thread -> column cc, 4 consecutive rows (16 threads cover the 64 rows of a column = 128 bytes)
  const int rq = (threadIdx.x & 15) * 4, cy = threadIdx.x >> 4;
#pragma unroll
  for (int cc = cy; cc < TW; cc += 16) {
    const int c = d.c0 + cc, r = d.r0 + rq;
    if (c < d.C) {
      if (r + 3 < d.R) {
        bf16x4 o = {(bf16_t)tile[rq][cc], (bf16_t)tile[rq + 1][cc], (bf16_t)tile[rq + 2][cc],
                    (bf16_t)tile[rq + 3][cc]};
        *reinterpret_cast<bf16x4*>(d.t + (size_t)c * d.ldt + r) = o;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (r + j < d.R) d.t[(size_t)c * d.ldt + r + j] = (bf16_t)tile[rq + j][cc];
      }
    }
  }
}

// 64x64 tile cast/transpose through LDS
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const T* __restrict__ src, int lds_,
                                                             int R, int C, bf16_t* __restrict__ dst,
                                                             int ldd, bf16_t* __restrict__ dstT,
                                                             int ldt) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) {
      v = (float)src[(size_t)r * lds_ + c];
      if (dst) dst[(size_t)r * ldd + c] = (bf16_t)v;
    }
    tile[rr][tx] = v;
  }
  if (!dstT) return;
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, r = r0 + tx;
    if (c < C && r < R) dstT[(size_t)c * ldt + r] = (bf16_t)tile[tx][cc];
  }
}

// multi-tensor weight-shadow refresh: block -> (tensor, 64x64 tile); float4 reads, 8-byte bf16 stores
// in both orientations (the transposed one through an LDS tile).
__global__ __launch_bounds__(256) void multi_cast_transpose_kernel(const CastDesc* __restrict__ descs,
                                                                   int n) {
  __shared__ float tile[64][65];
  __shared__ int which;
  if (threadIdx.x == 0) {
    int lo = 0, hi = n - 1;  // last desc with tile_begin <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (descs[mid].tile_begin <= (int)blockIdx.x) lo = mid;
      else hi = mid - 1;
    }
    which = lo;
  }
  __syncthreads();
  const CastDesc d = descs[which];
  const int tl = blockIdx.x - d.tile_begin;
  const int r0 = (tl / d.tiles_x) * 64, c0 = (tl % d.tiles_x) * 64;
  const int cx = (threadIdx.x & 15) * 4, ry = threadIdx.x >> 4;  // 16 column-quads x 16 rows
  const bool vec_ok = ((d.C & 3) == 0);
#pragma unroll
  for (int rr = ry; rr < 64; rr += 16) {
    const int r = r0 + rr, c = c0 + cx;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < d.R) {
      if (vec_ok && c + 3 < d.C) {
        const float4 f = *reinterpret_cast<const float4*>(d.src + (size_t)r * d.C + c);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
        bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(d.s + (size_t)r * d.lds + c) = o;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < d.C) {
            v[j] = d.src[(size_t)r * d.C + c + j];
            d.s[(size_t)r * d.lds + c + j] = (bf16_t)v[j];
          }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[rr][cx + j] = v[j];
  }
  __syncthreads();
  // transposed: thread -> column cc, 4 consecutive rows
  const int rq = (threadIdx.x & 15) * 4, cy = threadIdx.x >> 4;
#pragma unroll
  for (int cc = cy; cc < 64; cc += 16) {
    const int c = c0 + cc, r = r0 + rq;
    if (c < d.C) {
      if (r + 3 < d.R) {
        bf16x4 o = {(bf16_t)tile[rq][cc], (bf16_t)tile[rq + 1][cc], (bf16_t)tile[rq + 2][cc],
                    (bf16_t)tile[rq + 3][cc]};
        *reinterpret_cast<bf16x4*>(d.t + (size_t)c * d.ldt + r) = o;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (r + j < d.R) d.t[(size_t)c * d.ldt + r + j] = (bf16_t)tile[rq + j][cc];
      }
    }
  }
}

__global__ void pad_cast_kernel(const float* __restrict__ src, int n, size_t batch_stride, int M,
                                int F, bf16_t* __restrict__ dst, int Fp) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * Fp) return;
  const int r = (int)(i / Fp), c = (int)(i - (size_t)r * Fp);
  const int b = r / n, t = r - b * n;
  dst[i] = (c < F) ? (bf16_t)src[(size_t)b * batch_stride + (size_t)t * F + c] : (bf16_t)0.f;
}

__global__ void cast_bf16_kernel(const float4* __restrict__ src, bf16x4* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = src[i];
    bf16x4 o = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
    dst[i] = o;
  }
}

__global__ void cast_f32_kernel(const bf16x4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const bf16x4 v = src[i];
    dst[i] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
}

// tf.concat([a, b], axis=1): out (B, na+nb, C) from a (B, na, C) and b (B, nb, C)
__global__ void concat_seq_kernel(const float4* __restrict__ a, const float4* __restrict__ b, int B, int na,
                                  int nb, int C4, float4* __restrict__ out) {
  const size_t total = (size_t)B * (na + nb) * C4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t row = i / C4;
    const int c = (int)(i - row * C4);
    const int bi = (int)(row / (na + nb));
    const int t = (int)(row - (size_t)bi * (na + nb));
    out[i] = (t < na) ? a[((size_t)bi * na + t) * C4 + c] : b[((size_t)bi * nb + (t - na)) * C4 + c];
  }
}

// Supervised-rows shortcut of the last cross-modal layer (engine.hip): rows (b, t < T) of a (B, n, .) tensor
// <-> a compact (B*T, .) tensor.
//   gather : xc f32 [B*T][C] = x[(b*n + t)][C], ac bf16 [B*T][ld16] = a[(b*n + t)][ld16 ...]   (either may be null)
__global__ void gather_rows_kernel(const float4* __restrict__ x, const bf16x4* __restrict__ a, int B, int n, int T,
                                   int C4, int ld4, float4* __restrict__ xc, bf16x4* __restrict__ ac) {
  const size_t total = (size_t)B * T * C4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t row = i / C4;
    const int c = (int)(i - row * C4);
    const int b = (int)(row / T), t = (int)(row - (size_t)b * T);
    const size_t src = (size_t)b * n + t;
    if (x) xc[row * C4 + c] = x[src * C4 + c];
    if (a) ac[row * ld4 + c] = a[src * ld4 + c];
  }
}
//   scatter: dx f32 [B*n][C] = dxc[(b*T + t)] on rows t < T, 0 elsewhere (every row of dx is written)
__global__ void scatter_rows_zero_kernel(const float4* __restrict__ dxc, int B, int n, int T, int C4,
                                         float4* __restrict__ dx) {
  const size_t total = (size_t)B * n * C4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t row = i / C4;
    const int c = (int)(i - row * C4);
    const int b = (int)(row / n), t = (int)(row - (size_t)b * n);
    dx[i] = (t < T) ? dxc[((size_t)b * T + t) * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

__global__ void split_grad_kernel(const float4* __restrict__ dx, int B, int na, int nb, int C4, int ld4,
                                  float4* __restrict__ da, bf16x4* __restrict__ da16,
                                  float4* __restrict__ db, bf16x4* __restrict__ db16) {
  const size_t total = (size_t)B * (na + nb) * C4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t row = i / C4;
    const int c = (int)(i - row * C4);
    const int b = (int)(row / (na + nb));
    const int t = (int)(row - (size_t)b * (na + nb));
    const float4 v = dx[i];
    bf16x4 o = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
    if (t < na) {
      const size_t r = (size_t)b * na + t;
      da[r * C4 + c] = v;
      da16[r * ld4 + c] = o;
    } else {
      const size_t r = (size_t)b * nb + (t - na);
      db[r * C4 + c] = v;
      db16[r * ld4 + c] = o;
    }
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float4* __restrict__ g, size_t n4,
                                                    float* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = g[i];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// tf.clip_by_global_norm, second half: g *= clip / max(sqrt(sumsq), clip) with the sum of squares read ON THE DEVICE (no host
// round trip between the two kernels); norm_out (optional) receives the global norm
__global__ __launch_bounds__(256) void clip_scale_kernel(float4* __restrict__ g, size_t n4, const float* __restrict__ sumsq,
                                                         float clip, float* __restrict__ norm_out) {
  const float norm = sqrtf(*sumsq);
  const float sc = clip / fmaxf(norm, clip);
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = norm;
  if (sc == 1.0f) return;  // uniform: nothing to clip
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = g[i];
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    g[i] = v;
  }
}

// out[i] += sum_z slabs[z*stride + i]  (split-K partial sums of a wgrad GEMM)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, size_t stride,
                                                          int nslab, float* __restrict__ out, size_t n4) {
  const size_t gs = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gs) {
    float4 a = reinterpret_cast<float4*>(out)[i];
    for (int z = 0; z < nslab; ++z) {
      const float4 v = reinterpret_cast<const float4*>(slabs + (size_t)z * stride)[i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}
__global__ void slab_reduce_tail_kernel(const float* __restrict__ slabs, size_t stride, int nslab,
                                        float* __restrict__ out, size_t beg, size_t n) {
  const size_t i = beg + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = out[i];
  for (int z = 0; z < nslab; ++z) a += slabs[(size_t)z * stride + i];
  out[i] = a;
}

inline int grid_for(size_t n, int block, int cap = 4096) {
  size_t g = (n + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

int launch_ln_fwd(const float* x, const float* gamma, const float* beta, bf16_t* h, int ldh, float* mean,
                  float* rstd, int M, int C, float eps, hipStream_t s) {
  if ((C & 3) || C > 64 * 4 * LN_MAXV || M <= 0 || ldh < C || (ldh & 3)) return -1;
  if (C <= 1024)
    FACT_LAUNCH((ln_fwd_kernel<4>), dim3((M + 3) / 4), dim3(256), 0, s, x, gamma, beta, h, mean, rstd, M, C, ldh, eps);
  else if (C <= 1536)
    FACT_LAUNCH((ln_fwd_kernel<6>), dim3((M + 3) / 4), dim3(256), 0, s, x, gamma, beta, h, mean, rstd, M, C, ldh, eps);
  else
    FACT_LAUNCH((ln_fwd_kernel<LN_MAXV>), dim3((M + 3) / 4), dim3(256), 0, s, x, gamma, beta, h, mean, rstd, M, C, ldh,
                       eps);
  return 0;
}

void ln_set_bwd_rows(int r) { g_ln_bwd_rows = r < 8 ? 8 : r; }

size_t ln_bwd_ws_floats(int M, int C) {
  return (size_t)((M + 7) / 8) * 3 * C;  // sized for the smallest rows-per-block (8)
}

int launch_ln_bwd(const bf16_t* dh, const float* x, const float* mean, const float* rstd,
                  const float* gamma, const float* dres, float* dx, bf16_t* dx_bf16, float* dgamma,
                  float* dbeta, float* dbias_prev, float* ws, int M, int C, int ld16, hipStream_t s) {
  if ((C & 3) || C > 64 * 4 * LN_MAXV || M <= 0 || ld16 < C || (ld16 & 3)) return -1;
  const int rpb = g_ln_bwd_rows;
  const int grid = (M + rpb - 1) / rpb;
  const size_t shmem = (size_t)3 * 4 * C * sizeof(float);
  if (C <= 1024) {
    FACT_LAUNCH((ln_bwd_kernel<4>), dim3(grid), dim3(256), shmem, s, dh, x, mean, rstd, gamma,
                       dres, dx, dx_bf16, dgamma, dbeta, dbias_prev, ws, M, C, ld16, rpb);
  } else {
    FACT_LAUNCH((ln_bwd_kernel<8>), dim3(grid), dim3(256), shmem, s, dh, x, mean, rstd, gamma,
                       dres, dx, dx_bf16, dgamma, dbeta, dbias_prev, ws, M, C, ld16, rpb);
  }
  if (ws) {
    dim3 g2((3 * C + 255) / 256, (grid + 31) / 32);
    FACT_LAUNCH(colreduce_kernel, g2, dim3(256), 0, s, ws, grid, C, dgamma, dbeta,
                       (dres ? dbias_prev : nullptr));
  }
  return 0;
}

int launch_ln_bwd_dx(const bf16_t* dh, const float* x, const float* mean, const float* rstd, const float* gamma,
                     const float* dres, float* dx, bf16_t* dx_bf16, int M, int C, int ld16, hipStream_t s) {
  if ((C & 3) || C > 64 * 4 * LN_MAXV || M <= 0 || ld16 < C || (ld16 & 3)) return -1;
  const int grid = (M + 3) / 4;
  if (C <= 1024)
    FACT_LAUNCH((ln_bwd_dx_kernel<4>), dim3(grid), dim3(256), 0, s, dh, x, mean, rstd, gamma, dres, dx, dx_bf16,
                       M, C, ld16);
  else if (C <= 1536)
    FACT_LAUNCH((ln_bwd_dx_kernel<6>), dim3(grid), dim3(256), 0, s, dh, x, mean, rstd, gamma, dres, dx, dx_bf16,
                       M, C, ld16);
  else
    FACT_LAUNCH((ln_bwd_dx_kernel<8>), dim3(grid), dim3(256), 0, s, dh, x, mean, rstd, gamma, dres, dx, dx_bf16,
                       M, C, ld16);
  return 0;
}

int g_ln_cs_rpw = 4;  // rows per wave of ln_bwd_dx_cs_kernel (2 or 4): test / bench knob
void ln_set_cs_rows(int rpw) { g_ln_cs_rpw = (rpw == 2) ? 2 : 4; }
int ln_cs_blocks(int M) { return (M + 4 * g_ln_cs_rpw - 1) / (4 * g_ln_cs_rpw); }
size_t ln_cs_part_floats(int M, int C) { return (size_t)((M + 7) / 8) * 3 * C; }  // sized for 2 rows per wave

int launch_ln_bwd_dx_cs(const bf16_t* dh, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* dres, float* dx, bf16_t* dx_bf16, float* part, int M, int C, int ld16,
                        hipStream_t s) {
  if ((C & 3) || C > 1024 || M <= 0 || ld16 < C || (ld16 & 3) || !part) return -1;  // wider rows: the split kernels
  const int grid = ln_cs_blocks(M);
  const size_t shmem = (size_t)4 * 3 * C * sizeof(float);
  if (g_ln_cs_rpw == 2)
    FACT_LAUNCH((ln_bwd_dx_cs_kernel<4, 2>), dim3(grid), dim3(256), shmem, s, dh, x, mean, rstd, gamma, dres, dx, dx_bf16,
                part, M, C, ld16);
  else
    FACT_LAUNCH((ln_bwd_dx_cs_kernel<4, 4>), dim3(grid), dim3(256), shmem, s, dh, x, mean, rstd, gamma, dres, dx, dx_bf16,
                part, M, C, ld16);
  return 0;
}

// o_k[c] += sum over the nblk partial rows part[b][k][c], k = 0..2 (any o_k may be null)
int launch_colreduce(const float* part, int nblk, int C, float* o0, float* o1, float* o2, hipStream_t s) {
  if (nblk <= 0 || C <= 0) return -1;
  dim3 g2((3 * C + 255) / 256, (nblk + 31) / 32);
  FACT_LAUNCH(colreduce_kernel, g2, dim3(256), 0, s, part, nblk, C, o0, o1, o2);
  return 0;
}

int launch_ln_param_grads(const bf16_t* dh, int ld16, const float* x, const float* mean, const float* rstd,
                          const void* dy, int ldy, int dy_is_f32, float* dgamma, float* dbeta, float* dbias, int M,
                          int C, hipStream_t s) {
  if ((C & 3) || M <= 0 || ld16 < C || (ld16 & 3) || (dy && (ldy < C || (ldy & 3)))) return -1;
  const int rpb = 128;
  dim3 grid((C + 255) / 256, (M + rpb - 1) / rpb);
  if (dy_is_f32)
    FACT_LAUNCH((ln_param_grads_kernel<float>), grid, dim3(256), 0, s, dh, ld16, x, mean, rstd, (const float*)dy,
                       ldy, dgamma, dbeta, dbias, M, C, rpb);
  else
    FACT_LAUNCH((ln_param_grads_kernel<bf16_t>), grid, dim3(256), 0, s, dh, ld16, x, mean, rstd,
                       (const bf16_t*)dy, ldy, dgamma, dbeta, dbias, M, C, rpb);
  return 0;
}

int launch_col_tasks(ColTasks ts, hipStream_t s) {
  if (ts.n < 1 || ts.n > COL_TASKS_MAX || ts.M <= 0) return -1;
  int groups = 0;
  for (int i = 0; i < ts.n; ++i) {
    ColTask& t = ts.t[i];
    if ((t.C & 3) || (t.dy && (t.ldy < t.C || (t.ldy & 3))) || (t.x && (t.ld16 < t.C || (t.ld16 & 3)))) return -1;
    t.cg_begin = groups;
    const int ng = (t.C + 255) / 256;
    static const bool even = !(getenv("FACT_COL_EVEN") && atoi(getenv("FACT_COL_EVEN")) == 0);  // A/B knob
    t.cgw = even ? ((((t.C + ng - 1) / ng) + 3) & ~3) : 256;  // even column groups: 800 -> 4 x 200 (not 3 x 256 + 32: an eighth-filled group)
    groups += ng;
  }
  // rows per workgroup.  Round 6 A/B with only the two LayerNorm tasks left in the launch (bias_in_wgrad): 64 rows (720 workgroups)
  // make this kernel 42 -> 32 us and the dgrad GEMMs beside it 2-4 us slower each - step unchanged (profiles/r06_ab_col_rpb.txt)
  ts.rpb = 128;
  FACT_LAUNCH(col_tasks_kernel, dim3(groups, (ts.M + ts.rpb - 1) / ts.rpb), dim3(256), 0, s, ts);
  return 0;
}

int launch_colsum_bf16(const bf16_t* in, int ld, float* out, int M, int C, int Cout, hipStream_t s) {
  if ((C & 7) || (ld & 7)) return -1;
  const int rpb = 128;
  dim3 grid((C + 255) / 256, (M + rpb - 1) / rpb);
  FACT_LAUNCH((colsum_kernel<bf16_t, 8>), grid, dim3(256), 0, s, in, ld, out, M, C, Cout, rpb);
  return 0;
}

int launch_colsum_f32(const float* in, int ld, float* out, int M, int C, int Cout, hipStream_t s) {
  if ((C & 3) || (ld & 3)) return -1;
  const int rpb = 128;
  dim3 grid((C + 127) / 128, (M + rpb - 1) / rpb);
  FACT_LAUNCH((colsum_kernel<float, 4>), grid, dim3(256), 0, s, in, ld, out, M, C, Cout, rpb);
  return 0;
}

int launch_possum(const float* dx, float* dpos, int B, int n, int C, hipStream_t s) {
  const size_t nC = (size_t)n * C;
  FACT_LAUNCH(possum_kernel, dim3((unsigned)((nC + 255) / 256)), dim3(256), 0, s, dx, dpos, B, nC);
  return 0;
}

int launch_mse_loss(const float* pred, const float* target, float* loss_sum, bf16_t* dpred, int B,
                    int n, int T, int D, int ldp, float gscale, hipStream_t s) {
  if (T > n || D > ldp) return -1;
  const float inv_count = 1.0f / ((float)B * (float)T * (float)D);
  FACT_LAUNCH(mse_loss_kernel, dim3(B * n), dim3(256), 0, s, pred, target, loss_sum, dpred, B, n,
                     T, D, ldp, inv_count, gscale);
  return 0;
}

int launch_adam(float* p, float* m, float* v, float* g, size_t n, float lr_t, float b1, float b2,
                float eps, float gscale, hipStream_t s) {
  if (n & 3) return -1;
  const size_t n4 = n >> 2;
  FACT_LAUNCH(adam_kernel, dim3(grid_for(n4, 256, 8192)), dim3(256), 0, s, (float4*)p,
                     (float4*)m, (float4*)v, (float4*)g, n4, lr_t, b1, b2, eps, gscale);
  return 0;
}

int g_adam_variant = 1, g_adam_tw = 64;
int launch_adam_fused(const AdamBlock* blocks, int nblocks, float* p, float* m, float* v, float* g,
                      float lr_t, float b1, float b2, float eps, float gscale, hipStream_t s, const bf16_t* g16,
                      int keep) {
  if (nblocks <= 0) return 0;
  // g_adam_variant: 0 = round-2 kernel (64x64 tiles), 1 = round-3 kernel plain accesses, 2 = ... non-temporal;
  // the tile width is a property of the block table (adam_tile_width()), fixed when the handle is created
  if (g_adam_variant == 0 && g_adam_tw == 64 && !keep)
    FACT_LAUNCH(adam_fused_kernel, dim3(nblocks), dim3(256), 0, s, blocks, p, m, v, g, g16, lr_t, b1, b2, eps,
                       gscale);
  else if (g_adam_tw == 128) {
    if (g_adam_variant == 2)
      FACT_LAUNCH((adam_fused2_kernel<128, true>), dim3(nblocks), dim3(256), 0, s, blocks, p, m, v, g, g16, lr_t,
                         b1, b2, eps, gscale, keep);
    else
      FACT_LAUNCH((adam_fused2_kernel<128, false>), dim3(nblocks), dim3(256), 0, s, blocks, p, m, v, g, g16, lr_t,
                         b1, b2, eps, gscale, keep);
  } else {
    if (g_adam_variant == 2)
      FACT_LAUNCH((adam_fused2_kernel<64, true>), dim3(nblocks), dim3(256), 0, s, blocks, p, m, v, g, g16, lr_t,
                         b1, b2, eps, gscale, keep);
    else
      FACT_LAUNCH((adam_fused2_kernel<64, false>), dim3(nblocks), dim3(256), 0, s, blocks, p, m, v, g, g16, lr_t,
                         b1, b2, eps, gscale, keep);
  }
  return 0;
}
void adam_set_variant(int v) { g_adam_variant = v; }
int adam_tile_width() {
  static bool once = false;
  if (!once) {
    const char* e = getenv("FACT_ADAM_TW");
    if (e && atoi(e) == 128) g_adam_tw = 128;
    if (e && atoi(e) == 64) g_adam_tw = 64;
    once = true;
  }
  return g_adam_tw;
}

int launch_cast_transpose(const float* src, int R, int C, bf16_t* dst, int ldd, bf16_t* dstT, int ldt,
                          hipStream_t s) {
  dim3 grid((C + 63) / 64, (R + 63) / 64);
  FACT_LAUNCH((cast_transpose_kernel<float>), grid, dim3(256), 0, s, src, C, R, C, dst, ldd,
                     dstT, ldt);
  return 0;
}

int launch_multi_cast_transpose(const CastDesc* descs, int n, int total_tiles, hipStream_t s) {
  if (n <= 0 || total_tiles <= 0) return 0;
  FACT_LAUNCH(multi_cast_transpose_kernel, dim3(total_tiles), dim3(256), 0, s, descs, n);
  return 0;
}

int launch_transpose_bf16(const bf16_t* src, int ld, int R, int C, bf16_t* dstT, int ldt,
                          hipStream_t s) {
  dim3 grid((C + 63) / 64, (R + 63) / 64);
  FACT_LAUNCH((cast_transpose_kernel<bf16_t>), grid, dim3(256), 0, s, src, ld, R, C,
                     (bf16_t*)nullptr, 0, dstT, ldt);
  return 0;
}

int launch_pad_cast(const float* src, int n, size_t batch_stride, int M, int F, bf16_t* dst, int Fp,
                    hipStream_t s) {
  const size_t tot = (size_t)M * Fp;
  FACT_LAUNCH(pad_cast_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, n,
                     batch_stride, M, F, dst, Fp);
  return 0;
}

int launch_cast_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t s) {
  if (n & 3) return -1;
  FACT_LAUNCH(cast_bf16_kernel, dim3(grid_for(n >> 2, 256)), dim3(256), 0, s, (const float4*)src,
                     (bf16x4*)dst, n >> 2);
  return 0;
}

int launch_cast_f32(const bf16_t* src, float* dst, size_t n, hipStream_t s) {
  if (n & 3) return -1;
  FACT_LAUNCH(cast_f32_kernel, dim3(grid_for(n >> 2, 256)), dim3(256), 0, s, (const bf16x4*)src,
                     (float4*)dst, n >> 2);
  return 0;
}

int launch_concat_seq(const float* a, const float* b, int B, int na, int nb, int C, float* out, hipStream_t s) {
  if (C & 3) return -1;
  const size_t total = (size_t)B * (na + nb) * (C >> 2);
  FACT_LAUNCH(concat_seq_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, (const float4*)a,
                     (const float4*)b, B, na, nb, C >> 2, (float4*)out);
  return 0;
}

int launch_gather_rows(const float* x, const bf16_t* a, int B, int n, int T, int C, int ld16, float* xc, bf16_t* ac,
                       hipStream_t s) {
  if ((C & 3) || (ld16 & 3) || ld16 < C || T > n) return -1;
  const size_t total = (size_t)B * T * (C >> 2);
  FACT_LAUNCH(gather_rows_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, (const float4*)x,
                     (const bf16x4*)a, B, n, T, C >> 2, ld16 >> 2, (float4*)xc, (bf16x4*)ac);
  return 0;
}

int launch_scatter_rows_zero(const float* dxc, int B, int n, int T, int C, float* dx, hipStream_t s) {
  if ((C & 3) || T > n) return -1;
  const size_t total = (size_t)B * n * (C >> 2);
  FACT_LAUNCH(scatter_rows_zero_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, (const float4*)dxc, B, n, T,
                     C >> 2, (float4*)dx);
  return 0;
}

int launch_split_grad(const float* dx, int B, int na, int nb, int C, float* da, bf16_t* da16, float* db,
                      bf16_t* db16, int ld16, hipStream_t s) {
  if ((C & 3) || ld16 < C || (ld16 & 3)) return -1;
  const size_t total = (size_t)B * (na + nb) * (C >> 2);
  FACT_LAUNCH(split_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, (const float4*)dx,
                     B, na, nb, C >> 2, ld16 >> 2, (float4*)da, (bf16x4*)da16, (float4*)db, (bf16x4*)db16);
  return 0;
}

int launch_slab_reduce(const float* slabs, size_t stride, int nslab, float* out, size_t n, hipStream_t s) {
  if ((stride & 3) || ((uintptr_t)out & 15) || ((uintptr_t)slabs & 15)) return -1;
  const size_t n4 = n >> 2;
  if (n4) FACT_LAUNCH(slab_reduce_kernel, dim3(grid_for(n4, 256, 2048)), dim3(256), 0, s, slabs, stride, nslab, out, n4);
  if (n & 3) FACT_LAUNCH(slab_reduce_tail_kernel, dim3(1), dim3(64), 0, s, slabs, stride, nslab, out, n4 << 2, n);
  return 0;
}

int launch_sumsq(const float* g, size_t n, float* out, hipStream_t s) {
  if (n & 3) return -1;
  FACT_LAUNCH(sumsq_kernel, dim3(grid_for(n >> 2, 256, 2048)), dim3(256), 0, s,
                     (const float4*)g, n >> 2, out);
  return 0;
}

int launch_clip_scale(float* g, size_t n, const float* sumsq, float clip, float* norm_out, hipStream_t s) {
  if (n & 3) return -1;
  FACT_LAUNCH(clip_scale_kernel, dim3(grid_for(n >> 2, 256, 2048)), dim3(256), 0, s, (float4*)g, n >> 2, sumsq, clip, norm_out);
  return 0;
}
