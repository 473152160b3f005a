// Fused attention forward / backward kernels (see attention.h for layouts and conventions).
//
// MFMA slot convention for every "contraction over 32 tokens" product (P^T, dS^T as B operand;
// V^T, K^T, Q^T, dO^T tiles as A operand): for lane group g = lane>>4 the 8 k-slots are
//   slot j<4  -> token g*4 + j          (first 16-token sub-tile)
//   slot j>=4 -> token 16 + g*4 + (j-4) (second 16-token sub-tile)
// which is exactly what the C layout of two stacked 16x16 score tiles leaves in each lane, so the
// probabilities never move between lanes.
#include "attention.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <int DH>
struct Geo {
  static constexpr int ND = DH / 16;         // 16-wide head-dim tiles
  static constexpr int KD = (DH + 31) / 32;  // 32-deep contraction steps over the head dim
  static constexpr int DHP = KD * 32;
  static constexpr int KS = DHP + 8;  // LDS row stride (elements) of a [32-token][DHP] tile
};

DEVINL bf16x8 cat4(bf16x4 a, bf16x4 b) {
  bf16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return r;
}
DEVINL bf16x8 pack8(f32x4 a, f32x4 b) {
  bf16x8 r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3],
              (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
  return r;
}

// stage a [32 tokens][DHP] tile of a token-major buffer into LDS (row stride KS)
template <int DH>
DEVINL void stage_rows(bf16_t* dst, const bf16_t* src_tile /* &buf[bh][tok0][0] */, int tid) {
  constexpr int CH = Geo<DH>::DHP / 8;
  for (int c = tid; c < 32 * CH; c += 256) {
    const int row = c / CH, ch = c - row * CH;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(src_tile + (size_t)row * Geo<DH>::DHP + ch * 8);
    *reinterpret_cast<bf16x8*>(dst + row * Geo<DH>::KS + ch * 8) = v;
  }
}
// A-operand fragment [16 head-dims of tile dt][32 tokens] built from a token-major LDS tile with the
// LDS transpose read: lane group g reads the 4x16 block {tokens hh*16 + g*4 .. +3} x {dims dt*16 .. +15};
// lane s supplies the address of 4 contiguous dims of token hh*16 + g*4 + (s>>2); the hardware hands
// lane c the 4 token-values of dim dt*16 + c  ->  slots j<4: token g*4+j, j>=4: token 16+g*4+(j-4).
template <int DH>
DEVINL bf16x8 frag_tr(const bf16_t* t, int dt, int lane) {
  const int g = lane >> 4, s = lane & 15;
  const bf16_t* p = t + (g * 4 + (s >> 2)) * Geo<DH>::KS + dt * 16 + (s & 3) * 4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * Geo<DH>::KS));
  return cat4(lo, hi);
}
// A-operand fragment of a row tile: 16 tokens (sub), contraction step kd
template <int DH>
DEVINL bf16x8 frag_row(const bf16_t* t, int sub, int kd, int lane) {
  return *reinterpret_cast<const bf16x8*>(t + (sub * 16 + (lane & 15)) * Geo<DH>::KS + kd * 32 +
                                          (lane >> 4) * 8);
}

// ---------------------------------------------------------------------------------------------
// forward: block = (128-query tile, b*h), 4 waves x 32 queries, loop over 32-key tiles
// ---------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
  using G = Geo<DH>;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[32 * G::KS];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[32 * G::KS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int bh = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const size_t row_base = (size_t)bh * p.NP * G::DHP;

  bf16x8 Qf[2][G::KD];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs)
#pragma unroll
    for (int kd = 0; kd < G::KD; ++kd)
      Qf[qs][kd] = *reinterpret_cast<const bf16x8*>(
          p.qrow + row_base + (size_t)(q0 + qs * 16 + (lane & 15)) * G::DHP + kd * 32 + g * 8);

  f32x4 O[2][G::ND];
  float m[2], l[2];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    m[qs] = -INFINITY;
    l[qs] = 0.f;
#pragma unroll
    for (int dt = 0; dt < G::ND; ++dt) O[qs][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float sc = p.scale * LOG2E;
  const int nkt = (p.n + 31) / 32;
  for (int kt = 0; kt < nkt; ++kt) {
    stage_rows<DH>(Ks, p.krow + row_base + (size_t)kt * 32 * G::DHP, tid);
    stage_rows<DH>(Vs, p.vrow + row_base + (size_t)kt * 32 * G::DHP, tid);
    __syncthreads();
    f32x4 s[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) s[ks][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kd = 0; kd < G::KD; ++kd) {
        const bf16x8 kf = frag_row<DH>(Ks, ks, kd, lane);
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) s[ks][qs] = mfma16(kf, Qf[qs][kd], s[ks][qs]);
      }
    // scale + mask padded keys
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool valid = (kt * 32 + ks * 16 + g * 4 + r) < p.n;
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) s[ks][qs][r] = valid ? s[ks][qs][r] * sc : -INFINITY;
      }
    bf16x8 pb[2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      float mx = fmaxf(fmaxf(fmaxf(s[0][qs][0], s[0][qs][1]), fmaxf(s[0][qs][2], s[0][qs][3])),
                       fmaxf(fmaxf(s[1][qs][0], s[1][qs][1]), fmaxf(s[1][qs][2], s[1][qs][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m[qs], mx);
      const float alpha = __builtin_amdgcn_exp2f(m[qs] - mn);
      f32x4 p0, p1;
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p0[r] = __builtin_amdgcn_exp2f(s[0][qs][r] - mn);
        p1[r] = __builtin_amdgcn_exp2f(s[1][qs][r] - mn);
        ls += p0[r] + p1[r];
      }
      l[qs] = l[qs] * alpha + ls;
      m[qs] = mn;
#pragma unroll
      for (int dt = 0; dt < G::ND; ++dt) {
        O[qs][dt][0] *= alpha; O[qs][dt][1] *= alpha; O[qs][dt][2] *= alpha; O[qs][dt][3] *= alpha;
      }
      pb[qs] = pack8(p0, p1);
    }
#pragma unroll
    for (int dt = 0; dt < G::ND; ++dt) {
      const bf16x8 vf = frag_tr<DH>(Vs, dt, lane);
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) O[qs][dt] = mfma16(vf, pb[qs], O[qs][dt]);
    }
    __syncthreads();
  }
  const int b = bh / p.H, h = bh - b * p.H;
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    float lt = l[qs];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv = 1.0f / lt;
    const int t = q0 + qs * 16 + (lane & 15);
    if (g == 0 && t < p.NP) p.lse2[(size_t)bh * p.NP + t] = m[qs] + __log2f(lt);
    if (t < p.n) {
      bf16_t* orow = p.out + ((size_t)b * p.n + t) * p.ldo + h * DH + g * 4;
#pragma unroll
      for (int dt = 0; dt < G::ND; ++dt) {
        bf16x4 o = {(bf16_t)(O[qs][dt][0] * inv), (bf16_t)(O[qs][dt][1] * inv),
                    (bf16_t)(O[qs][dt][2] * inv), (bf16_t)(O[qs][dt][3] * inv)};
        *reinterpret_cast<bf16x4*>(orow + dt * 16) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward prep: D[bh][t] = sum_d dO[t][d] * O[t][d]
// ---------------------------------------------------------------------------------------------
template <int DH>
__global__ void attn_bwd_prep_kernel(const AttnParams p) {
  using G = Geo<DH>;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = p.B * p.H * p.NP;
  if (idx >= total) return;
  const int bh = idx / p.NP, t = idx - bh * p.NP;
  float s = 0.f;
  if (t < p.n) {
    const int b = bh / p.H, h = bh - b * p.H;
    const bf16_t* d = p.dorow + ((size_t)bh * p.NP + t) * G::DHP;
    const bf16_t* o = p.o + ((size_t)b * p.n + t) * p.ldo + h * DH;
#pragma unroll
    for (int c = 0; c < DH; c += 8) {
      const bf16x8 dv = *reinterpret_cast<const bf16x8*>(d + c);
      const bf16x8 ov = *reinterpret_cast<const bf16x8*>(o + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)dv[j] * (float)ov[j];
    }
  }
  p.dsum[idx] = s;
}

// ---------------------------------------------------------------------------------------------
// backward dQ: block = (128-query tile, b*h), loop over 32-key tiles
//   P^T = exp2(S^T*c - L2[q]);  dP^T = V dO^T;  dS^T = P^T o (dP^T - D[q]);  dQ^T += K^T dS^T
// ---------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnParams p) {
  using G = Geo<DH>;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[32 * G::KS];
  __shared__ __attribute__((aligned(16))) bf16_t Vr[32 * G::KS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int bh = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const size_t row_base = (size_t)bh * p.NP * G::DHP;

  bf16x8 Qf[2][G::KD], dOf[2][G::KD];
  float L2q[2], Dq[2];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    const int q = q0 + qs * 16 + (lane & 15);
#pragma unroll
    for (int kd = 0; kd < G::KD; ++kd) {
      const size_t off = row_base + (size_t)q * G::DHP + kd * 32 + g * 8;
      Qf[qs][kd] = *reinterpret_cast<const bf16x8*>(p.qrow + off);
      dOf[qs][kd] = *reinterpret_cast<const bf16x8*>(p.dorow + off);
    }
    L2q[qs] = p.lse2[(size_t)bh * p.NP + q];
    Dq[qs] = p.dsum[(size_t)bh * p.NP + q];
  }
  f32x4 dQ[2][G::ND];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs)
#pragma unroll
    for (int dt = 0; dt < G::ND; ++dt) dQ[qs][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float sc = p.scale * LOG2E;
  const int nkt = (p.n + 31) / 32;
  for (int kt = 0; kt < nkt; ++kt) {
    stage_rows<DH>(Ks, p.krow + row_base + (size_t)kt * 32 * G::DHP, tid);
    stage_rows<DH>(Vr, p.vrow + row_base + (size_t)kt * 32 * G::DHP, tid);
    __syncthreads();
    f32x4 s[2][2], dp[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        s[ks][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[ks][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kd = 0; kd < G::KD; ++kd) {
        const bf16x8 kf = frag_row<DH>(Ks, ks, kd, lane);
        const bf16x8 vf = frag_row<DH>(Vr, ks, kd, lane);
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
          s[ks][qs] = mfma16(kf, Qf[qs][kd], s[ks][qs]);
          dp[ks][qs] = mfma16(vf, dOf[qs][kd], dp[ks][qs]);
        }
      }
    bf16x8 dsb[2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      f32x4 d0, d1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool v0 = (kt * 32 + g * 4 + r) < p.n;
        const bool v1 = (kt * 32 + 16 + g * 4 + r) < p.n;
        const float p0 = v0 ? __builtin_amdgcn_exp2f(s[0][qs][r] * sc - L2q[qs]) : 0.f;
        const float p1 = v1 ? __builtin_amdgcn_exp2f(s[1][qs][r] * sc - L2q[qs]) : 0.f;
        d0[r] = p0 * (dp[0][qs][r] - Dq[qs]);
        d1[r] = p1 * (dp[1][qs][r] - Dq[qs]);
      }
      dsb[qs] = pack8(d0, d1);
    }
#pragma unroll
    for (int dt = 0; dt < G::ND; ++dt) {
      const bf16x8 ktf = frag_tr<DH>(Ks, dt, lane);
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) dQ[qs][dt] = mfma16(ktf, dsb[qs], dQ[qs][dt]);
    }
    __syncthreads();
  }
  const int b = bh / p.H, h = bh - b * p.H;
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    const int t = q0 + qs * 16 + (lane & 15);
    if (t < p.n) {
      bf16_t* orow = p.dqkv + ((size_t)b * p.n + t) * p.ldq + h * DH + g * 4;
#pragma unroll
      for (int dt = 0; dt < G::ND; ++dt) {
        bf16x4 o = {(bf16_t)(dQ[qs][dt][0] * p.scale), (bf16_t)(dQ[qs][dt][1] * p.scale),
                    (bf16_t)(dQ[qs][dt][2] * p.scale), (bf16_t)(dQ[qs][dt][3] * p.scale)};
        *reinterpret_cast<bf16x4*>(orow + dt * 16) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward dK/dV: block = (128-key tile, b*h), 4 waves x 32 keys, loop over 32-query tiles
//   S = Q K^T; P = exp2(S*c - L2[q]); dV^T += dO^T P; dP = dO V^T; dS = P o (dP - D[q]);
//   dK^T += Q^T dS
// ---------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(const AttnParams p) {
  using G = Geo<DH>;
  __shared__ __attribute__((aligned(16))) bf16_t Qs[32 * G::KS];
  __shared__ __attribute__((aligned(16))) bf16_t dOs[32 * G::KS];
  __shared__ __attribute__((aligned(16))) float L2s[32];
  __shared__ __attribute__((aligned(16))) float Dss[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int bh = blockIdx.y;
  const int key0 = blockIdx.x * 128 + wave * 32;
  const size_t row_base = (size_t)bh * p.NP * G::DHP;

  bf16x8 Kf[2][G::KD], Vf[2][G::KD];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int kd = 0; kd < G::KD; ++kd) {
      const size_t off = row_base + (size_t)(key0 + ks * 16 + (lane & 15)) * G::DHP + kd * 32 + g * 8;
      Kf[ks][kd] = *reinterpret_cast<const bf16x8*>(p.krow + off);
      Vf[ks][kd] = *reinterpret_cast<const bf16x8*>(p.vrow + off);
    }
  f32x4 dK[2][G::ND], dV[2][G::ND];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int dt = 0; dt < G::ND; ++dt) {
      dK[ks][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dV[ks][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  bool kvalid[2];
  kvalid[0] = (key0 + (lane & 15)) < p.n;
  kvalid[1] = (key0 + 16 + (lane & 15)) < p.n;

  const float sc = p.scale * LOG2E;
  const int nqt = (p.n + 31) / 32;
  for (int qt = 0; qt < nqt; ++qt) {
    stage_rows<DH>(Qs, p.qrow + row_base + (size_t)qt * 32 * G::DHP, tid);
    stage_rows<DH>(dOs, p.dorow + row_base + (size_t)qt * 32 * G::DHP, tid);
    if (tid < 32) L2s[tid] = p.lse2[(size_t)bh * p.NP + qt * 32 + tid];
    else if (tid < 64) Dss[tid - 32] = p.dsum[(size_t)bh * p.NP + qt * 32 + tid - 32];
    __syncthreads();
    f32x4 s[2][2], dp[2][2];  // [qsub][ksub]
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        s[qs][ks] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[qs][ks] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
      for (int kd = 0; kd < G::KD; ++kd) {
        const bf16x8 qf = frag_row<DH>(Qs, qs, kd, lane);
        const bf16x8 df = frag_row<DH>(dOs, qs, kd, lane);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s[qs][ks] = mfma16(qf, Kf[ks][kd], s[qs][ks]);
          dp[qs][ks] = mfma16(df, Vf[ks][kd], dp[qs][ks]);
        }
      }
    const f32x4 l2a = *reinterpret_cast<const f32x4*>(&L2s[g * 4]);
    const f32x4 l2b = *reinterpret_cast<const f32x4*>(&L2s[16 + g * 4]);
    const f32x4 dda = *reinterpret_cast<const f32x4*>(&Dss[g * 4]);
    const f32x4 ddb = *reinterpret_cast<const f32x4*>(&Dss[16 + g * 4]);
    bf16x8 pb[2], dsb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 p0, p1, d0, d1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // padded keys AND padded query rows are masked: the dO scratch is shared by stacks of different
        // geometry, so its padding rows are not guaranteed to be zero
        const bool qv0 = (qt * 32 + g * 4 + r) < p.n, qv1 = (qt * 32 + 16 + g * 4 + r) < p.n;
        p0[r] = (kvalid[ks] && qv0) ? __builtin_amdgcn_exp2f(s[0][ks][r] * sc - l2a[r]) : 0.f;
        p1[r] = (kvalid[ks] && qv1) ? __builtin_amdgcn_exp2f(s[1][ks][r] * sc - l2b[r]) : 0.f;
        d0[r] = p0[r] * (dp[0][ks][r] - dda[r]);
        d1[r] = p1[r] * (dp[1][ks][r] - ddb[r]);
      }
      pb[ks] = pack8(p0, p1);
      dsb[ks] = pack8(d0, d1);
    }
#pragma unroll
    for (int dt = 0; dt < G::ND; ++dt) {
      const bf16x8 dof = frag_tr<DH>(dOs, dt, lane);
      const bf16x8 qtf = frag_tr<DH>(Qs, dt, lane);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        dV[ks][dt] = mfma16(dof, pb[ks], dV[ks][dt]);
        dK[ks][dt] = mfma16(qtf, dsb[ks], dK[ks][dt]);
      }
    }
    __syncthreads();
  }
  const int b = bh / p.H, h = bh - b * p.H;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int t = key0 + ks * 16 + (lane & 15);
    if (t < p.n) {
      bf16_t* krow_o = p.dqkv + ((size_t)b * p.n + t) * p.ldq + p.hid + h * DH + g * 4;
      bf16_t* vrow_o = krow_o + p.hid;
#pragma unroll
      for (int dt = 0; dt < G::ND; ++dt) {
        bf16x4 ok = {(bf16_t)(dK[ks][dt][0] * p.scale), (bf16_t)(dK[ks][dt][1] * p.scale),
                     (bf16_t)(dK[ks][dt][2] * p.scale), (bf16_t)(dK[ks][dt][3] * p.scale)};
        bf16x4 ov = {(bf16_t)dV[ks][dt][0], (bf16_t)dV[ks][dt][1], (bf16_t)dV[ks][dt][2],
                     (bf16_t)dV[ks][dt][3]};
        *reinterpret_cast<bf16x4*>(krow_o + dt * 16) = ok;
        *reinterpret_cast<bf16x4*>(vrow_o + dt * 16) = ov;
      }
    }
  }
}

// =============================================================================================
// LDS-resident family (the fast path for FACT's sequence lengths, n <= 512)
// ---------------------------------------------------------------------------------------------
// One workgroup per (batch, head) with one wave per 32 query (or key) rows.  The operands every
// wave needs for ALL tiles - K and V (forward, dQ) or Q and dO (dK/dV) of this head - are brought
// into LDS ONCE with 16-byte LDS-DMA (global_load_lds_dwordx4) and stay resident: 360x96 bf16 is
// 69 KiB per operand, two fit in the 160 KiB LDS.  After one vmcnt(0)+barrier the waves run their
// whole tile loop barrier-free (MFMA / exp / LDS-read streams of the 3 waves per SIMD interleave).
//
// LDS images (no padding, DMA writes them linearly, the swizzle is applied to the SOURCE address):
//   img96: [rows][DHP] bf16, 16-byte chunk c of row r at position c ^ G[(r>>2)&3], G = {0,2,3,1}
//          -> ds_read_b128 row fragments are bank-conflict free; token-contracting fragments come
//          from the same image through ds_read_b64_tr_b16.
//   img80: [rows][DH] bf16 un-swizzled (forward V: only transpose reads; 160-byte rows already
//          spread 8 consecutive rows over all 64 banks).
// =============================================================================================
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

DEVINL int res_g(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

// DMA `rows` rows of a token-major [.][DHP] buffer into an LDS image with CPR 16-byte chunks per
// row (CPR*8 <= DHP columns are kept); SWZ selects the img96 swizzle.
template <int DHP, int CPR, bool SWZ>
DEVINL void res_load(unsigned char* img, const bf16_t* src, int rows, int wave, int nwaves, int lane) {
  const int total = rows * CPR;
  const int pieces = (total + 63) >> 6;
  for (int pi = wave; pi < pieces; pi += nwaves) {
    const int q = pi * 64 + lane;
    if (q < total) {
      const int r = q / CPR, cp = q - r * CPR;
      const int c = SWZ ? (cp ^ res_g(r)) : cp;
      __builtin_amdgcn_global_load_lds((gbl_cvoid*)(src + (size_t)r * DHP + c * 8), (lds_void*)(img + pi * 1024), 16,
                                       0, 0);
    }
  }
}

// row fragment (A or B operand, contraction over the head dim) from an img96
template <int DHP>
DEVINL bf16x8 res_frag_row(const unsigned char* img, int row, int kd, int lane) {
  const int r = row + (lane & 15);
  const int c = kd * 4 + (lane >> 4);
  return *reinterpret_cast<const bf16x8*>(img + r * (DHP * 2) + ((c ^ res_g(r)) << 4));
}

// (Round 6: the round-2/3 LDS-resident kernels - attn_fwd_res_kernel, attn_bwd_dq_res_kernel, attn_bwd_dkdv_res_kernel - were
// removed from the library; the lean resident backward below is their successor, the streaming forward the default forward, the
// tiled kernels the forced reference path.  git history: round 5.)

// =============================================================================================
// Streaming family (round 2).  The LDS-resident kernels above put one 8..12-wave workgroup per (batch, head)
// on a CU: 160 workgroups for 256 CUs, and each of them first waits ~12k cycles for its 138 KiB of K/V (or
// Q/dO) before any MFMA issues (PMC round 2: 9-14 % MFMA busy).  Here a workgroup is 4 waves x 32 own rows
// (queries for forward / dQ, keys for dK/dV) = a 128-row block of one head, grid = (ceil(n/128), B*H) = 480
// workgroups at FACT's cross-modal shape, and the OTHER side of the product is streamed through a 4-slot LDS
// ring in 32-row stages (two swizzled [32][DHP] images = the img96 layout above, 12 KiB) by 16-byte LDS-DMA with
// counted vmcnt: 48 KiB per workgroup, so 2-3 workgroups share a CU and one's DMA / softmax / epilogue runs
// under another's MFMAs.  One raw s_barrier per stage:
//   top of step t : wait for this wave's pieces of stage t (two later stages stay in flight), barrier,
//                   issue stage t+3 into the slot stage t-1 used (everyone's reads of it precede the barrier)
// Fragment conventions are those of the resident kernels.  Transpose reads are inline asm (hipcc puts a full
// vmcnt(0) in front of its ds_read_tr builtin while an LDS-DMA is in flight), retired by a hand lgkmcnt(0).
// =============================================================================================
template <int DH>
struct SG {
  static constexpr int DHP = Geo<DH>::DHP, ROWB = DHP * 2, CPR = DHP / 8;
  static constexpr int IMG = 32 * ROWB;    // one [32][DHP] image
  static constexpr int PPI = IMG / 1024;   // DMA pieces per image
  static constexpr int PPW = 2 * PPI / 4;  // pieces per wave per stage (2 images, 4 waves)
  static constexpr int NSLOT = 4;
};
template <int S>
struct SlotK { static constexpr int value = S; };

template <int N>
DEVINL void st_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
DEVINL void st_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int OFF>
DEVINL bf16x4 st_tr(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  bf16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
DEVINL unsigned st_lds_addr(const unsigned char* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

// per-wave DMA plan of a stage: this wave's PPW pieces of the two images
template <int DH>
struct StDma {
  const bf16_t* src[SG<DH>::PPW];  // per-lane source of tile 0 (swizzle applied to the source chunk)
  int dst[SG<DH>::PPW];            // byte offset of the piece inside a stage
};
template <int DH>
DEVINL void st_dma_init(StDma<DH>& d, const bf16_t* img0, const bf16_t* img1, int wave, int lane) {
  using S = SG<DH>;
#pragma unroll
  for (int i = 0; i < S::PPW; ++i) {
    const int P = wave * S::PPW + i, im = P / S::PPI, pi = P - im * S::PPI;
    const int q = pi * 64 + lane, r = q / S::CPR, cp = q - r * S::CPR;
    const int c = cp ^ res_g(r);
    d.src[i] = (im ? img1 : img0) + (size_t)r * S::DHP + c * 8;
    d.dst[i] = im * S::IMG + pi * 1024;
  }
}
template <int DH>
DEVINL void st_dma_issue(const StDma<DH>& d, unsigned char* stage, int tile) {
  using S = SG<DH>;
#pragma unroll
  for (int i = 0; i < S::PPW; ++i)
    __builtin_amdgcn_global_load_lds((gbl_cvoid*)(d.src[i] + (size_t)tile * 32 * S::DHP), (lds_void*)(stage + d.dst[i]),
                                     16, 0, 0);
}

// row fragment (contraction over the head dim) of tile-local rows sub*16.. of an image
template <int DH>
DEVINL bf16x8 st_frag_row(const unsigned char* img, int sub, int kd, int lane) {
  return res_frag_row<SG<DH>::DHP>(img, sub * 16, kd, lane);
}
// per-lane LDS byte addresses (image base 0) for the token-contracting fragments: even / odd 16-dim tiles
struct StTrAddr { unsigned e, o; };
template <int DH>
DEVINL StTrAddr st_tr_addr(unsigned lds0, int lane) {
  const int g = lane >> 4, s = lane & 15;
  const int r = g * 4 + (s >> 2), low = ((s & 3) >> 1) ^ res_g(r);
  StTrAddr a;
  a.e = lds0 + (unsigned)(r * SG<DH>::ROWB + low * 16 + (s & 1) * 8);
  a.o = lds0 + (unsigned)(r * SG<DH>::ROWB + (low ^ 2) * 16 + (s & 1) * 8);
  return a;
}
// all ND token-contracting fragments [16 dims of tile dt][32 tokens] of the image at byte offset OFF (compile time)
template <int DH, int OFF, int DT>
DEVINL void st_tr_all(const StTrAddr& a, bf16x4 (&lo)[Geo<DH>::ND], bf16x4 (&hi)[Geo<DH>::ND]) {
  if constexpr (DT < Geo<DH>::ND) {
    constexpr int O = OFF + (DT >> 1) * 64;
    lo[DT] = st_tr<O>((DT & 1) ? a.o : a.e);
    hi[DT] = st_tr<O + 16 * SG<DH>::ROWB>((DT & 1) ? a.o : a.e);
    st_tr_all<DH, OFF, DT + 1>(a, lo, hi);
  }
}
// the asm reads are invisible to hipcc's counters: retire them and tie the fragments to the wait
template <int N>
DEVINL void st_tr_retire(bf16x4 (&lo)[N], bf16x4 (&hi)[N]) {
  st_wait_lgkm0();
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(lo[i]), "+v"(hi[i]));
}

DEVINL float st_max_f32(float a, float b) {  // single v_max_f32 (fmaxf adds canonicalising v_max on MFMA outputs)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// max over the 4 lanes {l, l^16, l^32, l^48} with the gfx950 row / half swaps (no LDS round trip)
DEVINL float st_allmax4(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  const float a = st_max_f32(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
  const unsigned ua = __builtin_bit_cast(unsigned, a);
  const auto q = __builtin_amdgcn_permlane32_swap(ua, ua, false, false);
  const unsigned q0 = q[0], q1 = q[1];
  return st_max_f32(__builtin_bit_cast(float, q0), __builtin_bit_cast(float, q1));
}
DEVINL float st_allsum4(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  const float a = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
  const unsigned ua = __builtin_bit_cast(unsigned, a);
  const auto q = __builtin_amdgcn_permlane32_swap(ua, ua, false, false);
  const unsigned q0 = q[0], q1 = q[1];
  return __builtin_bit_cast(float, q0) + __builtin_bit_cast(float, q1);
}

constexpr float ST_THR = 6.0f;  // lazy running max: P = exp2(t) stays below 2^6 between exact updates

// ---- forward -------------------------------------------------------------------------------------
// Lean softmax: the running row maximum m is only raised when a score exceeds it by more than ST_THR (log2
// units; wave-uniform test, no cross-lane traffic on the common path) - softmax is shift invariant, so any m
// that keeps exp2 in range is exact up to rounding - and the row sum is not a VALU reduction: a constant
// "ones" A-fragment adds one 16-row tile to O^T whose row 0 is sum_k P[k][q], accumulated (and rescaled) by the
// MFMA pipe from the same bf16 P that multiplies V.
template <int DH>
__global__ __launch_bounds__(256, DH > 96 ? 2 : 3) void attn_fwd_st_kernel(const AttnParams p) {
  using G = Geo<DH>;
  using S = SG<DH>;
  constexpr int IMG = S::IMG, STAGE = 2 * IMG, PPW = S::PPW, ND = G::ND;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid of (query blocks per head) x (heads), re-dealt so that the query blocks of one head are NEIGHBOURS ON ONE
  // XCD (block b runs on XCD b % 8, observed): the 2-3 workgroups that stream the same K / V tiles then hit one private L2
  // instead of pulling them through the fabric once per XCD (round 4: the kernel moves 69 MB of K / V by LDS-DMA for
  // 23.6 MB of operands - 5.3 TB/s over its loop, the fabric's rate).  Speed only; any placement is correct.
  int bh, qblk;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, per = nwg >> 3, rem = nwg & 7;
    const int lid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (orig >> 3);
    bh = lid / p.qblocks;
    qblk = lid - bh * p.qblocks;
  }
  const int T = (p.n + 31) >> 5;
  const size_t row_base = (size_t)bh * p.NP * G::DHP;
  const int q0 = qblk * 128 + wave * 32;
  // wave-uniform; inactive waves still stage and synchronise.  Supervised-rows shortcut (nq): only the first nq queries of
  // a sequence are wanted - the launcher sizes the grid for them, waves past them only help with the staging
  const bool active = q0 < ((p.nq > 0 && p.nq < p.n) ? p.nq : p.n);

  bf16x8 Qf[2][G::KD];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs)
#pragma unroll
    for (int kd = 0; kd < G::KD; ++kd)
      Qf[qs][kd] = *reinterpret_cast<const bf16x8*>(
          p.qrow + row_base + (size_t)(q0 + qs * 16 + (lane & 15)) * G::DHP + kd * 32 + g * 8);
  StDma<DH> dma;
  st_dma_init<DH>(dma, p.krow + row_base, p.vrow + row_base, wave, lane);
  st_dma_issue<DH>(dma, smem, 0);
  st_dma_issue<DH>(dma, smem + STAGE, min(1, T - 1));
  st_dma_issue<DH>(dma, smem + 2 * STAGE, min(2, T - 1));

  // the compiler must see the Q loads retired BEFORE the loop (otherwise it drains vmcnt to 0 in every step)
#pragma unroll
  for (int qs = 0; qs < 2; ++qs)
#pragma unroll
    for (int kd = 0; kd < G::KD; ++kd) asm volatile("" : "+v"(Qf[qs][kd]));
  const StTrAddr va = st_tr_addr<DH>(st_lds_addr(smem), lane);
  f32x4 O[2][ND + 1];  // [ND]: the ones tile (row sums)
  float m[2] = {0.f, 0.f};
#pragma unroll
  for (int qs = 0; qs < 2; ++qs)
#pragma unroll
    for (int dt = 0; dt <= ND; ++dt) O[qs][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones = zero_bf16x8();
  if ((lane & 15) == 0) {
    const bf16_t one = (bf16_t)1.0f;
    ones = bf16x8{one, one, one, one, one, one, one, one};
  }
  const float sc = p.scale * LOG2E;

  auto body = [&](auto slot_c, int kt) {
    constexpr int SLOT = decltype(slot_c)::value;
    st_wait_vm<2 * PPW>();
    __builtin_amdgcn_s_barrier();
    st_dma_issue<DH>(dma, smem + ((SLOT + 3) & 3) * STAGE, min(kt + 3, T - 1));
    if (!active) return;
    const unsigned char* Kimg = smem + SLOT * STAGE;
    f32x4 s[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) s[ks][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kd = 0; kd < G::KD; ++kd) {
        const bf16x8 kf = st_frag_row<DH>(Kimg, ks, kd, lane);
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) s[ks][qs] = mfma16(kf, Qf[qs][kd], s[ks][qs]);
      }
    // V^T fragments of this stage: issued now, consumed after the softmax
    bf16x4 vlo[ND], vhi[ND];
    st_tr_all<DH, SLOT * STAGE + IMG, 0>(va, vlo, vhi);
    if (kt == T - 1 && (p.n & 31)) {  // padded keys only exist in the last tile (wave-uniform)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool valid = (kt * 32 + ks * 16 + g * 4 + r) < p.n;
#pragma unroll
          for (int qs = 0; qs < 2; ++qs) s[ks][qs][r] = valid ? s[ks][qs][r] : -INFINITY;
        }
    }
    float t[2][8], mx[2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        t[qs][r] = fmaf(s[0][qs][r], sc, -m[qs]);
        t[qs][4 + r] = fmaf(s[1][qs][r], sc, -m[qs]);
      }
      mx[qs] = st_max_f32(st_max_f32(st_max_f32(t[qs][0], t[qs][1]), st_max_f32(t[qs][2], t[qs][3])),
                          st_max_f32(st_max_f32(t[qs][4], t[qs][5]), st_max_f32(t[qs][6], t[qs][7])));
    }
    if (kt == 0 || __any(st_max_f32(mx[0], mx[1]) > ST_THR)) {
      // exact update of the running maximum (always on the first tile, then only when a score outgrew it)
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        const float tm = st_allmax4(mx[qs]);                  // row maximum of this tile relative to m
        const float d = (kt == 0) ? tm : fmaxf(tm, 0.f);      // raise only
        m[qs] += d;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[qs][j] -= d;
        if (kt != 0) {
          const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
          for (int dt = 0; dt <= ND; ++dt) {
            O[qs][dt][0] *= alpha; O[qs][dt][1] *= alpha; O[qs][dt][2] *= alpha; O[qs][dt][3] *= alpha;
          }
        }
      }
    }
    bf16x8 pb[2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      f32x4 p0, p1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p0[r] = __builtin_amdgcn_exp2f(t[qs][r]);
        p1[r] = __builtin_amdgcn_exp2f(t[qs][4 + r]);
      }
      pb[qs] = pack8(p0, p1);
    }
    st_tr_retire<ND>(vlo, vhi);
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      const bf16x8 vf = cat4(vlo[dt], vhi[dt]);
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) O[qs][dt] = mfma16(vf, pb[qs], O[qs][dt]);
    }
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) O[qs][ND] = mfma16(ones, pb[qs], O[qs][ND]);
  };
  for (int kt = 0; kt < T; kt += 4) {
    body(SlotK<0>{}, kt);
    if (kt + 1 < T) body(SlotK<1>{}, kt + 1);
    if (kt + 2 < T) body(SlotK<2>{}, kt + 2);
    if (kt + 3 < T) body(SlotK<3>{}, kt + 3);
  }
  st_wait_vm<0>();  // run-ahead stages: landed before the LDS is handed to the next workgroup
  if (!active) return;
  const int b = bh / p.H, h = bh - b * p.H;
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    // row sum: row 0 of the ones tile = element 0 of lane group 0; hand it to the query's other 3 lanes
    const float lt = __shfl(O[qs][ND][0], lane & 15, 64);
    const float inv = 1.0f / lt;
    const int t = q0 + qs * 16 + (lane & 15);
    if (g == 0 && t < p.NP) p.lse2[(size_t)bh * p.NP + t] = m[qs] + __log2f(lt);
    if (t < p.n) {
      bf16_t* orow = p.out + ((size_t)b * p.n + t) * p.ldo + h * DH + g * 4;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        bf16x4 o = {(bf16_t)(O[qs][dt][0] * inv), (bf16_t)(O[qs][dt][1] * inv),
                    (bf16_t)(O[qs][dt][2] * inv), (bf16_t)(O[qs][dt][3] * inv)};
        *reinterpret_cast<bf16x4*>(orow + dt * 16) = o;
      }
    }
  }
}

// ---- dQ ------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256, DH > 96 ? 2 : 3) void attn_bwd_dq_st_kernel(const AttnParams p) {
  using G = Geo<DH>;
  using S = SG<DH>;
  constexpr int IMG = S::IMG, STAGE = 2 * IMG, PPW = S::PPW, ND = G::ND;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.y;
  const int T = (p.n + 31) >> 5;
  const size_t row_base = (size_t)bh * p.NP * G::DHP;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const bool active = q0 < p.n;
  const int b = bh / p.H, h = bh - b * p.H;

  bf16x8 Qf[2][G::KD], dOf[2][G::KD];
  float L2q[2], Dq[2];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    const int q = q0 + qs * 16 + (lane & 15);
    float part = 0.f;  // D[q] = rowsum(dO o O): this lane's 8 head columns per 32-column slab
#pragma unroll
    for (int kd = 0; kd < G::KD; ++kd) {
      const size_t off = row_base + (size_t)q * G::DHP + kd * 32 + g * 8;
      Qf[qs][kd] = *reinterpret_cast<const bf16x8*>(p.qrow + off);
      dOf[qs][kd] = *reinterpret_cast<const bf16x8*>(p.dorow + off);
      const int d0 = kd * 32 + g * 8;
      if (q < p.n && d0 < DH) {
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(p.o + ((size_t)b * p.n + q) * p.ldo + h * DH + d0);
#pragma unroll
        for (int j = 0; j < 8; ++j) part += (float)dOf[qs][kd][j] * (float)ov[j];
      }
    }
    part = st_allsum4(part);
    L2q[qs] = p.lse2[(size_t)bh * p.NP + q];
    Dq[qs] = part;
    if (g == 0) p.dsum[(size_t)bh * p.NP + q] = part;  // for the dK/dV kernel (0 on padded query rows)
  }
  StDma<DH> dma;
  st_dma_init<DH>(dma, p.krow + row_base, p.vrow + row_base, wave, lane);
  st_wait_vm<0>();  // the operand loads above: the counted waits below then only see DMA pieces
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    asm volatile("" : "+v"(L2q[qs]), "+v"(Dq[qs]));
#pragma unroll
    for (int kd = 0; kd < G::KD; ++kd) asm volatile("" : "+v"(Qf[qs][kd]), "+v"(dOf[qs][kd]));
  }
  st_dma_issue<DH>(dma, smem, 0);
  st_dma_issue<DH>(dma, smem + STAGE, min(1, T - 1));
  st_dma_issue<DH>(dma, smem + 2 * STAGE, min(2, T - 1));
  const StTrAddr ka = st_tr_addr<DH>(st_lds_addr(smem), lane);
  f32x4 dQ[2][ND];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs)
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) dQ[qs][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float sc = p.scale * LOG2E;

  auto body = [&](auto slot_c, int kt) {
    constexpr int SLOT = decltype(slot_c)::value;
    st_wait_vm<2 * PPW>();
    __builtin_amdgcn_s_barrier();
    st_dma_issue<DH>(dma, smem + ((SLOT + 3) & 3) * STAGE, min(kt + 3, T - 1));
    if (!active) return;
    const unsigned char* Kimg = smem + SLOT * STAGE;
    const unsigned char* Vimg = Kimg + IMG;
    f32x4 s[2][2], dp[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        s[ks][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[ks][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kd = 0; kd < G::KD; ++kd) {
        const bf16x8 kf = st_frag_row<DH>(Kimg, ks, kd, lane);
        const bf16x8 vf = st_frag_row<DH>(Vimg, ks, kd, lane);
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
          s[ks][qs] = mfma16(kf, Qf[qs][kd], s[ks][qs]);
          dp[ks][qs] = mfma16(vf, dOf[qs][kd], dp[ks][qs]);
        }
      }
    const bool last = (kt == T - 1) && (p.n & 31);
    bf16x8 dsb[2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      f32x4 d0, d1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p0 = __builtin_amdgcn_exp2f(fmaf(s[0][qs][r], sc, -L2q[qs]));
        float p1 = __builtin_amdgcn_exp2f(fmaf(s[1][qs][r], sc, -L2q[qs]));
        if (last) {
          if ((kt * 32 + g * 4 + r) >= p.n) p0 = 0.f;
          if ((kt * 32 + 16 + g * 4 + r) >= p.n) p1 = 0.f;
        }
        d0[r] = p0 * (dp[0][qs][r] - Dq[qs]);
        d1[r] = p1 * (dp[1][qs][r] - Dq[qs]);
      }
      dsb[qs] = pack8(d0, d1);
    }
    bf16x4 klo[ND], khi[ND];  // (after the math: 20 registers less across it; the other waves of the SIMD cover the latency)
    st_tr_all<DH, SLOT * STAGE, 0>(ka, klo, khi);
    st_tr_retire<ND>(klo, khi);
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      const bf16x8 ktf = cat4(klo[dt], khi[dt]);
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) dQ[qs][dt] = mfma16(ktf, dsb[qs], dQ[qs][dt]);
    }
  };
  for (int kt = 0; kt < T; kt += 4) {
    body(SlotK<0>{}, kt);
    if (kt + 1 < T) body(SlotK<1>{}, kt + 1);
    if (kt + 2 < T) body(SlotK<2>{}, kt + 2);
    if (kt + 3 < T) body(SlotK<3>{}, kt + 3);
  }
  st_wait_vm<0>();
  if (!active) return;
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    const int t = q0 + qs * 16 + (lane & 15);
    if (t < p.n) {
      bf16_t* orow = p.dqkv + ((size_t)b * p.n + t) * p.ldq + h * DH + g * 4;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        bf16x4 o = {(bf16_t)(dQ[qs][dt][0] * p.scale), (bf16_t)(dQ[qs][dt][1] * p.scale),
                    (bf16_t)(dQ[qs][dt][2] * p.scale), (bf16_t)(dQ[qs][dt][3] * p.scale)};
        *reinterpret_cast<bf16x4*>(orow + dt * 16) = o;
      }
    }
  }
}

// ---- dK / dV -------------------------------------------------------------------------------------
// Own rows = 32 keys per wave (K, V fragments and both accumulators in registers); streamed: Q and dO tiles
// plus the 32 log-sum-exp / rowsum(dO o O) values of the tile's queries (one 256-byte piece, wave 0).
template <int DH>
__global__ __launch_bounds__(256, DH > 96 ? 1 : 2) void attn_bwd_dkdv_st_kernel(const AttnParams p) {
  using G = Geo<DH>;
  using S = SG<DH>;
  constexpr int IMG = S::IMG, STAGE = 2 * IMG + 256, PPW = S::PPW, ND = G::ND;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.y;
  const int T = (p.n + 31) >> 5;
  const size_t row_base = (size_t)bh * p.NP * G::DHP;
  const int key0 = blockIdx.x * 128 + wave * 32;
  const bool active = key0 < p.n;
  const int b = bh / p.H, h = bh - b * p.H;

  bf16x8 Kf[2][G::KD], Vf[2][G::KD];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int kd = 0; kd < G::KD; ++kd) {
      const size_t off = row_base + (size_t)(key0 + ks * 16 + (lane & 15)) * G::DHP + kd * 32 + g * 8;
      Kf[ks][kd] = *reinterpret_cast<const bf16x8*>(p.krow + off);
      Vf[ks][kd] = *reinterpret_cast<const bf16x8*>(p.vrow + off);
    }
  StDma<DH> dma;
  st_dma_init<DH>(dma, p.qrow + row_base, p.dorow + row_base, wave, lane);
  // the statistics piece: lanes 0-7 fetch 4 lse2 values each, lanes 8-15 4 dsum values each
  const float* stat_src = (lane < 8 ? p.lse2 : p.dsum) + (size_t)bh * p.NP + (lane & 7) * 4;
  auto issue = [&](int slot, int tile) {
    st_dma_issue<DH>(dma, smem + slot * STAGE, tile);
    if (wave == 0 && lane < 16)
      __builtin_amdgcn_global_load_lds((gbl_cvoid*)(stat_src + tile * 32), (lds_void*)(smem + slot * STAGE + 2 * IMG), 16,
                                       0, 0);
  };
  st_wait_vm<0>();
  issue(0, 0);
  issue(1, min(1, T - 1));
  issue(2, min(2, T - 1));
  const StTrAddr ta = st_tr_addr<DH>(st_lds_addr(smem), lane);
  f32x4 dK[2][ND], dV[2][ND];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      dK[ks][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dV[ks][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  bool kvalid[2];
  kvalid[0] = (key0 + (lane & 15)) < p.n;
  kvalid[1] = (key0 + 16 + (lane & 15)) < p.n;
  const float sc = p.scale * LOG2E;

  auto body = [&](auto slot_c, int qt) {
    constexpr int SLOT = decltype(slot_c)::value;
    if (wave == 0) st_wait_vm<2 * (PPW + 1)>();
    else st_wait_vm<2 * PPW>();
    __builtin_amdgcn_s_barrier();
    issue((SLOT + 3) & 3, min(qt + 3, T - 1));
    if (!active) return;
    const unsigned char* Qimg = smem + SLOT * STAGE;
    const unsigned char* dOimg = Qimg + IMG;
    const float* st = reinterpret_cast<const float*>(Qimg + 2 * IMG);
    f32x4 s[2][2], dp[2][2];  // [qsub][ksub]
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        s[qs][ks] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[qs][ks] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
      for (int kd = 0; kd < G::KD; ++kd) {
        const bf16x8 qf = st_frag_row<DH>(Qimg, qs, kd, lane);
        const bf16x8 df = st_frag_row<DH>(dOimg, qs, kd, lane);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s[qs][ks] = mfma16(qf, Kf[ks][kd], s[qs][ks]);
          dp[qs][ks] = mfma16(df, Vf[ks][kd], dp[qs][ks]);
        }
      }
    const f32x4 l2a = *reinterpret_cast<const f32x4*>(st + g * 4);
    const f32x4 l2b = *reinterpret_cast<const f32x4*>(st + 16 + g * 4);
    const f32x4 dda = *reinterpret_cast<const f32x4*>(st + 32 + g * 4);
    const f32x4 ddb = *reinterpret_cast<const f32x4*>(st + 48 + g * 4);
    const bool lastq = (qt == T - 1) && (p.n & 31);
    bf16x8 pb[2], dsb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 p0, p1, d0, d1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // padded keys AND padded query rows are masked (shared dO scratch: see the tiled kernel)
        bool ok0 = kvalid[ks], ok1 = kvalid[ks];
        if (lastq) {
          ok0 = ok0 && (qt * 32 + g * 4 + r) < p.n;
          ok1 = ok1 && (qt * 32 + 16 + g * 4 + r) < p.n;
        }
        p0[r] = ok0 ? __builtin_amdgcn_exp2f(fmaf(s[0][ks][r], sc, -l2a[r])) : 0.f;
        p1[r] = ok1 ? __builtin_amdgcn_exp2f(fmaf(s[1][ks][r], sc, -l2b[r])) : 0.f;
        d0[r] = p0[r] * (dp[0][ks][r] - dda[r]);
        d1[r] = p1[r] * (dp[1][ks][r] - ddb[r]);
      }
      pb[ks] = pack8(p0, p1);
      dsb[ks] = pack8(d0, d1);
    }
    bf16x4 qlo[ND], qhi[ND], olo[ND], ohi[ND];
    st_tr_all<DH, SLOT * STAGE + IMG, 0>(ta, olo, ohi);
    st_tr_all<DH, SLOT * STAGE, 0>(ta, qlo, qhi);
    st_tr_retire<ND>(olo, ohi);
    st_tr_retire<ND>(qlo, qhi);
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      const bf16x8 dof = cat4(olo[dt], ohi[dt]);
      const bf16x8 qtf = cat4(qlo[dt], qhi[dt]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        dV[ks][dt] = mfma16(dof, pb[ks], dV[ks][dt]);
        dK[ks][dt] = mfma16(qtf, dsb[ks], dK[ks][dt]);
      }
    }
  };
  for (int qt = 0; qt < T; qt += 4) {
    body(SlotK<0>{}, qt);
    if (qt + 1 < T) body(SlotK<1>{}, qt + 1);
    if (qt + 2 < T) body(SlotK<2>{}, qt + 2);
    if (qt + 3 < T) body(SlotK<3>{}, qt + 3);
  }
  st_wait_vm<0>();
  if (!active) return;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int t = key0 + ks * 16 + (lane & 15);
    if (t < p.n) {
      bf16_t* krow_o = p.dqkv + ((size_t)b * p.n + t) * p.ldq + p.hid + h * DH + g * 4;
      bf16_t* vrow_o = krow_o + p.hid;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        bf16x4 ok = {(bf16_t)(dK[ks][dt][0] * p.scale), (bf16_t)(dK[ks][dt][1] * p.scale),
                     (bf16_t)(dK[ks][dt][2] * p.scale), (bf16_t)(dK[ks][dt][3] * p.scale)};
        bf16x4 ov = {(bf16_t)dV[ks][dt][0], (bf16_t)dV[ks][dt][1], (bf16_t)dV[ks][dt][2],
                     (bf16_t)dV[ks][dt][3]};
        *reinterpret_cast<bf16x4*>(krow_o + dt * 16) = ok;
        *reinterpret_cast<bf16x4*>(vrow_o + dt * 16) = ov;
      }
    }
  }
}

// =============================================================================================
// Lean LDS-resident backward (round 4).  Same residency, fragment conventions and images as the
// resident kernels above; what changes is everything that is not an MFMA in the tile loop:
//   * no per-score masks: padded KEY rows of K / V are zero rows (attention.h), so a padded key
//     contributes 0 * finite to dQ, and a padded key's dK / dV rows are simply never stored;
//     padded / unsupervised QUERY rows are masked in the one query tile that holds them
//     (wave-uniform branch), because the dO scratch is shared by stacks of different geometry;
//   * dP - D comes out of the matrix pipe: the dP accumulator starts at -D[q] instead of 0;
//   * the tile loop is unrolled four times over ONE per-lane LDS base per image, so every
//     ds_read carries its tile / sub-tile / head-dim-step displacement in the 16-bit offset
//     field instead of eight per-iteration v_add_u32 address updates;
//   * template knob PK: the exponent FMA and dS = P o dP as packed fp32 pairs (attn_variant 7 / 8).
// Per 32 x 32 score tile and wave: dQ 34 MFMA + 16 v_exp + ~40 VALU (was 126 + 16),
// dK/dV 44 MFMA + 16 v_exp + ~50 VALU (was 148 + 16).
// =============================================================================================
template <int DHP>
struct R2 {
  static constexpr int ROWB = DHP * 2, TILE = 32 * ROWB, SUB = 16 * ROWB;
};
// per-lane byte offset of a row fragment inside a 16-row group of an img96 (add kd * 64, sub * SUB, tile * TILE)
template <int DHP>
DEVINL unsigned r2_row_off(int lane) {
  const int r = lane & 15;
  return (unsigned)(r * R2<DHP>::ROWB + (((lane >> 4) ^ res_g(r)) << 4));
}
// per-lane byte offsets of the token-contracting fragments of a 32-row tile: even / odd 16-dim tiles
// (add (dt >> 1) * 64, tile * TILE; second half of the tokens at + SUB)
template <int DHP>
DEVINL void r2_tr_off(int lane, unsigned& e, unsigned& o) {
  const int g = lane >> 4, s = lane & 15;
  const int r = g * 4 + (s >> 2), low = ((s & 3) >> 1) ^ res_g(r);
  e = (unsigned)(r * R2<DHP>::ROWB + low * 16 + (s & 1) * 8);
  o = (unsigned)(r * R2<DHP>::ROWB + (low ^ 2) * 16 + (s & 1) * 8);
}
DEVINL bf16x8 r2_ld_row(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }
template <int SUB>
DEVINL bf16x8 r2_ld_tr(const unsigned char* p) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + SUB));
  return cat4(lo, hi);
}
// P = exp2(s * sc + nl) for this lane's 4 scores of one 16 x 16 sub-tile, then dS = P o dp (dp already holds dP - D);
// nl = -L2[q]: one value per lane in the dQ kernel (queries are columns), one per register in dK/dV (queries are rows).
// PK = 0: single v_fma_f32 / v_mul_f32.  Each result passes through an EMPTY asm statement: hipcc's SLP pass otherwise
// pairs scores of DIFFERENT accumulators into v_pk_mul_f32 and pays for it with v_mov / v_perm / v_alignbit shuffles; the
// arithmetic itself stays a compiler instruction (an asm v_fma reading an MFMA result would skip the MFMA -> VALU wait
// states hipcc pads: measured NaN).  PK = 1: explicit aligned pairs (v_pk_fma_f32 / v_pk_mul_f32).
DEVINL float r2_opaque(float x) {
  asm("" : "+v"(x));
  return x;
}
template <int PK>
DEVINL void r2_p(const f32x4& s, float sc, const f32x4& nl, f32x4& pr) {
  if constexpr (PK) {
    const f32x2 sc2 = {sc, sc};
    const f32x2 a = __builtin_elementwise_fma(f32x2{s[0], s[1]}, sc2, f32x2{nl[0], nl[1]});
    const f32x2 b = __builtin_elementwise_fma(f32x2{s[2], s[3]}, sc2, f32x2{nl[2], nl[3]});
    pr[0] = __builtin_amdgcn_exp2f(a[0]); pr[1] = __builtin_amdgcn_exp2f(a[1]);
    pr[2] = __builtin_amdgcn_exp2f(b[0]); pr[3] = __builtin_amdgcn_exp2f(b[1]);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) pr[r] = __builtin_amdgcn_exp2f(r2_opaque(fmaf(s[r], sc, nl[r])));
  }
}
template <int PK>
DEVINL f32x4 r2_mul(const f32x4& a, const f32x4& b) {
  if constexpr (PK) {
    const f32x2 x = f32x2{a[0], a[1]} * f32x2{b[0], b[1]};
    const f32x2 y = f32x2{a[2], a[3]} * f32x2{b[2], b[3]};
    return f32x4{x[0], x[1], y[0], y[1]};
  } else {
    return f32x4{r2_opaque(a[0] * b[0]), r2_opaque(a[1] * b[1]), r2_opaque(a[2] * b[2]), r2_opaque(a[3] * b[3])};
  }
}

template <int DH, int PK>
__global__ __launch_bounds__(768) void attn_bwd_dq_r2_kernel(const AttnParams p) {
  using G = Geo<DH>;
  using R = R2<G::DHP>;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, g = lane >> 4;
  const int bh = blockIdx.x;
  const int rows = (p.n + 31) & ~31;
  const size_t img_bytes = ((size_t)rows * G::DHP * 2 + 1023) & ~(size_t)1023;
  unsigned char* Kimg = smem;
  unsigned char* Vimg = smem + img_bytes;
  const size_t row_base = (size_t)bh * p.NP * G::DHP;
  res_load<G::DHP, G::DHP / 8, true>(Kimg, p.krow + row_base, rows, wave, nw, lane);
  res_load<G::DHP, G::DHP / 8, true>(Vimg, p.vrow + row_base, rows, wave, nw, lane);

  const int q0 = wave * 32;
  const int b = bh / p.H, h = bh - b * p.H;
  if (p.nq > 0 && q0 >= p.nq) {
    // supervised-rows shortcut: dO is zero on these query rows, so is dQ (the QKV dgrad reads every row)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      const int t = q0 + qs * 16 + (lane & 15);
      if (t < p.n) {
        bf16_t* orow = p.dqkv + ((size_t)b * p.n + t) * p.ldq + h * DH + g * 4;
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) *reinterpret_cast<bf16x4*>(orow + dt * 16) = bf16x4{};
      }
    }
    return;
  }
  bf16x8 Qf[2][G::KD], dOf[2][G::KD];
  float nL2[2], nD[2];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    const int q = q0 + qs * 16 + (lane & 15);
    // D[q] = rowsum(dO o O): this lane holds 8 head columns of dO per 32-column slab, lanes g = 0..3 cover the slab
    float part = 0.f;
#pragma unroll
    for (int kd = 0; kd < G::KD; ++kd) {
      const size_t off = row_base + (size_t)q * G::DHP + kd * 32 + g * 8;
      Qf[qs][kd] = *reinterpret_cast<const bf16x8*>(p.qrow + off);
      dOf[qs][kd] = *reinterpret_cast<const bf16x8*>(p.dorow + off);
      const int d0 = kd * 32 + g * 8;
      if (q < p.n && d0 < DH) {
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(p.o + ((size_t)b * p.n + q) * p.ldo + h * DH + d0);
#pragma unroll
        for (int j = 0; j < 8; ++j) part += (float)dOf[qs][kd][j] * (float)ov[j];
      }
    }
    part = st_allsum4(part);
    nL2[qs] = -p.lse2[(size_t)bh * p.NP + q];
    nD[qs] = -part;
    if (g == 0) p.dsum[(size_t)bh * p.NP + q] = part;  // for the dK/dV kernel (0 on padded query rows)
  }
  f32x4 dQ[2][G::ND];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs)
#pragma unroll
    for (int dt = 0; dt < G::ND; ++dt) dQ[qs][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned trE, trO;
  r2_tr_off<G::DHP>(lane, trE, trO);
  const unsigned char* kr = Kimg + r2_row_off<G::DHP>(lane);  // per-lane bases, advanced 4 tiles at a time
  const unsigned char* vr = Vimg + r2_row_off<G::DHP>(lane);
  const unsigned char* ke = Kimg + trE;
  const unsigned char* ko = Kimg + trO;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float sc = p.scale * LOG2E;
  const int nkt = rows >> 5;
  // Padded keys (n % 32 != 0: the last key tile only).  Their K / V rows are zero rows, so the score is 0 and
  // P = exp2(-L2[q]) - finite for any realistic row, but +inf once L2[q] < -128, and inf * (0 - D) would reach dQ as
  // NaN through the dS . K product.  The one tile that holds them zeroes those dS entries by SELECT (wave-uniform
  // branch, one tile per head: free), which is what the masked round-2/3 kernels did for every score.
  const int padk_tile = (p.n & 31) ? nkt - 1 : -1;
  auto body = [&](auto jc, int kt) {
    constexpr int J = decltype(jc)::value;
    f32x4 s[2][2], dp[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        s[ks][qs] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[ks][qs] = f32x4{nD[qs], nD[qs], nD[qs], nD[qs]};
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kd = 0; kd < G::KD; ++kd) {
        const bf16x8 kf = r2_ld_row(kr + J * R::TILE + ks * R::SUB + kd * 64);
        const bf16x8 vf = r2_ld_row(vr + J * R::TILE + ks * R::SUB + kd * 64);
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
          s[ks][qs] = mfma16(kf, Qf[qs][kd], s[ks][qs]);
          dp[ks][qs] = mfma16(vf, dOf[qs][kd], dp[ks][qs]);
        }
      }
    bf16x8 dsb[2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      const f32x4 nl = {nL2[qs], nL2[qs], nL2[qs], nL2[qs]};
      f32x4 p0, p1;
      r2_p<PK>(s[0][qs], sc, nl, p0);
      r2_p<PK>(s[1][qs], sc, nl, p1);
      f32x4 d0 = r2_mul<PK>(p0, dp[0][qs]), d1 = r2_mul<PK>(p1, dp[1][qs]);
      if (kt == padk_tile) {  // S^T layout: this lane's 4 scores of sub-tile ks are keys tile * 32 + ks * 16 + g * 4 + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (kt * 32 + g * 4 + r >= p.n) d0[r] = 0.f;
          if (kt * 32 + 16 + g * 4 + r >= p.n) d1[r] = 0.f;
        }
      }
      dsb[qs] = pack8(d0, d1);
    }
#pragma unroll
    for (int dt = 0; dt < G::ND; ++dt) {
      const bf16x8 ktf = r2_ld_tr<R::SUB>(((dt & 1) ? ko : ke) + J * R::TILE + (dt >> 1) * 64);
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) dQ[qs][dt] = mfma16(ktf, dsb[qs], dQ[qs][dt]);
    }
  };
  for (int kt = 0; kt < nkt; kt += 4) {
    body(SlotK<0>{}, kt);
    if (kt + 1 < nkt) body(SlotK<1>{}, kt + 1);
    if (kt + 2 < nkt) body(SlotK<2>{}, kt + 2);
    if (kt + 3 < nkt) body(SlotK<3>{}, kt + 3);
    kr += 4 * R::TILE; vr += 4 * R::TILE; ke += 4 * R::TILE; ko += 4 * R::TILE;
  }
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    const int t = q0 + qs * 16 + (lane & 15);
    if (t < p.n) {
      bf16_t* orow = p.dqkv + ((size_t)b * p.n + t) * p.ldq + h * DH + g * 4;
#pragma unroll
      for (int dt = 0; dt < G::ND; ++dt) {
        bf16x4 o = {(bf16_t)(dQ[qs][dt][0] * p.scale), (bf16_t)(dQ[qs][dt][1] * p.scale),
                    (bf16_t)(dQ[qs][dt][2] * p.scale), (bf16_t)(dQ[qs][dt][3] * p.scale)};
        *reinterpret_cast<bf16x4*>(orow + dt * 16) = o;
      }
    }
  }
}

// dK/dV: 8 waves (2 per SIMD), 32-key slots dealt round-robin; per slot the wave walks the query tiles.
// The statistics live in LDS negated (-L2, -D): the dP accumulator starts from the loaded -D vector.
template <int DH, int PK>
__global__ __launch_bounds__(512) void attn_bwd_dkdv_r2_kernel(const AttnParams p) {
  using G = Geo<DH>;
  using R = R2<G::DHP>;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, g = lane >> 4;
  const int bh = blockIdx.x;
  const int rows = (p.n + 31) & ~31;
  const size_t img_bytes = ((size_t)rows * G::DHP * 2 + 1023) & ~(size_t)1023;
  unsigned char* Qimg = smem;
  unsigned char* dOimg = smem + img_bytes;
  float* nL2s = reinterpret_cast<float*>(smem + 2 * img_bytes);
  float* nDs = nL2s + rows;
  const size_t row_base = (size_t)bh * p.NP * G::DHP;
  res_load<G::DHP, G::DHP / 8, true>(Qimg, p.qrow + row_base, rows, wave, nw, lane);
  res_load<G::DHP, G::DHP / 8, true>(dOimg, p.dorow + row_base, rows, wave, nw, lane);
  for (int i = tid; i < rows; i += blockDim.x) {
    nL2s[i] = -p.lse2[(size_t)bh * p.NP + i];
    nDs[i] = -p.dsum[(size_t)bh * p.NP + i];
  }
  unsigned trE, trO;
  r2_tr_off<G::DHP>(lane, trE, trO);
  const unsigned rowoff = r2_row_off<G::DHP>(lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float sc = p.scale * LOG2E;
  const int nqt = rows >> 5;
  const int nqv = (p.nq > 0 && p.nq < p.n) ? p.nq : p.n;  // valid query rows (supervised-rows shortcut: dO rows >= nq are zero)
  const int nfull = nqv >> 5;  // query tiles without padded / unsupervised rows
  const int b = bh / p.H, h = bh - b * p.H;
  for (int slot = wave; slot < nqt; slot += nw) {
    const int key0 = slot * 32;
    bf16x8 Kf[2][G::KD], Vf[2][G::KD];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kd = 0; kd < G::KD; ++kd) {
        const size_t off = row_base + (size_t)(key0 + ks * 16 + (lane & 15)) * G::DHP + kd * 32 + g * 8;
        Kf[ks][kd] = *reinterpret_cast<const bf16x8*>(p.krow + off);
        Vf[ks][kd] = *reinterpret_cast<const bf16x8*>(p.vrow + off);
      }
    f32x4 dK[2][G::ND], dV[2][G::ND];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dt = 0; dt < G::ND; ++dt) {
        dK[ks][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dV[ks][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    const unsigned char* qr = Qimg + rowoff;
    const unsigned char* dr = dOimg + rowoff;
    const unsigned char* qe = Qimg + trE;
    const unsigned char* qo = Qimg + trO;
    const unsigned char* de = dOimg + trE;
    const unsigned char* dd = dOimg + trO;
    const float* st = nL2s + g * 4;  // this lane's 4 query rows of a 16-row sub-tile
    const float* sd = nDs + g * 4;
    // J: tile inside the group of four the bases point at; MASK: the one tile that holds padded / unsupervised query
    // rows (peeled off the loop: the dO scratch is shared by stacks of different geometry, its padding is not zero)
    auto body = [&](auto jc, auto mc, int qt) {
      constexpr int J = decltype(jc)::value;
      constexpr bool MASK = decltype(mc)::value != 0;
      f32x4 nl[2], s[2][2], dp[2][2];  // [qsub][ksub]
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        nl[qs] = *reinterpret_cast<const f32x4*>(st + J * 32 + qs * 16);
        const f32x4 nd = *reinterpret_cast<const f32x4*>(sd + J * 32 + qs * 16);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s[qs][ks] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[qs][ks] = nd;
        }
      }
#pragma unroll
      for (int qs = 0; qs < 2; ++qs)
#pragma unroll
        for (int kd = 0; kd < G::KD; ++kd) {
          const bf16x8 qf = r2_ld_row(qr + J * R::TILE + qs * R::SUB + kd * 64);
          const bf16x8 df = r2_ld_row(dr + J * R::TILE + qs * R::SUB + kd * 64);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            s[qs][ks] = mfma16(qf, Kf[ks][kd], s[qs][ks]);
            dp[qs][ks] = mfma16(df, Vf[ks][kd], dp[qs][ks]);
          }
        }
      bf16x8 pb[2], dsb[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x4 p0, p1;
        r2_p<PK>(s[0][ks], sc, nl[0], p0);
        r2_p<PK>(s[1][ks], sc, nl[1], p1);
        if constexpr (MASK) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if ((qt * 32 + g * 4 + r) >= nqv) p0[r] = 0.f;
            if ((qt * 32 + 16 + g * 4 + r) >= nqv) p1[r] = 0.f;
          }
        }
        pb[ks] = pack8(p0, p1);
        dsb[ks] = pack8(r2_mul<PK>(p0, dp[0][ks]), r2_mul<PK>(p1, dp[1][ks]));
      }
#pragma unroll
      for (int dt = 0; dt < G::ND; ++dt) {
        const bf16x8 dof = r2_ld_tr<R::SUB>(((dt & 1) ? dd : de) + J * R::TILE + (dt >> 1) * 64);
        const bf16x8 qtf = r2_ld_tr<R::SUB>(((dt & 1) ? qo : qe) + J * R::TILE + (dt >> 1) * 64);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          dV[ks][dt] = mfma16(dof, pb[ks], dV[ks][dt]);
          dK[ks][dt] = mfma16(qtf, dsb[ks], dK[ks][dt]);
        }
      }
    };
    for (int qt = 0; qt < nfull; qt += 4) {
      body(SlotK<0>{}, SlotK<0>{}, qt);
      if (qt + 1 < nfull) body(SlotK<1>{}, SlotK<0>{}, qt + 1);
      if (qt + 2 < nfull) body(SlotK<2>{}, SlotK<0>{}, qt + 2);
      if (qt + 3 < nfull) body(SlotK<3>{}, SlotK<0>{}, qt + 3);
      qr += 4 * R::TILE; dr += 4 * R::TILE; qe += 4 * R::TILE; qo += 4 * R::TILE; de += 4 * R::TILE; dd += 4 * R::TILE;
      st += 128; sd += 128;
    }
    if (nqv & 31) {
      const int back = ((nfull + 3) & ~3) - nfull;  // the bases stand at the next multiple of four tiles
      qr -= back * R::TILE; dr -= back * R::TILE; qe -= back * R::TILE; qo -= back * R::TILE; de -= back * R::TILE;
      dd -= back * R::TILE; st -= back * 32; sd -= back * 32;
      body(SlotK<0>{}, SlotK<1>{}, nfull);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int t = key0 + ks * 16 + (lane & 15);
      if (t < p.n) {
        bf16_t* krow_o = p.dqkv + ((size_t)b * p.n + t) * p.ldq + p.hid + h * DH + g * 4;
        bf16_t* vrow_o = krow_o + p.hid;
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) {
          bf16x4 ok = {(bf16_t)(dK[ks][dt][0] * p.scale), (bf16_t)(dK[ks][dt][1] * p.scale),
                       (bf16_t)(dK[ks][dt][2] * p.scale), (bf16_t)(dK[ks][dt][3] * p.scale)};
          bf16x4 ov = {(bf16_t)dV[ks][dt][0], (bf16_t)dV[ks][dt][1], (bf16_t)dV[ks][dt][2],
                       (bf16_t)dV[ks][dt][3]};
          *reinterpret_cast<bf16x4*>(krow_o + dt * 16) = ok;
          *reinterpret_cast<bf16x4*>(vrow_o + dt * 16) = ov;
        }
      }
    }
  }
}

// Kernel families (round 6: pruned to the default and one reference path per op, round-5 review item 7):
//   5 (default) = streaming forward (480 workgroups, runs alone on the chip) + LEAN LDS-resident backward where the head fits
//                 (n <= 384, head dim <= 96: runs beside the 95-workgroup wgrad launches and leaves them the CUs), streaming
//                 backward otherwise (the scaled configuration: n = 1440, head dim 128);
//   2           = streaming kernels everywhere (A/B knob);
//   force_tiled = the tiled kernels of round 1 (standard online softmax, no lazy maximum): the forced reference path.
// Rounds 2-5 also carried the round-2/3 resident family (variants 1 / 3 / 4 / 6) and packed-fp32 forms of the lean backward
// (7 / 8: slower, profiles/r04_ab_attn_variants.txt); any other value now selects the default.
int g_attn_variant = 5;
int g_attn_force_tiled = 0;  // test knob: 1 = always use the tiled (streaming) kernels

// LDS bytes of the lean resident backward kernels for n tokens (1 = dQ, 2 = dK/dV); 0 = does not fit -> streaming kernels
template <int DH>
size_t res_lds_bytes(int n, int which /*0 fwd, 1 dq, 2 dkdv*/) {
  using G = Geo<DH>;
  const size_t rows = (size_t)((n + 31) & ~31);
  const size_t img = (rows * G::DHP * 2 + 1023) & ~(size_t)1023;
  size_t b = 0;
  if (which == 0) b = img + ((rows * DH * 2 + 1023) & ~(size_t)1023);
  else if (which == 1) b = 2 * img;
  else b = 2 * img + rows * 8;
  // <= 12 waves (768 threads, 168 registers): head dims above 96 need more registers than that (the DH = 128
  // instantiations spilled 150-290 B per lane) and take the tiled kernels, which have no such limit
  if (DH > 96 || rows > 384 || b > 160 * 1024 || g_attn_force_tiled) return 0;
  return b;
}

int check(const AttnParams& p) {
  if (p.B <= 0 || p.H <= 0 || p.n <= 0) return -1;
  if (p.NP % 128 != 0 || p.NP < p.n) return -2;
  if (p.hid != p.H * p.dh) return -3;
  if (p.hid % 4 != 0) return -4;
  if (p.ldo < p.hid || (p.ldo & 3) || (p.dqkv && (p.ldq < 3 * p.hid || (p.ldq & 3)))) return -6;
  return 0;
}

template <typename K>
int allow_big_lds(K kernel, bool* done) {
  if (!*done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return -20;
    *done = true;
  }
  return 0;
}

template <typename K>
int allow_lds_bytes(K kernel, bool* done, size_t bytes) {
  if (!*done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
      return -20;
    *done = true;
  }
  return 0;
}

// does launch_attn_bwd run the lean LDS-resident kernels for n tokens?  They honour AttnParams::nq; the streaming backward
// (variant 2, or whatever the resident kernels cannot hold) does not.
template <int DH>
bool bwd_is_resident(int n) {
  return g_attn_variant != 2 && !g_attn_force_tiled && res_lds_bytes<DH>(n, 1) && res_lds_bytes<DH>(n, 2);
}

template <int DH>
int fwd_t(const AttnParams& p, hipStream_t s) {
  if (!g_attn_force_tiled) {  // streaming kernel: the forward of every shape
    constexpr size_t lds = (size_t)SG<DH>::NSLOT * 2 * SG<DH>::IMG;
    static bool attr = false;
    if (int rc = allow_lds_bytes(attn_fwd_st_kernel<DH>, &attr, lds)) return rc;
    // supervised-rows shortcut: only where the backward that may follow is nq-aware too (the resident kernels; the streaming
    // backward reads the log-sum-exp of every row)
    AttnParams q = p;
    if (!bwd_is_resident<DH>(p.n)) q.nq = 0;
    const int nqv = (q.nq > 0 && q.nq < q.n) ? q.nq : q.n;
    q.qblocks = (nqv + 127) / 128;
    FACT_LAUNCH(attn_fwd_st_kernel<DH>, dim3(q.qblocks * q.B * q.H), dim3(256), lds, s, q);
    return 0;
  }
  dim3 grid((p.n + 127) / 128, p.B * p.H);  // forced reference path
  FACT_LAUNCH(attn_fwd_kernel<DH>, grid, dim3(256), 0, s, p);
  return 0;
}
template <int DH>
int bwd_t(const AttnParams& p, hipStream_t s) {
  const dim3 grid((p.n + 127) / 128, p.B * p.H);
  if (g_attn_force_tiled) {  // forced reference path: D = rowsum(dO o O) pass, then the tiled dQ and dK/dV kernels
    const int total = p.B * p.H * p.NP;
    FACT_LAUNCH(attn_bwd_prep_kernel<DH>, dim3((total + 255) / 256), dim3(256), 0, s, p);
    FACT_LAUNCH(attn_bwd_dq_kernel<DH>, grid, dim3(256), 0, s, p);
    FACT_LAUNCH(attn_bwd_dkdv_kernel<DH>, grid, dim3(256), 0, s, p);
    return 0;
  }
  if constexpr (DH <= 96) {
    if (bwd_is_resident<DH>(p.n)) {  // lean LDS-resident backward (the dQ kernel computes D itself)
      const size_t lds1 = res_lds_bytes<DH>(p.n, 1), lds2 = res_lds_bytes<DH>(p.n, 2);
      const int nw = ((p.n + 31) & ~31) / 32;
      static bool attr1 = false, attr2 = false;
      if (int rc = allow_big_lds(attn_bwd_dq_r2_kernel<DH, 0>, &attr1)) return rc;
      if (int rc = allow_big_lds(attn_bwd_dkdv_r2_kernel<DH, 0>, &attr2)) return rc;
      FACT_LAUNCH((attn_bwd_dq_r2_kernel<DH, 0>), dim3(p.B * p.H), dim3(nw * 64), lds1, s, p);
      FACT_LAUNCH((attn_bwd_dkdv_r2_kernel<DH, 0>), dim3(p.B * p.H), dim3((nw < 8 ? nw : 8) * 64), lds2, s, p);
      return 0;
    }
  }
  constexpr size_t lds1 = (size_t)SG<DH>::NSLOT * 2 * SG<DH>::IMG, lds2 = (size_t)SG<DH>::NSLOT * (2 * SG<DH>::IMG + 256);
  static bool attr1 = false, attr2 = false;
  if (int rc = allow_lds_bytes(attn_bwd_dq_st_kernel<DH>, &attr1, lds1)) return rc;
  if (int rc = allow_lds_bytes(attn_bwd_dkdv_st_kernel<DH>, &attr2, lds2)) return rc;
  FACT_LAUNCH(attn_bwd_dq_st_kernel<DH>, grid, dim3(256), lds1, s, p);
  FACT_LAUNCH(attn_bwd_dkdv_st_kernel<DH>, grid, dim3(256), lds2, s, p);
  return 0;
}

}  // namespace

void attn_set_force_tiled(int on) { g_attn_force_tiled = on; }
void attn_set_variant(int v) { g_attn_variant = (v == 2) ? 2 : 5; }
int attn_get_variant() { return g_attn_variant; }

int launch_attn_fwd(const AttnParams& p_in, hipStream_t s) {
  const AttnParams& p = p_in;
  int rc = check(p);
  if (rc) return rc;
  switch (p.dh) {
    case 32: return fwd_t<32>(p, s);
    case 64: return fwd_t<64>(p, s);
    case 80: return fwd_t<80>(p, s);
    case 128: return fwd_t<128>(p, s);
  }
  return -10;
}

int launch_attn_bwd(const AttnParams& p, hipStream_t s) {
  int rc = check(p);
  if (rc) return rc;
  switch (p.dh) {
    case 32: return bwd_t<32>(p, s);
    case 64: return bwd_t<64>(p, s);
    case 80: return bwd_t<80>(p, s);
    case 128: return bwd_t<128>(p, s);
  }
  return -10;
}
