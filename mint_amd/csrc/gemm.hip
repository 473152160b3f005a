// bf16 MFMA GEMM engine (see gemm.h).  128x128x64 block tile, 256 threads = 4 waves (2x2),
// each wave a 64x64 sub-tile = 4x4 MFMA 16x16x32 tiles.
//
// Fast kernels (gemm_*_fast): operand tiles go HBM -> LDS directly with 16-byte LDS-DMA loads
//   (global_load_lds_dwordx4: destination = wave-uniform base + lane*16, so the XOR swizzle is
//   applied to the per-lane SOURCE address and to the fragment reads), double-buffered LDS, the
//   next tile's DMA in flight under the current tile's MFMAs, one vmcnt(0)+barrier per K step.
//   Out-of-range rows are clamped (their results are never stored), so no zero-fill is needed.
// Generic kernels (gemm_*_generic): register-staged with zero-fill, any shape (fallback).
//
// MFMA roles are swapped (weights/B tile as the A operand) so that every lane ends up holding
// 4 CONSECUTIVE OUTPUT COLUMNS of one output row: epilogues load bias/residual and store results
// with 8/16-byte vector accesses.
#include "gemm.h"

#include <cmath>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

// One lane's share of the epilogue: output row `row`, columns col0..col0+3 (col0 % 4 == 0).
template <int EPI>
DEVINL void epilogue_store(const EpiParams& ep, int M, int N, int row, int col0, f32x4 v) {
  if (row >= M || col0 >= N) return;
  const bool full = (col0 + 3 < N);
  if constexpr (EPI == EPI_BF16) {
    bf16_t* o = (bf16_t*)ep.out0 + (size_t)row * ep.ldo0 + col0;
    if (full && !(ep.ldo0 & 3)) {
      bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      *reinterpret_cast<bf16x4*>(o) = pk;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col0 + r < N) o[r] = (bf16_t)v[r];
    }
  } else if constexpr (EPI == EPI_F32_BIAS || EPI == EPI_F32_BIAS_POS || EPI == EPI_F32_BIAS_RESID) {
    float* o = (float*)ep.out0 + (size_t)row * ep.ldo0 + col0;
    const float* extra = nullptr;
    if constexpr (EPI == EPI_F32_BIAS_POS) extra = ep.pos + (size_t)(row % ep.seq) * N + col0;
    if constexpr (EPI == EPI_F32_BIAS_RESID) extra = ep.resid + (size_t)row * ep.ldr + col0;
    if (full && !(ep.ldo0 & 3) && !(N & 3) && !(ep.ldr & 3)) {
      float4 b = ep.bias ? *reinterpret_cast<const float4*>(ep.bias + col0) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 r = make_float4(v[0] + b.x, v[1] + b.y, v[2] + b.z, v[3] + b.w);
      if (extra) {
        const float4 e = *reinterpret_cast<const float4*>(extra);
        r.x += e.x; r.y += e.y; r.z += e.z; r.w += e.w;
      }
      *reinterpret_cast<float4*>(o) = r;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col0 + r < N) o[r] = v[r] + (ep.bias ? ep.bias[col0 + r] : 0.f) + (extra ? extra[r] : 0.f);
    }
  } else if constexpr (EPI == EPI_BIAS_GELU) {
    bf16_t* o0 = (bf16_t*)ep.out0 + (size_t)row * ep.ldo0 + col0;
    bf16_t* o1 = (bf16_t*)ep.out1 + (size_t)row * ep.ldo1 + col0;
    float pre[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pre[r] = v[r] + ((col0 + r < N) ? ep.bias[col0 + r] : 0.f);
    if (full && !(ep.ldo0 & 3) && !(ep.ldo1 & 3)) {
      bf16x4 p0 = {(bf16_t)pre[0], (bf16_t)pre[1], (bf16_t)pre[2], (bf16_t)pre[3]};
      bf16x4 p1 = {(bf16_t)gelu_tanh(pre[0]), (bf16_t)gelu_tanh(pre[1]), (bf16_t)gelu_tanh(pre[2]),
                   (bf16_t)gelu_tanh(pre[3])};
      if (ep.out0) *reinterpret_cast<bf16x4*>(o0) = p0;
      *reinterpret_cast<bf16x4*>(o1) = p1;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col0 + r < N) {
          if (ep.out0) o0[r] = (bf16_t)pre[r];
          o1[r] = (bf16_t)gelu_tanh(pre[r]);
        }
    }
  } else if constexpr (EPI == EPI_GELU_BWD) {
    bf16_t* o = (bf16_t*)ep.out0 + (size_t)row * ep.ldo0 + col0;
    const bf16_t* pp = ep.pre + (size_t)row * ep.ldp + col0;
    if (full && !(ep.ldo0 & 3) && !(ep.ldp & 3)) {
      const bf16x4 pv = *reinterpret_cast<const bf16x4*>(pp);
      bf16x4 pk = {(bf16_t)(v[0] * gelu_tanh_grad((float)pv[0])), (bf16_t)(v[1] * gelu_tanh_grad((float)pv[1])),
                   (bf16_t)(v[2] * gelu_tanh_grad((float)pv[2])), (bf16_t)(v[3] * gelu_tanh_grad((float)pv[3]))};
      *reinterpret_cast<bf16x4*>(o) = pk;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col0 + r < N) o[r] = (bf16_t)(v[r] * gelu_tanh_grad((float)pp[r]));
    }
  } else if constexpr (EPI == EPI_HEADS) {
    // column c -> (which, h, d); dh % 4 == 0 so the 4 columns share (which, h)
    const int which = col0 / ep.hid;
    const int rem = col0 - which * ep.hid;
    const int h = rem / ep.dh;
    const int d = rem - h * ep.dh;
    const int b = row / ep.n_tok;
    const int t = row - b * ep.n_tok;
    // three-way select on scalar pointers: indexing the by-value kernarg array with a per-lane value
    // makes hipcc spill the whole EpiParams to scratch (168-344 B/lane in round 1)
    // (and the three loads go through an opaque asm: otherwise LLVM folds the select of constant-index loads back
    // into ONE dynamically indexed load and copies the kernarg struct to scratch after all - 176-392 B per lane)
    unsigned long long h0 = (unsigned long long)ep.hrow[0], h1 = (unsigned long long)ep.hrow[1],
                       h2 = (unsigned long long)ep.hrow[2];
    asm volatile("" : "+s"(h0), "+s"(h1), "+s"(h2));
    bf16_t* hr = reinterpret_cast<bf16_t*>((which == 0) ? h0 : (which == 1) ? h1 : h2);
    if (hr) {
      bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      *reinterpret_cast<bf16x4*>(hr + ((size_t)(b * ep.heads + h) * ep.n_pad + t) * ep.dhp + d) = pk;
    }
  } else if constexpr (EPI == EPI_F32_BF16) {
    float* o0 = (float*)ep.out0 + (size_t)row * ep.ldo0 + col0;
    bf16_t* o1 = (bf16_t*)ep.out1 + (size_t)row * ep.ldo1 + col0;
    if (full && !(ep.ldo0 & 3) && !(ep.ldo1 & 3)) {
      *reinterpret_cast<float4*>(o0) = make_float4(v[0], v[1], v[2], v[3]);
      bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      *reinterpret_cast<bf16x4*>(o1) = pk;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col0 + r < N) {
          o0[r] = v[r];
          o1[r] = (bf16_t)v[r];
        }
    }
  } else if constexpr (EPI == EPI_F32_SLAB) {
    float* o = (float*)ep.out0 + (size_t)ep.z * ep.slab_stride + (size_t)row * ep.ldo0 + col0;
    if (full && !(ep.ldo0 & 3)) {
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col0 + r < N) o[r] = v[r];
    }
  } else if constexpr (EPI == EPI_ATOMIC_F32) {
    float* o = (float*)ep.out0 + (size_t)row * ep.ldo0 + col0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col0 + r < N) atomicAdd(o + r, v[r] * ep.alpha);
  }
}

// The atomic (split-K wgrad) epilogue keeps the un-swapped MFMA roles: a lane then holds 4 consecutive
// ROWS of one column and the 16 lanes of a group hit 16 consecutive floats -> coalesced atomics.
template <int EPI>
constexpr bool kSwap = (EPI != EPI_ATOMIC_F32);

// swapped:    acc[i][j] = rows m0+wm*64+i*16+(lane&15),       cols n0+wn*64+j*16+(lane>>4)*4 + r
// un-swapped: acc[i][j] = rows m0+wm*64+i*16+(lane>>4)*4 + r,  cols n0+wn*64+j*16+(lane&15)
template <int EPI>
DEVINL void run_epilogue(const GemmParams& p, f32x4 (&acc)[4][4], int m0, int n0, int wm, int wn, int lane,
                         int z = 0) {
  EpiParams ep = p.ep;
  ep.z = z;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (kSwap<EPI>) {
        const int row = m0 + wm * 64 + i * 16 + (lane & 15);
        const int col0 = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
        epilogue_store<EPI>(ep, p.M, p.N, row, col0, acc[i][j]);
      } else {
        const int row0 = m0 + wm * 64 + i * 16 + (lane >> 4) * 4;
        const int col = n0 + wn * 64 + j * 16 + (lane & 15);
        if (col < p.N) {
          float* o = (float*)p.ep.out0 + (size_t)row0 * p.ep.ldo0 + col;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) atomicAdd(o + (size_t)r * p.ep.ldo0, acc[i][j][r] * p.ep.alpha);
        }
      }
    }
  }
}

// Block -> (m-tile, n-tile, k-split).  The 1-D grid is re-dealt so that each of the 8 XCDs (block b
// runs on XCD b % 8, observed) owns a CONTIGUOUS range of logical tile ids: neighbouring tiles share
// A-row / B-column panels, so the re-reads hit that XCD's private 4 MiB L2 instead of the fabric.
// Bijective for any grid size (cdna guide T1).  Logical id order: k-split fastest, then n, then m.
struct BlockCoord {
  int m0, n0, z;
};
DEVINL BlockCoord block_coord(const GemmParams& p) {
  const int tn = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int orig = blockIdx.x;
  const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  BlockCoord c;
  c.z = id % p.splitk;
  const int t = id / p.splitk;
  if (p.band > 1) {
    // bands of `band` m-tiles, m fastest inside a band: the ~64 tiles an XCD runs at once form a
    // near-square patch, so they share band + 64/band operand slabs per K step instead of tn + 64/tn
    const int tm = (p.M + BM - 1) / BM;
    const int per = p.band * tn;
    const int b = t / per, w = t - b * per;
    const int hb = min(p.band, tm - b * p.band);
    c.m0 = (b * p.band + w % hb) * BM;
    c.n0 = (w / hb) * BN;
  } else {
    c.n0 = (t % tn) * BN;
    c.m0 = (t / tn) * BM;
  }
  return c;
}

DEVINL void split_range(const GemmParams& p, int z, int& kt_beg, int& kt_end) {
  const int ktiles = (p.K + BK - 1) / BK;
  const int per = (ktiles + p.splitk - 1) / p.splitk;
  kt_beg = z * per;
  kt_end = min(ktiles, kt_beg + per);
}

// ---- LDS images ---------------------------------------------------------------------------------
// NT tile [128 rows][64 k]: 128-byte rows; logical 16-byte chunk c of row r sits at chunk position
//   c ^ (r & 7): the 16 lanes of a ds_read_b128 service group hit 16 distinct 16-byte bank slots.
// TN tile [64 k][128 m]: 256-byte rows; logical 32-byte unit u of row k sits at unit position
//   u ^ f(k), f(k) = (k&3) | ((k>>3)&1)<<2: the 8 row-segments one ds_read_b64_tr_b16 service group
//   touches fall in 8 distinct 32-byte bank slots.
DEVINL int tn_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

DEVINL bf16x8 nt_frag(const unsigned char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}

// TN fragment via the LDS transpose read: for k-slot group g = lane>>4, lane s = lane&15 supplies
// the address of 4 contiguous m-elements of row k = g*8 + hh*4 + (s>>2), columns (s&3)*4..+3; the
// hardware hands lane c the 4 k-values of column c (verified by tests/test_gpu_ops.py probe).
//
// The read is issued as inline asm: hipcc puts a full `s_waitcnt vmcnt(0)` in front of the
// __builtin_amdgcn_ds_read_tr16_b64 builtin whenever an LDS-DMA load is in flight (it cannot tell the
// two LDS buffers apart), which serialises the next stage's DMA with this stage's MFMAs.  The asm form
// is invisible to that pass; its completion is tracked by hand (tn_wait_lds + sched_barrier).
template <int OFF>
DEVINL bf16x4 tr_read(unsigned addr) {
  bf16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int N>
DEVINL void tn_wait_lds() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
DEVINL unsigned lds_addr(const unsigned char* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
// Per-lane byte offset (inside a [64 k][ROWB bytes] operand tile) of unit u's fragment reads:
//   k = ks*32 + g*8 + hh*4 + (s>>2);  f(k) = (s>>2)&3 | (g&1)<<2 does not depend on ks / hh, so
//   offset(u, ks, hh) = lane_part + ((u ^ f) << 5) + ks*32*ROWB + hh*4*ROWB  (the last two are immediates).
template <int ROWB>
DEVINL unsigned tn_lane_off(int u, int lane) {
  const int g = lane >> 4, s = lane & 15;
  const int f = ((s >> 2) & 3) | ((g & 1) << 2);
  return (unsigned)((g * 8 + (s >> 2)) * ROWB + (s & 3) * 8 + ((u ^ f) << 5));
}
template <int ROWB, int KS>
DEVINL bf16x8 tn_frag_at(unsigned addr) {
  const bf16x4 lo = tr_read<KS * 32 * ROWB>(addr);
  const bf16x4 hi = tr_read<KS * 32 * ROWB + 4 * ROWB>(addr);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// One 128x128x64 TN stage: the transpose reads of both 32-deep sub-steps are issued up front (30 of 32:
// lgkmcnt saturates at 15), the first sub-step's MFMAs start when its 16 reads have landed, the second's
// reads complete underneath them.
template <bool SWAP>
DEVINL void compute_tile_tn(const unsigned char* As, const unsigned char* Bs, f32x4 (&acc)[4][4], int wm, int wn,
                            int lane) {
  const unsigned a0 = lds_addr(As), b0 = lds_addr(Bs);
  unsigned aa[4], ba[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aa[i] = a0 + tn_lane_off<256>(wm * 4 + i, lane);
    ba[i] = b0 + tn_lane_off<256>(wn * 4 + i, lane);
  }
  bf16x8 af0[4], bf0[4], af1[4], bf1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) af0[i] = tn_frag_at<256, 0>(aa[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) bf0[i] = tn_frag_at<256, 0>(ba[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) af1[i] = tn_frag_at<256, 1>(aa[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) bf1[i] = tn_frag_at<256, 1>(ba[i]);
  tn_wait_lds<14>();  // lgkmcnt is a 4-bit in-order counter: <= 14 outstanding <=> the 16 sub-step-0 reads landed
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[i][j] = SWAP ? mfma16(bf0[j], af0[i], acc[i][j]) : mfma16(af0[i], bf0[j], acc[i][j]);
  __builtin_amdgcn_sched_barrier(0);  // keep the sub-step-0 MFMAs above the second wait
  bf1[3] = tn_frag_at<256, 1>(ba[3]);
  tn_wait_lds<0>();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[i][j] = SWAP ? mfma16(bf1[j], af1[i], acc[i][j]) : mfma16(af1[i], bf1[j], acc[i][j]);
}

// builtin form, kept for the generic (register-staged, no LDS-DMA in flight) kernel
DEVINL bf16x8 tn_frag(const unsigned char* tile, int ks, int u, int lane) {
  const int g = lane >> 4, s = lane & 15;
  bf16x4 r[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int k = ks * 32 + g * 8 + hh * 4 + (s >> 2);
    const int off = k * 256 + ((u ^ tn_f(k)) << 5) + (s & 3) * 8;
    r[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + off));
  }
  bf16x8 f = {r[0][0], r[0][1], r[0][2], r[0][3], r[1][0], r[1][1], r[1][2], r[1][3]};
  return f;
}

template <bool TN, bool SWAP>
DEVINL void compute_tile(const unsigned char* As, const unsigned char* Bs, f32x4 (&acc)[4][4], int nks,
                         int wm, int wn, int lane) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    if (ks < nks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (TN) af[i] = tn_frag(As, ks, wm * 4 + i, lane);
        else af[i] = nt_frag(As, wm * 64 + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (TN) bfr[j] = tn_frag(Bs, ks, wn * 4 + j, lane);
        else bfr[j] = nt_frag(Bs, wn * 64 + j * 16 + (lane & 15), ks * 4 + (lane >> 4));
      }
      // swapped roles: D[row = n-local][col = m-local]; un-swapped: D[row = m-local][col = n-local]
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = SWAP ? mfma16(bfr[j], af[i], acc[i][j]) : mfma16(af[i], bfr[j], acc[i][j]);
    }
  }
}

// Full-line epilogue of the 128x128 NT kernel (same idea as in the big-tile kernel): each wave re-shapes
// its 64x64 accumulator tile in a private XOR-swizzled piece of the idle stage buffers and stores 16 bytes
// per lane, so a wave store covers whole 128-byte (bf16) / 256-byte (fp32) row segments instead of
// 16 rows x 32/64 bytes; the fp32 residual is read in the same pattern.  Returns false when the output
// geometry is not 16-byte granular (the caller then takes the direct path).
template <int EPI>
DEVINL bool staged_epilogue_128(const GemmParams& p, f32x4 (&acc)[4][4], unsigned char* smem, int m0, int n0,
                                int wave, int wm, int wn, int lane) {
  const EpiParams& ep = p.ep;
  const int wrow0 = m0 + wm * 64, wcol0 = n0 + wn * 64;
  if constexpr (EPI == EPI_BF16) {
    if ((p.N & 7) || (ep.ldo0 & 7)) return false;
    unsigned char* stg = smem + wave * 8192;  // [64 rows][128 B]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lr = i * 16 + (lane & 15), lc = j * 16 + (lane >> 4) * 4;
        const f32x4 v = acc[i][j];
        const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(stg + lr * 128 + (((lc >> 3) ^ (lr & 7)) << 4) + (lc & 7) * 2) = o;
      }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int lr = rr * 8 + (lane >> 3), ch = lane & 7;
      const int row = wrow0 + lr, col = wcol0 + ch * 8;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + lr * 128 + ((ch ^ (lr & 7)) << 4));
      if (row < p.M && col < p.N) *reinterpret_cast<bf16x8*>((bf16_t*)ep.out0 + (size_t)row * ep.ldo0 + col) = v;
    }
    return true;
  } else if constexpr (EPI == EPI_F32_BIAS || EPI == EPI_F32_BIAS_POS || EPI == EPI_F32_BIAS_RESID ||
                       EPI == EPI_F32_SLAB) {
    if ((p.N & 3) || (ep.ldo0 & 3) || (EPI == EPI_F32_BIAS_RESID && (ep.ldr & 3))) return false;
    float* outp = (float*)ep.out0;
    if constexpr (EPI == EPI_F32_SLAB) {  // split-K partial: slab z of the caller's slab buffer
      if (ep.slab_stride & 3) return false;
      outp += (size_t)ep.z * ep.slab_stride;
    }
    unsigned char* stg = smem + wave * 16384;  // [64 rows][256 B], 16 float4 chunks per row
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lr = i * 16 + (lane & 15), cq = j * 4 + (lane >> 4);  // float4 chunk index in the row
        const f32x4 v = acc[i][j];
        *reinterpret_cast<float4*>(stg + lr * 256 + ((cq ^ (lr & 15)) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
      }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int lr = rr * 4 + (lane >> 4), cq = lane & 15;
      const int row = wrow0 + lr, col = wcol0 + cq * 4;
      float4 v = *reinterpret_cast<const float4*>(stg + lr * 256 + ((cq ^ (lr & 15)) << 4));
      if (row < p.M && col < p.N) {
        if (EPI != EPI_F32_SLAB && ep.bias) {
          const float4 b = *reinterpret_cast<const float4*>(ep.bias + col);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if constexpr (EPI == EPI_F32_BIAS_POS) {
          const float4 e = *reinterpret_cast<const float4*>(ep.pos + (size_t)(row % ep.seq) * p.N + col);
          v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
        }
        if constexpr (EPI == EPI_F32_BIAS_RESID) {
          const float4 e = *reinterpret_cast<const float4*>(ep.resid + (size_t)row * ep.ldr + col);
          v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
        }
        *reinterpret_cast<float4*>(outp + (size_t)row * ep.ldo0 + col) = v;
      }
    }
    return true;
  } else {
    return false;
  }
}

DEVINL void zero_acc(f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

#ifndef GLDS_AUX
#define GLDS_AUX 0
#endif
DEVINL void glds16(const bf16_t* src, unsigned char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)lds_dst, 16, 0, GLDS_AUX);
}

// ------------------------------------------------------------------------------------------
// fast NT: A[m*lda + k], B[n*ldb + k]; requires K % 32 == 0
// ------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_fast_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const BlockCoord bc = block_coord(p);
  const int m0 = bc.m0, n0 = bc.n0;
  int kt_beg, kt_end;
  split_range(p, bc.z, kt_beg, kt_end);
  if (kt_beg >= kt_end) return;

  f32x4 acc[4][4];
  zero_acc(acc);

  // DMA piece i of this wave covers tile rows (wave*4+i)*8 .. +7; lane -> row lane>>3, LDS chunk
  // position lane&7, which must receive logical chunk (lane&7) ^ (row&7).
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  const bf16_t* arow[4];
  const bf16_t* brow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + lrow;
    arow[i] = p.A + (size_t)min(m0 + r, p.M - 1) * p.lda;
    brow[i] = p.B + (size_t)min(n0 + r, p.N - 1) * p.ldb;
  }
  auto stage = [&](int buf, int kt) {
    int k = kt * BK + lchunk * 8;
    if (k >= p.K) k -= 32;  // K-tail (K % 64 == 32): duplicate valid data, ks = 1 is skipped
    unsigned char* As = smem + buf * 2 * TILE_BYTES + wave * 4096;
    unsigned char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(arow[i] + k, As + i * 1024);
      glds16(brow[i] + k, Bs + i * 1024);
    }
  };

  stage(0, kt_beg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (int kt = kt_beg; kt < kt_end; ++kt) {
    if (kt + 1 < kt_end) stage(buf ^ 1, kt + 1);
    const unsigned char* As = smem + buf * 2 * TILE_BYTES;
    const int nks = (p.K - kt * BK >= BK) ? 2 : 1;
    compute_tile<false, kSwap<EPI>>(As, As + TILE_BYTES, acc, nks, wm, wn, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf ^= 1;
  }
  if (p.splitk == 1 && staged_epilogue_128<EPI>(p, acc, smem, m0, n0, wave, wm, wn, lane)) return;
  run_epilogue<EPI>(p, acc, m0, n0, wm, wn, lane, bc.z);
}

// (Round 6: the round-1 big-tile kernel gemm_nt_big_kernel - nt variants 6 / 7, big_impl = 0 - was removed; gemm_big.hip
// generalised it in round 2 and has been the only big-tile path of the engine since.  git history: round 5.)

// ------------------------------------------------------------------------------------------
// fast TN: A[k*lda + m], B[k*ldb + n]; requires K % 64 == 0
// ------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tn_fast_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const BlockCoord bc = block_coord(p);
  const int m0 = bc.m0, n0 = bc.n0;
  int kt_beg, kt_end;
  split_range(p, bc.z, kt_beg, kt_end);
  if (kt_beg >= kt_end) return;

  f32x4 acc[4][4];
  zero_acc(acc);

  // DMA piece i of this wave covers tile rows k = (wave*4+i)*4 .. +3; lane -> row lane>>4, LDS
  // chunk position p = lane&15 which must receive logical chunk ((p>>1) ^ f(k))<<1 | (p&1).
  const int lk = lane >> 4, lp = lane & 15;
  int acol[4], bcol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = (wave * 4 + i) * 4 + lk;
    const int mc = ((((lp >> 1) ^ tn_f(k)) << 1) | (lp & 1)) * 8;
    acol[i] = min(m0 + mc, p.lda - 8);
    bcol[i] = min(n0 + mc, p.ldb - 8);
  }
  auto stage = [&](int buf, int kt) {
    unsigned char* As = smem + buf * 2 * TILE_BYTES + wave * 4096;
    unsigned char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t k = (size_t)kt * BK + (wave * 4 + i) * 4 + lk;
      glds16(p.A + k * p.lda + acol[i], As + i * 1024);
      glds16(p.B + k * p.ldb + bcol[i], Bs + i * 1024);
    }
  };

  stage(0, kt_beg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (int kt = kt_beg; kt < kt_end; ++kt) {
    if (kt + 1 < kt_end) stage(buf ^ 1, kt + 1);
    const unsigned char* As = smem + buf * 2 * TILE_BYTES;
    compute_tile_tn<kSwap<EPI>>(As, As + TILE_BYTES, acc, wm, wn, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf ^= 1;
  }
  if constexpr (EPI == EPI_F32_SLAB) {
    GemmParams q = p;
    q.ep.z = bc.z;
    if (staged_epilogue_128<EPI>(q, acc, smem, m0, n0, wave, wm, wn, lane)) return;
  }
  run_epilogue<EPI>(p, acc, m0, n0, wm, wn, lane, bc.z);
}

// ------------------------------------------------------------------------------------------
// generic kernels: register-staged, zero-filled bounds (any M, N; K % 8 == 0 for NT)
// ------------------------------------------------------------------------------------------
template <int EPI, bool TN>
__global__ __launch_bounds__(256, 2) void gemm_generic_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const BlockCoord bc = block_coord(p);
  const int m0 = bc.m0, n0 = bc.n0;
  int kt_beg, kt_end;
  split_range(p, bc.z, kt_beg, kt_end);
  if (kt_beg >= kt_end) return;

  f32x4 acc[4][4];
  zero_acc(acc);
  bf16x8 ra[4], rb[4];
  const bf16x8 zero = zero_bf16x8();

  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256;
      if constexpr (!TN) {
        const int row = c >> 3, kc = c & 7;
        const int k = k0 + kc * 8;
        const int gm = m0 + row, gn = n0 + row;
        ra[i] = (gm < p.M && k < p.K) ? ld_global_bf16x8(p.A + (size_t)gm * p.lda + k) : zero;
        rb[i] = (gn < p.N && k < p.K) ? ld_global_bf16x8(p.B + (size_t)gn * p.ldb + k) : zero;
      } else {
        const int krow = c >> 4, mc = c & 15;
        const int k = k0 + krow;
        const int gm = m0 + mc * 8, gn = n0 + mc * 8;
        ra[i] = (k < p.K && gm < p.lda) ? ld_global_bf16x8(p.A + (size_t)k * p.lda + gm) : zero;
        rb[i] = (k < p.K && gn < p.ldb) ? ld_global_bf16x8(p.B + (size_t)k * p.ldb + gn) : zero;
      }
    }
  };
  auto sstore = [&](int buf) {
    unsigned char* As = smem + buf * 2 * TILE_BYTES;
    unsigned char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256;
      int off;
      if constexpr (!TN) {
        const int row = c >> 3, kc = c & 7;
        off = row * 128 + ((kc ^ (row & 7)) << 4);
      } else {
        const int krow = c >> 4, mc = c & 15;
        off = krow * 256 + (((((mc >> 1) ^ tn_f(krow)) << 1) | (mc & 1)) << 4);
      }
      *reinterpret_cast<bf16x8*>(As + off) = ra[i];
      *reinterpret_cast<bf16x8*>(Bs + off) = rb[i];
    }
  };

  gload(kt_beg);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int kt = kt_beg; kt < kt_end; ++kt) {
    const bool more = (kt + 1 < kt_end);
    if (more) gload(kt + 1);
    const unsigned char* As = smem + buf * 2 * TILE_BYTES;
    compute_tile<TN, kSwap<EPI>>(As, As + TILE_BYTES, acc, 2, wm, wn, lane);
    if (more) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  run_epilogue<EPI>(p, acc, m0, n0, wm, wn, lane, bc.z);
}

int g_nt_band = 8;  // tile band height of the 128x128 NT kernel (bench knob; measured: 8 >= 4 > row-major)
// 0 auto; 1 = 128x128 kernel;
// 10 / 11 / 12 = big-tile family of gemm_big.hip at 288x256 / 256x256 / 256x160 (bench/test knob)
int g_nt_variant = 0;
int g_big_impl = 1;  // auto mode: 1 = gemm_big.hip family incl. the 256x128 pairs, 2 = without them (A/B knob), 0 = 128x128 kernel only
int g_k64 = 3;  // auto mode: 64-deep ring slots (bit 0: the 256x160 tile, bit 1: 288x256 / 256x256); 0 = round-2 32-deep slots
int g_splitk_max = 4;  // in-kernel split-K of the 256x160 tile when the caller hands in a workspace (1 = off)
int g_tile192 = 0;     // auto mode knob (debug option tile192): whole-K N = 800 GEMMs without a split-K workspace on 192x160 tiles when
                       // M % 192 == 0.  Round 4: 8 % faster alone (42.8 -> 39.6 / 35.1 -> 32.0 us), -1..-2.5 us per launch in the step, step
                       // 7.605 vs 7.58 ms: 150 + 95 workgroups leave the row kernels no whole CU - off

template <int EPI>
constexpr bool kBigEpi = (EPI == EPI_BF16 || EPI == EPI_F32_BIAS || EPI == EPI_F32_BIAS_RESID ||
                          EPI == EPI_BIAS_GELU || EPI == EPI_GELU_BWD || EPI == EPI_HEADS || EPI == EPI_F32_BF16);

// tile band (in m-tiles) that makes the ~tiles/8 tiles one XCD owns a near-square patch in bytes
int band_for(int tiles, int tm, int bm, int bn) {
  const double per_xcd = tiles / 8.0;
  int band = (int)(std::sqrt(per_xcd * bn / bm) + 0.5);
  if (band < 1) band = 1;
  if (band > tm) band = tm;
  return band;
}

// short-K N = 800 GEMMs (out-projection forward / dgrad) on 128x160 instead of 256x128 tiles, two workgroups per CU either way:
// 45 x 5 = 225 workgroups instead of 23 x 7 = 161 (with 12.5 % column padding) at M = 5760, so the epilogue streams from 225
// CUs.  Round 5: stand-alone 21.2 -> 18.8 us (+ residual), 14.3 -> 13.1 us (bf16); in the step 26.0 -> 25.0 and 28.2 -> 25.6 us
// per launch, step 7.574 -> 7.525 ms (3 interleaved rounds, profiles/r05_ab_tile128x160.txt).  0 = the round-2 tile (A/B knob)
int g_tile128x160 = 1;  // 2 = also the LONG-K N = 800 GEMMs that fall through to 256x128 (the encoder stacks' whole-K dgrads: round-5 behaviour, A/B)

template <int EPI>
int launch_nt_t(const GemmParams& p_in, hipStream_t s) {
  GemmParams p = p_in;
  const int user_band = p.band;
  if (p.band == 0) p.band = g_nt_band;
  const int tn = (p.N + BN - 1) / BN;
  const int tiles128 = tn * ((p.M + 127) / 128);
  if (p.force_generic || (p.K & 31)) {
    FACT_LAUNCH((gemm_generic_kernel<EPI, false>), dim3(tiles128 * p.splitk), dim3(256), 0, s, p);
    return 0;
  }
  int variant = g_nt_variant;
  if (p.splitk > 1) variant = 1;
  if constexpr (kBigEpi<EPI>) {
    const bool ws = p.sk_slab && p.sk_cnt && g_splitk_max > 1 && p.splitk == 1;
    int cfg = -1;
    if (variant >= 10 && variant <= 12) cfg = variant - 10;
    else if (variant == 14) cfg = BIG_256x128;
    else if (variant == 17)  // 64-deep slots need whole lines in memory: row pitch >= K rounded up to 64
      cfg = (p.lda >= ((p.K + 63) & ~63) && p.ldb >= ((p.K + 63) & ~63)) ? BIG_256x160_K64 : BIG_256x160;
    else if (variant == 0) {
      // Big tiles (one 8-wave workgroup per CU) are taken when their rounds of 256 workgroups are well
      // filled, counting tile padding: useful outputs / (rounds * 256 * tile area) >= 0.6 - e.g. M = 5760,
      // N = 3072 -> 20 x 12 = 240 tiles of 288 x 256 (0.94).  N = 800 outputs (5 x 160 columns exactly)
      // take the 256x160 tile from 100 tiles up: a round of 115 leaves CUs to the weight-gradient stream.
      const int tnb = (p.N + 255) / 256;
      const int t9 = tnb * ((p.M + 287) / 288), t8 = tnb * ((p.M + 255) / 256);
      const double useful = (double)p.M * p.N;
      const double e9 = useful / ((double)((t9 + 255) / 256) * 256 * 288 * 256);
      const double e8 = useful / ((double)((t8 + 255) / 256) * 256 * 256 * 256);
      const int t160 = ((p.N + 159) / 160) * ((p.M + 255) / 256);
      const int t128 = ((p.N + 127) / 128) * ((p.M + 255) / 256);
      const bool n800 = (p.N % 160 == 0 && p.N <= 960);
      if ((e9 >= 0.6 || e8 >= 0.6) && g_big_impl) {
        cfg = (e8 > e9) ? BIG_256x256 : BIG_288x256;
      } else if (g_big_impl && n800 && p.K >= 1536 && (t160 >= 100 || (ws && t160 >= 32))) {
        cfg = BIG_256x160;  // long K: the in-kernel split-K (with a workspace) fills the chip
      } else if (g_big_impl && n800 && t160 > 160 && t160 <= 256) {
        // short K, but one well-filled round of 256x160 tiles (the AR sampler at 32 sequences: M = 11520 -> 225 tiles):
        // out-proj + residual 28.0 vs 34.2 us on the 256x128 pairs (tools/bench_r2.py ar32)
        cfg = BIG_256x160;
      } else if (g_big_impl == 1 && t128 >= 48 && t128 <= 512) {  // g_big_impl == 2: A/B without this config
        // short-K N = 800 GEMMs and the wide GEMMs of a short token count (motion encoder): 256x128 tiles, two
        // workgroups per CU (round-2 bench: N800 K800 20.1 vs 22.4 us, M1920 N3072 20.2 vs 23.1 us)
        cfg = BIG_256x128;
      }
    }
    const bool k64_ok = p.lda >= ((p.K + 63) & ~63) && p.ldb >= ((p.K + 63) & ~63);
    if (variant == 18) cfg = k64_ok ? BIG_288x256_K64 : BIG_288x256;
    if (variant == 19) cfg = k64_ok ? BIG_256x256_K64 : BIG_256x256;
    if (variant == 20) cfg = BIG_192x160_K64;
    if (variant == 21) cfg = BIG_128x160;
    if (variant == 22) cfg = k64_ok ? BIG_256x256_M32 : BIG_256x256;   // 32 x 32 x 16 MFMA tiles (round 6)
    if (variant == 23) cfg = k64_ok ? BIG_384x192_M32 : BIG_288x256;
    if (variant == 0 && g_tile128x160 && cfg == BIG_256x128 && p.N % 160 == 0 && p.N <= 960 && (p.K < 1536 || g_tile128x160 == 2) &&
        (EPI == EPI_F32_BIAS_RESID || EPI == EPI_HEADS || EPI == EPI_BF16))
      cfg = BIG_128x160;  // the SHORT-K N = 800 GEMMs on 128x160 tiles (long K: the tile loses, profiles/r05_tile128x160_long_k.txt)
    if (g_k64 && variant == 0 && k64_ok) {
      if (cfg == BIG_256x160 && EPI == EPI_BF16 && g_tile192 && !ws && p.M % 192 == 0 && (p.M / 192) * ((p.N + 159) / 160) <= 256)
        cfg = BIG_192x160_K64;  // the whole-K dgrads of the backward chain: more, smaller tiles in the one round they get
      else if (cfg == BIG_256x160) cfg = BIG_256x160_K64;
      else if ((g_k64 & 2) && cfg == BIG_288x256) cfg = BIG_288x256_K64;
      else if ((g_k64 & 2) && cfg == BIG_256x256) cfg = BIG_256x256_K64;
    }
    if ((cfg == BIG_256x160 || cfg == BIG_256x160_K64) && ws) {
      // N = 800 outputs give only 5 column tiles: M = 5760 -> 115 tiles on 256 CUs.  With a workspace each
      // tile is cut along K into 2-4 slices (<= 256 workgroups) that finish in-kernel.
      const int t160 = ((p.N + 159) / 160) * ((p.M + 255) / 256);
      int sk = 256 / t160;
      if (sk > g_splitk_max) sk = g_splitk_max;
      // the finish moves 2 x the fp32 tile through L2/fabric (~10 us at 230 workgroups, round-2 bench): it
      // pays from ~24 stages (768 k) per slice - K = 3072 / 2400 yes, K = 800 no
      while (sk > 1 && (p.K / 32) / sk < 24) --sk;
      if (sk > 1) p.splitk = sk;
    }
    if (cfg >= 0 && (EPI != EPI_HEADS || (p.M < 65536 && p.N < 65536))) {
      int bm = 0, bn = 0;
      big_tile_dims(cfg, &bm, &bn);
      const int tm = (p.M + bm - 1) / bm, tnn = (p.N + bn - 1) / bn;
      p.band = user_band > 0 ? user_band : band_for(tm * tnn, tm, bm, bn);
      return launch_big_nt(cfg, EPI, p, s);
    }
  }
  FACT_LAUNCH(gemm_nt_fast_kernel<EPI>, dim3(tiles128 * p.splitk), dim3(256), 0, s, p);
  return 0;
}
template <int EPI>
int launch_tn_t(const GemmParams& p, hipStream_t s) {
  const int tn = (p.N + BN - 1) / BN;
  dim3 grid(tn * ((p.M + BM - 1) / BM) * p.splitk);
  if ((p.K & 63) == 0 && p.lda >= 8 && p.ldb >= 8 && !p.force_generic) {
    FACT_LAUNCH(gemm_tn_fast_kernel<EPI>, grid, dim3(256), 0, s, p);
  } else {
    FACT_LAUNCH((gemm_generic_kernel<EPI, true>), grid, dim3(256), 0, s, p);
  }
  return 0;
}

int check_common(const GemmParams& p, int epi) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return -1;
  if ((p.lda & 7) || (p.ldb & 7)) return -2;
  if (p.splitk < 1) return -3;
  if (p.splitk > 1 && epi != EPI_ATOMIC_F32 && epi != EPI_F32_SLAB) return -4;
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) return -5;
  if (epi == EPI_HEADS && ((p.ep.dh & 3) || (p.ep.dhp & 3))) return -8;
  return 0;
}

}  // namespace

void gemm_set_nt_variant(int v) { g_nt_variant = v; }
void gemm_set_tile192(int v) { g_tile192 = v; }
void gemm_set_big_impl(int v) { g_big_impl = v; }
void gemm_set_k64(int v) { g_k64 = v; }
void gemm_set_splitk_max(int v) { g_splitk_max = v < 1 ? 1 : (v > 4 ? 4 : v); }
void gemm_set_tile128x160(int v) { g_tile128x160 = v; }
void gemm_set_nt_band(int band) { g_nt_band = band; }

// K split of the skinny-M path, 0 = the GEMM does not take it
static int skinny_split(int epi, const GemmParams& p) {
  if (!(p.skinny_acc && p.M <= 512 && p.splitk == 1 && !p.force_generic && !(p.K & 31) && !(p.K & 7) && g_nt_variant == 0 &&
        (epi == EPI_BF16 || epi == EPI_F32_BIAS || epi == EPI_F32_BIAS_RESID || epi == EPI_BIAS_GELU ||
         epi == EPI_GELU_BWD || epi == EPI_HEADS || epi == EPI_F32_BF16)))
    return 0;
  const int ldacc = (p.N + 3) & ~3;
  const int tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128), ktiles = (p.K + 63) / 64;
  // ~136 workgroups: every extra K slice adds a 128x128 tile of fp32 atomics (tools/skinny_bench.py at M = 360: QKV 11.4 us
  // at 2 slices / 13.2 at 4, FFN1 12.8 at 2 / 14.2 at 3, FFN2 13.0 at 6 / 16.3 at 12)
  int sk = (136 + tiles / 2) / tiles;
  if (sk > ktiles / 2) sk = ktiles / 2;
  return (sk >= 2 && (size_t)p.M * ldacc <= p.skinny_floats) ? sk : 0;
}
bool gemm_nt_takes_skinny(int epi, const GemmParams& p) { return check_common(p, epi) == 0 && skinny_split(epi, p) > 0; }

int launch_gemm_nt(int epi, const GemmParams& p, hipStream_t s) {
  int rc = check_common(p, epi);
  if (rc) return rc;
  if (p.K & 7) return -6;
  // Skinny-M path (GemmParams::skinny_acc): few row tiles -> cut K so that ~256 workgroups share the GEMM
  if (const int sk = skinny_split(epi, p)) {
    const int ldacc = (p.N + 3) & ~3;
    {
      GemmParams q = p;
      q.splitk = sk;
      q.skinny_acc = nullptr;
      q.ep.out0 = p.skinny_acc;
      q.ep.ldo0 = ldacc;
      q.ep.alpha = 1.0f;
      rc = launch_nt_t<EPI_ATOMIC_F32>(q, s);
      if (rc) return rc;
      GemmParams e = p;
      if (epi == EPI_HEADS) {
        if (p.M >= 65536 || p.N >= 65536) return -8;
      }
      return launch_skinny_epilogue(epi, p.skinny_acc, ldacc, e, s);
    }
  }
  switch (epi) {
    case EPI_BF16: return launch_nt_t<EPI_BF16>(p, s);
    case EPI_F32_BIAS: return launch_nt_t<EPI_F32_BIAS>(p, s);
    case EPI_F32_BIAS_POS: return launch_nt_t<EPI_F32_BIAS_POS>(p, s);
    case EPI_F32_BIAS_RESID: return launch_nt_t<EPI_F32_BIAS_RESID>(p, s);
    case EPI_BIAS_GELU: return launch_nt_t<EPI_BIAS_GELU>(p, s);
    case EPI_GELU_BWD: return launch_nt_t<EPI_GELU_BWD>(p, s);
    case EPI_HEADS: return launch_nt_t<EPI_HEADS>(p, s);
    case EPI_ATOMIC_F32: return launch_nt_t<EPI_ATOMIC_F32>(p, s);
    case EPI_F32_BF16: return launch_nt_t<EPI_F32_BF16>(p, s);
  }
  return -7;
}

int launch_gemm_tn(int epi, const GemmParams& p, hipStream_t s) {
  int rc = check_common(p, epi);
  if (rc) return rc;
  switch (epi) {
    case EPI_BF16: return launch_tn_t<EPI_BF16>(p, s);
    case EPI_ATOMIC_F32: return launch_tn_t<EPI_ATOMIC_F32>(p, s);
    case EPI_F32_SLAB: return launch_tn_t<EPI_F32_SLAB>(p, s);
  }
  return -7;
}
