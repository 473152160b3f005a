// bf16 MFMA GEMM engine (see gemm.h).  128x128x64 block tile, 256 threads = 4 waves (2x2),
// each wave a 64x64 sub-tile = 4x4 MFMA 16x16x32 tiles.  Global->register->LDS staging with the
// next tile's loads in flight during the current tile's MFMAs; double-buffered LDS, one barrier
// per K step.  LDS images are XOR-swizzled so ds_read_b128 / ds_read_b64_tr_b16 fragment reads
// are bank-conflict free.
#include "gemm.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <int EPI>
DEVINL void epilogue_store(const EpiParams& ep, int M, int N, int row0, int col, f32x4 v) {
  if (col >= N || row0 >= M) return;
  if constexpr (EPI == EPI_BF16) {
    bf16_t* o = (bf16_t*)ep.out0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (row0 + r < M) o[(size_t)(row0 + r) * ep.ldo0 + col] = (bf16_t)v[r];
  } else if constexpr (EPI == EPI_F32_BIAS) {
    float* o = (float*)ep.out0;
    const float b = ep.bias ? ep.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (row0 + r < M) o[(size_t)(row0 + r) * ep.ldo0 + col] = v[r] + b;
  } else if constexpr (EPI == EPI_F32_BIAS_POS) {
    float* o = (float*)ep.out0;
    const float b = ep.bias[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + r;
      if (row < M) o[(size_t)row * ep.ldo0 + col] = v[r] + b + ep.pos[(size_t)(row % ep.seq) * N + col];
    }
  } else if constexpr (EPI == EPI_F32_BIAS_RESID) {
    float* o = (float*)ep.out0;
    const float b = ep.bias[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + r;
      if (row < M) o[(size_t)row * ep.ldo0 + col] = v[r] + b + ep.resid[(size_t)row * ep.ldr + col];
    }
  } else if constexpr (EPI == EPI_BIAS_GELU) {
    bf16_t* o0 = (bf16_t*)ep.out0;
    bf16_t* o1 = (bf16_t*)ep.out1;
    const float b = ep.bias[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + r;
      if (row < M) {
        const float pre = v[r] + b;
        o0[(size_t)row * ep.ldo0 + col] = (bf16_t)pre;
        o1[(size_t)row * ep.ldo1 + col] = (bf16_t)gelu_tanh(pre);
      }
    }
  } else if constexpr (EPI == EPI_GELU_BWD) {
    bf16_t* o = (bf16_t*)ep.out0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + r;
      if (row < M) {
        const float pre = (float)ep.pre[(size_t)row * ep.ldp + col];
        o[(size_t)row * ep.ldo0 + col] = (bf16_t)(v[r] * gelu_tanh_grad(pre));
      }
    }
  } else if constexpr (EPI == EPI_HEADS) {
    const int which = col / ep.hid;
    const int rem = col - which * ep.hid;
    const int h = rem / ep.dh;
    const int d = rem - h * ep.dh;
    bf16_t* hr = ep.hrow[which];
    bf16_t* ht = ep.htr[which];
    const int b0 = row0 / ep.n_tok;
    const int t0 = row0 - b0 * ep.n_tok;
    const bool vec = ((ep.n_tok & 3) == 0) && (row0 + 3 < M);
    if (hr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + r;
        if (row < M) {
          int b = b0, t = t0 + r;
          if (!vec) { b = row / ep.n_tok; t = row - b * ep.n_tok; }
          hr[((size_t)(b * ep.heads + h) * ep.n_pad + t) * ep.dhp + d] = (bf16_t)v[r];
        }
      }
    }
    if (ht) {
      if (vec) {
        bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(ht + ((size_t)(b0 * ep.heads + h) * ep.dh + d) * ep.n_pad + t0) = pk;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + r;
          if (row < M) {
            const int b = row / ep.n_tok, t = row - b * ep.n_tok;
            ht[((size_t)(b * ep.heads + h) * ep.dh + d) * ep.n_pad + t] = (bf16_t)v[r];
          }
        }
      }
    }
  } else if constexpr (EPI == EPI_F32_BF16) {
    float* o0 = (float*)ep.out0;
    bf16_t* o1 = (bf16_t*)ep.out1;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (row0 + r < M) {
        o0[(size_t)(row0 + r) * ep.ldo0 + col] = v[r];
        o1[(size_t)(row0 + r) * ep.ldo1 + col] = (bf16_t)v[r];
      }
  } else if constexpr (EPI == EPI_ATOMIC_F32) {
    float* o = (float*)ep.out0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (row0 + r < M) atomicAdd(o + (size_t)(row0 + r) * ep.ldo0 + col, v[r] * ep.alpha);
  }
}

// ------------------------------------------------------------------------------------------
// NT: A[m*lda + k], B[n*ldb + k]
// LDS image of a [128 rows][64 k] tile: 128-byte rows, 16-byte chunk c of row r stored at
// r*128 + ((c ^ (r & 7)) << 4): the 16 lanes of a ds_read_b128 service group land on 16 distinct
// 16-byte bank slots.
// ------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  const int ktiles = (p.K + BK - 1) / BK;
  const int per = (ktiles + p.splitk - 1) / p.splitk;
  const int kt_beg = blockIdx.z * per;
  const int kt_end = min(ktiles, kt_beg + per);
  if (kt_beg >= kt_end) return;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 ra[4], rb[4];
  const bf16x8 zero = zero_bf16x8();

  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256;
      const int row = c >> 3, kc = c & 7;
      const int k = k0 + kc * 8;
      const int gm = m0 + row, gn = n0 + row;
      ra[i] = (gm < p.M && k < p.K) ? ld_global_bf16x8(p.A + (size_t)gm * p.lda + k) : zero;
      rb[i] = (gn < p.N && k < p.K) ? ld_global_bf16x8(p.B + (size_t)gn * p.ldb + k) : zero;
    }
  };
  auto sstore = [&](int buf) {
    unsigned char* As = smem + buf * 2 * TILE_BYTES;
    unsigned char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256;
      const int row = c >> 3, kc = c & 7;
      const int off = row * 128 + ((kc ^ (row & 7)) << 4);
      *reinterpret_cast<bf16x8*>(As + off) = ra[i];
      *reinterpret_cast<bf16x8*>(Bs + off) = rb[i];
    }
  };
  auto compute = [&](int buf) {
    const unsigned char* As = smem + buf * 2 * TILE_BYTES;
    const unsigned char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
      const int chunk = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + (lane & 15);
        af[i] = *reinterpret_cast<const bf16x8*>(As + row * 128 + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + (lane & 15);
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * 128 + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
    }
  };

  gload(kt_beg);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int kt = kt_beg; kt < kt_end; ++kt) {
    const bool more = (kt + 1 < kt_end);
    if (more) gload(kt + 1);
    compute(buf);
    if (more) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row0 = m0 + wm * 64 + i * 16 + (lane >> 4) * 4;
      const int col = n0 + wn * 64 + j * 16 + (lane & 15);
      epilogue_store<EPI>(p.ep, p.M, p.N, row0, col, acc[i][j]);
    }
}

// ------------------------------------------------------------------------------------------
// TN: A[k*lda + m], B[k*ldb + n]  (wgrad: contraction over the token rows of both operands)
// LDS image of a [64 k][128 m] tile: 256-byte rows; 32-byte unit u (16 elements) of row k is
// stored at unit u ^ f(k), f(k) = (k&3) | ((k>>3)&1)<<2, so the 8 row-segments touched by one
// 32-lane ds_read_b64_tr_b16 service group fall in 8 distinct 32-byte bank slots.
// Fragment build: for MFMA k-slot group g = lane>>4, lane s = lane&15 supplies the address of
// 4 contiguous m-elements of row k = g*8 + hh*4 + (s>>2), columns (s&3)*4..+3; the hardware
// returns to lane c the 4 k-values of column c (ck_tile LaneGroupTransposeTraits: 4x16 -> 16x4).
// ------------------------------------------------------------------------------------------
DEVINL int tn_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  const int ktiles = (p.K + BK - 1) / BK;
  const int per = (ktiles + p.splitk - 1) / p.splitk;
  const int kt_beg = blockIdx.z * per;
  const int kt_end = min(ktiles, kt_beg + per);
  if (kt_beg >= kt_end) return;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 ra[4], rb[4];
  const bf16x8 zero = zero_bf16x8();

  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256;
      const int krow = c >> 4, mc = c & 15;
      const int k = k0 + krow;
      const int gm = m0 + mc * 8, gn = n0 + mc * 8;
      ra[i] = (k < p.K && gm < p.lda) ? ld_global_bf16x8(p.A + (size_t)k * p.lda + gm) : zero;
      rb[i] = (k < p.K && gn < p.ldb) ? ld_global_bf16x8(p.B + (size_t)k * p.ldb + gn) : zero;
    }
  };
  auto sstore = [&](int buf) {
    unsigned char* As = smem + buf * 2 * TILE_BYTES;
    unsigned char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256;
      const int krow = c >> 4, mc = c & 15;
      const int off = krow * 256 + (((((mc >> 1) ^ tn_f(krow)) << 1) | (mc & 1)) << 4);
      *reinterpret_cast<bf16x8*>(As + off) = ra[i];
      *reinterpret_cast<bf16x8*>(Bs + off) = rb[i];
    }
  };
  auto tr_frag = [&](const unsigned char* tile, int ks, int u) -> bf16x8 {
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
  };
  auto compute = [&](int buf) {
    const unsigned char* As = smem + buf * 2 * TILE_BYTES;
    const unsigned char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tr_frag(As, ks, wm * 4 + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = tr_frag(Bs, ks, wn * 4 + j);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
    }
  };

  gload(kt_beg);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int kt = kt_beg; kt < kt_end; ++kt) {
    const bool more = (kt + 1 < kt_end);
    if (more) gload(kt + 1);
    compute(buf);
    if (more) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row0 = m0 + wm * 64 + i * 16 + (lane >> 4) * 4;
      const int col = n0 + wn * 64 + j * 16 + (lane & 15);
      epilogue_store<EPI>(p.ep, p.M, p.N, row0, col, acc[i][j]);
    }
}

template <int EPI>
int launch_nt_t(const GemmParams& p, hipStream_t s) {
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.splitk);
  hipLaunchKernelGGL(gemm_nt_kernel<EPI>, grid, dim3(256), 0, s, p);
  return 0;
}
template <int EPI>
int launch_tn_t(const GemmParams& p, hipStream_t s) {
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.splitk);
  hipLaunchKernelGGL(gemm_tn_kernel<EPI>, grid, dim3(256), 0, s, p);
  return 0;
}

int check_common(const GemmParams& p, int epi) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return -1;
  if ((p.lda & 7) || (p.ldb & 7)) return -2;
  if (p.splitk < 1) return -3;
  if (p.splitk > 1 && epi != EPI_ATOMIC_F32) return -4;
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) return -5;
  return 0;
}

}  // namespace

int launch_gemm_nt(int epi, const GemmParams& p, hipStream_t s) {
  int rc = check_common(p, epi);
  if (rc) return rc;
  if (p.K & 7) return -6;
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
  }
  return -7;
}
