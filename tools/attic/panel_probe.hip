// Feasibility probe (round 3): "row-panel" NT GEMM for the N = 800 outputs of FACT - one workgroup owns BM FULL rows
// (all 800 columns) so that an epilogue can do per-row work (LayerNorm forward / backward) without a second kernel.
//   C[M][800] = A[M][K] * B[800][K]^T, bf16 in, bf16 out.   10 waves: wave w owns columns 80w .. 80w+79.
// Every wave streams its OWN 80 weight rows through a private slice of a 3-slot LDS ring (LDS-DMA), the BM activation
// rows of a K step are shared (2-4 DMA pieces), one s_barrier per 32-deep K step.  The question the probe answers: how
// close to the 64 B/clk/CU L1 fill rate does a CU stream the weight matrix while it multiplies (the kernel is bound by
// it: K * 25 cycles per workgroup whatever BM is)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/panel_probe tools/panel_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define DEVINL __device__ __forceinline__
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

DEVINL void glds16(const void* src, unsigned char* dst) {
  __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)dst, 16, 0, 0);
}
template <int N>
DEVINL void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
DEVINL int ring_g(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

constexpr int NW = 10, NCOL = 800, WCOLS = 80, NR = 5;  // 5 column tiles of 16 per wave

template <int BM>
struct Cfg {
  static constexpr int MR = BM / 16;                 // row tiles (shared by all waves)
  static constexpr int A_PIECES = BM / 16;           // 1 KiB pieces of the activation rows per stage
  static constexpr int STAGE = (A_PIECES + NCOL / 16) * 1024;
  static constexpr int NSLOT = 3;
  static constexpr int LDS = NSLOT * STAGE;
  static_assert(LDS <= 160 * 1024, "LDS");
};

template <int BM, int MODE = 0, bool LINE = false>
__global__ __launch_bounds__(NW * 64) void panel_nt_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B,
                                                            int ldb, bf16_t* __restrict__ C, int ldc, int M, int K) {
  using G = Cfg<BM>;
  constexpr int MR = G::MR, STAGE = G::STAGE;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM;
  const int nk = K / 32;
  // DMA sources: this wave's 5 weight pieces (rows 80w + 16p + lrow) and, for waves < A_PIECES, one activation piece
  // LINE: whole-line pieces (8 rows x 128 B, DMA-only modes): the wave's 80 weight rows x 64 k = 10 pieces per 64-deep stage,
  // issued 5 per 32-deep step (pieces 0-4 on even steps, 5-9 on odd ones): same bytes per step as the half-line form
  const int lrow = LINE ? (lane >> 3) : (lane >> 2), lchunk = LINE ? ((lane & 7) ^ lrow) : ((lane & 3) ^ ring_g(lrow));
  const char* bsrc[NR];
  const char* bsrc2[NR];
#pragma unroll
  for (int p = 0; p < NR; ++p) {
    bsrc2[p] = reinterpret_cast<const char*>(B) + ((size_t)(wave * WCOLS + (p + NR) * 8 + lrow) * ldb + lchunk * 8) * 2;
    bsrc[p] = reinterpret_cast<const char*>(B) + ((size_t)(wave * WCOLS + p * (LINE ? 8 : 16) + lrow) * ldb + lchunk * 8) * 2 +
              (MODE == 3 ? (size_t)blockIdx.x * NCOL * ldb * 2 : MODE == 4 ? (size_t)(blockIdx.x >> 3) * NCOL * ldb * 2 : MODE == 5 ? (size_t)(blockIdx.x & 7) * NCOL * ldb * 2 : 0);
  }
  const bool has_a = wave < G::A_PIECES;
  const char* asrc = reinterpret_cast<const char*>(A) + ((size_t)min(m0 + (has_a ? wave : 0) * 16 + lrow, M - 1) * lda + lchunk * 8) * 2;
  const int a_dst = (has_a ? wave : 0) * 1024, b_dst = G::A_PIECES * 1024 + wave * NR * 1024;

  auto issue = [&](int slot, int kt) {
    unsigned char* base = smem + slot * STAGE;
    const size_t koff = LINE ? (size_t)(min(kt, nk - 1) >> 1) * 128 : (size_t)min(kt, nk - 1) * 64;
#pragma unroll
    for (int p = 0; p < NR; ++p) glds16((LINE && (kt & 1) ? bsrc2[p] : bsrc[p]) + koff, base + b_dst + p * 1024);
    if (has_a) glds16(asrc + koff, base + a_dst);
  };
  f32x4 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int r = lane & 15, chunk = lane >> 4;
  const unsigned lanepart = (unsigned)(r * 64 + ((chunk ^ ring_g(r)) << 4));

  issue(0, 0);
  issue(1, 1);
  auto body = [&](auto slot_c, int kt) {
    constexpr int SLOT = decltype(slot_c)::value;
    // stage kt landed (this wave's pieces): one later stage may stay in flight
    if (has_a) wait_vm<NR + 1>(); else wait_vm<NR>();
    if (MODE < 2) __builtin_amdgcn_s_barrier();
    issue((SLOT + 2) % 3, kt + 2);
    if (MODE != 0) return;
    const unsigned char* st = smem + SLOT * STAGE;
    bf16x8 af[MR], bfr[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(st + b_dst + j * 1024 + lanepart);
#pragma unroll
    for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + i * 1024 + lanepart);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  for (int kt = 0; kt < nk; kt += 3) {
    body(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) body(std::integral_constant<int, 1>{}, kt + 1);
    if (kt + 2 < nk) body(std::integral_constant<int, 2>{}, kt + 2);
  }
  wait_vm<0>();
  // swapped roles: lane holds C[m = 16i + (lane&15)][n = 80w + 16j + 4(lane>>4) + 0..3]
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    const int m = m0 + i * 16 + (lane & 15);
    if (m < M) {
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int n = wave * WCOLS + j * 16 + (lane >> 4) * 4;
        bf16x4 o = {(bf16_t)acc[i][j][0], (bf16_t)acc[i][j][1], (bf16_t)acc[i][j][2], (bf16_t)acc[i][j][3]};
        *reinterpret_cast<bf16x4*>(C + (size_t)m * ldc + n) = o;
      }
    }
  }
}

static float bf2f(bf16_t v) { return (float)v; }

template <int BM, int MODE = 0, bool LINE = false>
void run(int M, int K, const bf16_t* dA, const bf16_t* dB, bf16_t* dC, const std::vector<bf16_t>& hA, const std::vector<bf16_t>& hB,
         int lda, int ldb) {
  using G = Cfg<BM>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(panel_nt_kernel<BM, MODE, LINE>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
  const int grid = (M + BM - 1) / BM;
  hipMemset(dC, 0, (size_t)M * 832 * 2);
  hipLaunchKernelGGL((panel_nt_kernel<BM, MODE, LINE>), dim3(grid), dim3(NW * 64), G::LDS, 0, dA, lda, dB, ldb, dC, 832, M, K);
  hipDeviceSynchronize();
  std::vector<bf16_t> hC((size_t)M * 832);
  hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int t = 0; t < 400; ++t) {
    const int m = (t * 7919 + (t % 3) * (M - 1)) % M, n = (t * 104729) % NCOL;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * lda + k]) * bf2f(hB[(size_t)n * ldb + k]);
    maxerr = fmax(maxerr, fabs(ref - bf2f(hC[(size_t)m * 832 + n])));
    maxref = fmax(maxref, fabs(ref));
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((panel_nt_kernel<BM, MODE, LINE>), dim3(grid), dim3(NW * 64), G::LDS, 0, dA, lda, dB, ldb, dC, 832, M, K);
  hipEventRecord(e0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((panel_nt_kernel<BM, MODE, LINE>), dim3(grid), dim3(NW * 64), G::LDS, 0, dA, lda, dB, ldb, dC, 832, M, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / iters;
  printf("%s mode %d panel BM %2d  M %d K %4d: %3d workgroups  %7.1f us  %5.0f TFLOP/s  per K step %.0f cycles@2.4GHz  max err %.3g (ref max %.3g)\n", LINE ? "whole-line" : "half-line ", MODE, BM, M, K,
         grid, us, 2.0 * M * NCOL * K / us / 1e6, us * 2400.0 / (K / 32), maxerr, maxref);
}

int main() {
  const int M = 5760;
  for (int K : {800, 3072}) {
    const int lda = (K + 63) / 64 * 64, ldb = lda;
    std::vector<bf16_t> hA((size_t)M * lda), hB((size_t)NCOL * ldb);
    const size_t bcopies = 181;
    srand(1);
    for (auto& v : hA) v = (bf16_t)((rand() % 2001 - 1000) / 1000.0f);
    for (auto& v : hB) v = (bf16_t)((rand() % 2001 - 1000) / 20000.0f);
    bf16_t *dA, *dB, *dC;
    hipMalloc((void**)&dA, hA.size() * 2); hipMalloc((void**)&dB, hB.size() * 2 * bcopies); hipMalloc((void**)&dC, (size_t)M * 832 * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    for (size_t c = 0; c < bcopies; ++c) hipMemcpy((char*)dB + c * hB.size() * 2, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    run<32>(M, K, dA, dB, dC, hA, hB, lda, ldb);
    run<48>(M, K, dA, dB, dC, hA, hB, lda, ldb);
    run<32, 1>(M, K, dA, dB, dC, hA, hB, lda, ldb);
    run<32, 2>(M, K, dA, dB, dC, hA, hB, lda, ldb);
    run<32, 2, true>(M, K, dA, dB, dC, hA, hB, lda, ldb);
    run<32, 5, true>(M, K, dA, dB, dC, hA, hB, lda, ldb);
    run<32, 3>(M, K, dA, dB, dC, hA, hB, lda, ldb);   // DMA only, every workgroup its own copy of B
    run<32, 4>(M, K, dA, dB, dC, hA, hB, lda, ldb);   // ... one copy per 8 consecutive workgroups (= one per XCD slot round)
    run<32, 5>(M, K, dA, dB, dC, hA, hB, lda, ldb);   // ... one copy per XCD (blockIdx & 7)

    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
