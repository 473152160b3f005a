// Layout probes (diagnostics only): run one MFMA / one LDS transpose-read with caller-chosen
// per-lane register contents so the host can verify the fragment layouts assumed in common.h.
#include "../../include/fact_hip_debug.h"
#include "common.h"

namespace {
__global__ void probe_mfma_kernel(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  bf16x8 av, bv;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    av[j] = (bf16_t)a[l * 8 + j];
    bv[j] = (bf16_t)b[l * 8 + j];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = mfma16(av, bv, acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__global__ void probe_tr_kernel(const float* vals, int n, const int* addrs, float* out) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[4096];
  for (int i = threadIdx.x; i < n && i < 4096; i += 64) lds[i] = (bf16_t)vals[i];
  __syncthreads();
  const int l = threadIdx.x;
  bf16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)((unsigned char*)lds + addrs[l]));
#pragma unroll
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)r[j];
}
}  // namespace

extern "C" {
int fact_probe_mfma(const float* a_regs, const float* b_regs, float* d_regs, void* stream) {
  FACT_LAUNCH(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a_regs, b_regs, d_regs);
  return 0;
}
int fact_probe_tr(const float* lds_vals, int n, const int* byte_addrs, float* out, void* stream) {
  FACT_LAUNCH(probe_tr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lds_vals, n, byte_addrs, out);
  return 0;
}
}
