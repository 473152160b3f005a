// Common device helpers for the gfx950 (CDNA4 / MI355X) FACT kernels.
// Wave = 64 lanes. All matrix math uses v_mfma_f32_16x16x32_bf16:
//   A fragment: lane l holds A[i = l&15][k = (l>>4)*8 + j], j = 0..7   (8 bf16 = 4 VGPR)
//   B fragment: lane l holds B[k = (l>>4)*8 + j][n = l&15]
//   C/D       : lane l holds D[row = (l>>4)*4 + r][col = l&15], r = 0..3
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define DEVINL __device__ __forceinline__

DEVINL f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// v_mfma_f32_32x32x16_bf16 (round 6, the N = 3072 / 2400 NT GEMMs):
//   A fragment: lane l holds A[i = l&31][k = (l>>5)*8 + j], j = 0..7;  B fragment: B[k = (l>>5)*8 + j][n = l&31]
//   C/D       : lane l holds D[row = 8*(r>>2) + 4*(l>>5) + (r&3)][col = l&31], r = 0..15
DEVINL f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

DEVINL bf16x8 zero_bf16x8() {
  u32x4 z = {0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8, z);
}

DEVINL bf16x8 ld_global_bf16x8(const bf16_t* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}

DEVINL float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

DEVINL float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// tanh-approximation GELU, the reference's form (mint/core/base_model_util.py:94-107):
//   0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))
// tanh(u) = 1 - 2/(1 + e^(2u)) with the hardware exp2 / rcp (v_exp_f32, v_rcp_f32, ~1 ulp):
// saturates cleanly to +-1 for large |u| (exp2 -> inf -> rcp -> 0, exp2 -> 0 -> 1 - 2).
DEVINL float fast_tanh(float u) {
  const float e = __builtin_amdgcn_exp2f(u * 2.8853900817779268f);  // e^(2u)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

DEVINL float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float t = fast_tanh(k0 * (x + k1 * x * x * x));
  return 0.5f * x * (1.0f + t);
}

DEVINL float gelu_tanh_grad(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float x2 = x * x;
  const float t = fast_tanh(k0 * (x + k1 * x * x2));
  const float du = k0 * (1.0f + 3.0f * k1 * x2);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}

// Keras Adam, one element (optimizer_v2/adam.py _resource_apply_dense, non-amsgrad): lr_t carries both bias corrections,
// epsilon is added OUTSIDE them.  One definition for the optimizer pass (rowops.hip) and the fused epilogue of the grouped
// wgrad kernel (gemm_big.hip), so the two paths round identically.
DEVINL void adam1(float& p, float& m, float& v, float g, float lr_t, float b1, float b2, float eps) {
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  p -= lr_t * m / (sqrtf(v) + eps);
}

// Every kernel launch of the library goes through FACT_LAUNCH.  While the in-step kernel-class recorder is armed
// (fact_kprof, engine.hip) the launch is also noted - host function, grid, block, dynamic LDS - so that the bench line can
// name the kernel symbol that really ran and the CUs its grid can hold, instead of a hand-kept table.  One predictable
// branch otherwise.
extern thread_local bool g_fact_note_on;  // true only on a thread that has a recorder scope of an armed handle open
void fact_note_launch(const void* host_fn, dim3 grid, dim3 block, size_t lds_bytes);
#define FACT_LAUNCH(kernel, grid, block, lds, stream, ...)                                                    \
  do {                                                                                                          \
    if (g_fact_note_on) fact_note_launch(reinterpret_cast<const void*>(kernel), (grid), (block), (size_t)(lds)); \
    hipLaunchKernelGGL(kernel, (grid), (block), (lds), (stream), __VA_ARGS__);                                  \
  } while (0)

#define HIP_CHECK_RET(expr)                                  \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) return -100 - (int)_e;             \
  } while (0)
