// Issue rate of the bf16 MFMA shapes on gfx950: one wave per SIMD, NACC independent accumulators, back-to-back issue.
// Question behind it (round 4): is the legacy k = 16 shape (v_mfma_f32_16x16x16_bf16) half the cycles of 16x16x32, i.e.
// could a head dim of 80 run as 2 x k32 + 1 x k16 instead of being padded to 96?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip -o tools/bin/mfma_rate_probe && tools/bin/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 8 independent accumulate chains, issued through asm so that hipcc cannot shuffle the accumulators between iterations
// (its own code for an array of accumulators carried v_accvgpr_read / write pairs between the MFMAs: first version of
// this probe measured 42 cycles per 16x16x32).  An MFMA -> MFMA chain on one accumulator needs no wait states; the
// s_nop before the final reads covers the MFMA -> VALU hazard hipcc does not pad around asm.
#define CHAIN4(OP, A, B) \
  asm volatile(OP " %0, %8, %9, %0\n\t" OP " %1, %8, %9, %1\n\t" OP " %2, %8, %9, %2\n\t" OP " %3, %8, %9, %3\n\t" \
               OP " %4, %8, %9, %4\n\t" OP " %5, %8, %9, %5\n\t" OP " %6, %8, %9, %6\n\t" OP " %7, %8, %9, %7" \
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(A), "v"(B))
#define CHAIN16(OP, A, B) \
  asm volatile(OP " %0, %4, %5, %0\n\t" OP " %1, %4, %5, %1\n\t" OP " %2, %4, %5, %2\n\t" OP " %3, %4, %5, %3" \
               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(A), "v"(B))
template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int iters) {
  bf16x8 a8, b8;
  s16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(1.0f + i * 0.5f); }
  for (int i = 0; i < 4; ++i) { a4[i] = (short)(threadIdx.x + i); b4[i] = (short)(0x3f80 + i); }
  f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
  f32x16 d0 = {}, d1 = {}, d2 = {}, d3 = {};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) CHAIN4("v_mfma_f32_16x16x32_bf16", a8, b8);
    else if constexpr (KIND == 1) CHAIN4("v_mfma_f32_16x16x16_bf16", a4, b4);
    else if constexpr (KIND == 2) CHAIN16("v_mfma_f32_32x32x16_bf16", a8, b8);
    else CHAIN16("v_mfma_f32_32x32x8_bf16", a4, b4);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3] + d0[0] + d1[5] + d2[10] + d3[15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter, double flop_each) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * per_iter;
  printf("%-28s %6.2f shader cycles / MFMA, wall %.3f ms -> %.1f ns / MFMA / SIMD, %.0f TFLOP/s chip\n", name,
         (double)h[0] / n, ms, ms * 1e6 / n, n * flop_each * 1024 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("v_mfma_f32_16x16x32_bf16", 8, 2.0 * 16 * 16 * 32);
  run<1>("v_mfma_f32_16x16x16_bf16", 8, 2.0 * 16 * 16 * 16);
  run<2>("v_mfma_f32_32x32x16_bf16", 4, 2.0 * 32 * 32 * 16);
  run<3>("v_mfma_f32_32x32x8_bf16", 4, 2.0 * 32 * 32 * 8);
  return 0;
}
