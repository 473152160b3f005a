// How much HBM traffic can k CUs move?  One 512-thread workgroup per CU (150 KiB of dynamic LDS keeps a second one out), each
// streaming its own contiguous chunk - read-only, or read + write (the mix of an optimizer / residual epilogue) - with 8
// independent 16-byte accesses per lane in flight.  Grid = k workgroups -> k CUs.  Prints GB/s per k: the slope is the per-CU
// streaming rate that bounds every HBM-bound phase inside a launch that holds k CUs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>  // 0 = read (sum), 1 = copy (read + write), 2 = read-modify-write in place
__global__ __launch_bounds__(512) void stream_kernel(f32x4* __restrict__ a, f32x4* __restrict__ b, size_t per_wg, float* sink) {
  extern __shared__ float lds[];
  f32x4* src = a + (size_t)blockIdx.x * per_wg;
  f32x4* dst = b + (size_t)blockIdx.x * per_wg;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = threadIdx.x; i + 7 * 512 < per_wg; i += 8 * 512) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[i + u * 512];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) acc += v[u];
      else if (MODE == 1) dst[i + u * 512] = v[u];
      else src[i + u * 512] = v[u] * 1.0001f;
    }
  }
  if (MODE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = lds[threadIdx.x];
}
int main() {
  const size_t per_wg = (size_t)(8u << 20) / 16;  // 8 MiB per workgroup
  const int kmax = 256;
  f32x4 *a, *b; float* sink;
  hipMalloc(&a, (size_t)kmax * per_wg * 16); hipMalloc(&b, (size_t)kmax * per_wg * 16); hipMalloc(&sink, 64);
  hipMemset(a, 0, (size_t)kmax * per_wg * 16); hipMemset(b, 0, (size_t)kmax * per_wg * 16);
  hipFuncSetAttribute((const void*)stream_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipFuncSetAttribute((const void*)stream_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipFuncSetAttribute((const void*)stream_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ks[] = {16, 32, 64, 95, 115, 128, 161, 192, 225, 256};
  printf("%-6s %14s %14s %14s   (GB/s of HBM traffic: bytes read + bytes written; one workgroup per CU)\n", "CUs", "read", "copy", "rmw in place");
  for (int k : ks) {
    float gbs[3];
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(k), dim3(512), 150 * 1024, 0, a, b, per_wg, sink);
        if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(k), dim3(512), 150 * 1024, 0, a, b, per_wg, sink);
        if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(k), dim3(512), 150 * 1024, 0, a, b, per_wg, sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      const double bytes = (double)k * per_wg * 16 * (mode == 0 ? 1 : 2);
      gbs[mode] = (float)(bytes / (best * 1e-3) / 1e9);
    }
    printf("%-6d %10.0f (%4.1f/CU) %8.0f (%4.1f/CU) %8.0f (%4.1f/CU)\n", k, gbs[0], gbs[0] / k, gbs[1], gbs[1] / k, gbs[2], gbs[2] / k);
  }
  return 0;
}
