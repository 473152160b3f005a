// Back-to-back launch cost of (nearly) empty kernels on one stream: what a dependent kernel chain pays per launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ __launch_bounds__(256, 3) void store_kernel(float* p, int n) {
  extern __shared__ float sm[];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (float)i;
}
int main() {
  float* d; hipMalloc(&d, 64 << 20);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute((const void*)store_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  struct { int grid, block, lds, n; const char* what; } cases[] = {
    {1, 64, 0, 0, "empty 1x64"}, {480, 256, 0, 0, "empty 480x256"}, {480, 256, 49152, 0, "empty 480x256 lds48K"},
    {4096, 256, 0, 0, "empty 4096x256"}, {480, 256, 49152, 480 * 256, "store 0.5MB lds48K"}, {9000, 256, 0, 9000 * 256, "store 9.2MB"}};
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      const int N = 2000;
      hipEventRecord(a, s);
      for (int i = 0; i < N; ++i) {
        if (c.n) hipLaunchKernelGGL(store_kernel, dim3(c.grid), dim3(c.block), c.lds, s, d, c.n);
        else hipLaunchKernelGGL(empty_kernel, dim3(c.grid), dim3(c.block), c.lds, s, (float*)nullptr);
      }
      hipEventRecord(b, s); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep) printf("%-28s %.2f us per launch (2000 back-to-back on one stream)\n", c.what, ms * 1e3 / N);
    }
  }
  return 0;
}
