// Dependent-launch boundary on one stream: trivial kernel vs the same kernel with a GemmParams-sized by-value kernarg (320 B),
// with / without an event record + cross-stream wait between launches.  Run under HIP_FORCE_DEV_KERNARG=0/1.
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { void* p[24]; int v[32]; };  // 320 bytes
__global__ void k_small(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void k_big(Big b) { if (b.p[3] && threadIdx.x == 9999) ((float*)b.p[3])[b.v[5]] = 1.f; }
__global__ __launch_bounds__(512) void k_lds(Big b) { extern __shared__ float sm[]; if (b.p[3] && threadIdx.x == 9999) sm[b.v[5]] = 1.f; }
int main() {
  hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t a, b, e[64]; hipEventCreate(&a); hipEventCreate(&b);
  for (auto& x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
  hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
  Big big = {};
  const int N = 3000;
  for (int mode = 0; mode < 6; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a, s);
      for (int i = 0; i < N; ++i) {
        switch (mode) {
          case 0: hipLaunchKernelGGL(k_small, dim3(240), dim3(512), 0, s, (float*)nullptr); break;
          case 1: hipLaunchKernelGGL(k_big, dim3(240), dim3(512), 0, s, big); break;
          case 2: hipLaunchKernelGGL(k_lds, dim3(240), dim3(512), 136 * 1024, s, big); break;
          case 3: hipLaunchKernelGGL(k_big, dim3(240), dim3(512), 0, s, big); hipEventRecord(e[i & 63], s); break;
          case 4: hipLaunchKernelGGL(k_big, dim3(240), dim3(512), 0, s, big); hipEventRecord(e[i & 63], s); hipStreamWaitEvent(s2, e[i & 63], 0); break;
          case 5: hipLaunchKernelGGL(k_big, dim3(240), dim3(512), 0, s, big); hipEventRecord(e[i & 63], s2); hipStreamWaitEvent(s, e[i & 63], 0); break;
        }
      }
      hipEventRecord(b, s); hipEventSynchronize(b); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, a, b);
      const char* what[] = {"8B kernarg", "320B kernarg", "320B kernarg + 136KB LDS", "320B + event record", "320B + record + other stream waits", "320B + wait on other stream's event"};
      if (rep) printf("%-40s %.2f us per launch\n", what[mode], ms * 1e3 / N);
    }
  }
  return 0;
}
