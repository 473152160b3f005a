// Micro-probe: how fast can one CU pull operand tiles? LDS-DMA (global_load_lds) vs register loads,
// as a function of bytes in flight, from an L2-resident or HBM-streaming source.
// build: hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o tools/bin/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

template <int N> __device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// each wave: DEPTH groups of 4 x 1KiB glds in flight; ring of DEPTH+1 slots of 4 KiB per wave
template <int DEPTH>
__global__ __launch_bounds__(256) void glds_stream(const unsigned char* src, size_t span_bytes, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* wbase = smem + wave * (DEPTH + 1) * 4096;
  // block's stream start; wraps inside span
  size_t off = (size_t)blockIdx.x * 81920 + (size_t)(threadIdx.x >> 6) * 4096 + (threadIdx.x & 63) * 16;
  const size_t step = 1024;
  auto issue = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gbl_cvoid*)(src + (off & (span_bytes - 1))), (lds_void*)(wbase + slot * 4096 + i * 1024), 16, 0, 0);
      off += step;
    }
    off += 12288;
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d);
  int slot = DEPTH % (DEPTH + 1);
  for (int it = 0; it < iters; ++it) {
    wait_vm<(DEPTH - 1) * 4>();
    issue(slot);
    slot = (slot + 1) % (DEPTH + 1);
  }
  wait_vm<0>();
  __syncthreads();
  if (threadIdx.x == 0 && smem[blockIdx.x & 1023] == 77 && iters < 0) sink[0] = 1;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void reg_stream(const unsigned char* src, size_t span_bytes, int iters, int* sink) {
  size_t off = (size_t)blockIdx.x * 81920 + (size_t)(threadIdx.x >> 6) * 4096 + (threadIdx.x & 63) * 16;
  const size_t step = 1024;
  uint4 r[DEPTH][4];
  uint4 accv = {0, 0, 0, 0};
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[d][i] = *(const uint4*)(src + (off & (span_bytes - 1))); off += step; }
    off += 12288;
  for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { accv.x ^= r[d][i].x; accv.y ^= r[d][i].y; accv.z ^= r[d][i].z; accv.w ^= r[d][i].w; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { r[d][i] = *(const uint4*)(src + (off & (span_bytes - 1))); off += step; }
    off += 12288;
    }
  }
  if ((accv.x ^ accv.y ^ accv.z ^ accv.w) == 0x12345678 && iters < 0) sink[0] = 1;
}

template <typename F>
float time_ms(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
  const size_t BUF = 1ull << 30;
  unsigned char* d; int* sink;
  hipMalloc(&d, BUF); hipMemset(d, 1, BUF); hipMalloc(&sink, 4);
  const int iters = 2000;
  size_t spans[4] = {1ull << 20, 2ull << 20, 64ull << 20, 1ull << 30};
  int grids[3] = {256, 512, 1024};
  for (size_t span : spans) for (int g : grids) {
    printf("span %4zu MiB grid %4d:", span >> 20, g);
#define RUN_G(D) { hipFuncSetAttribute((const void*)glds_stream<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (D + 1) * 4096); \
      float ms = time_ms([&] { hipLaunchKernelGGL(glds_stream<D>, dim3(g), dim3(256), 4 * (D + 1) * 4096, 0, d, span, iters, sink); }); \
      double gb = (double)g * 256 * 16 * 4 * (iters + D) / 1e9; printf("  glds d%d %6.0f GB/s", D, gb / (ms * 1e-3)); }
    RUN_G(1) RUN_G(2) RUN_G(4) RUN_G(8)
#define RUN_R(D) { float ms = time_ms([&] { hipLaunchKernelGGL(reg_stream<D>, dim3(g), dim3(256), 0, 0, d, span, iters, sink); }); \
      double gb = (double)g * 256 * 16 * 4 * (iters + D) / 1e9; printf("  reg d%d %6.0f GB/s", D, gb / (ms * 1e-3)); }
    RUN_R(1) RUN_R(2) RUN_R(4)
    printf("\n");
  }
  return 0;
}
