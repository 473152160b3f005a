// HBM ceiling probe for the optimizer pass: flat streaming kernels with Adam's access mix (no transposes).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/adam_probe tools/adam_probe.hip && tools/bin/adam_probe
// variants: R = float4 streams read, W = float4 streams written, +S = one bf16x4 stream written; nt = non-temporal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int NR, int NW, int NS, bool NT, int UNR>
__global__ __launch_bounds__(256) void k(f32x4* a0, f32x4* a1, f32x4* a2, f32x4* a3, bf16x4* s0, bf16x4* s1, size_t n4) {
  f32x4* arr[4] = {a0, a1, a2, a3};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * UNR) {
    f32x4 v[UNR][4];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n4) {
#pragma unroll
        for (int r = 0; r < NR; ++r) v[u][r] = NT ? __builtin_nontemporal_load(arr[r] + i) : arr[r][i];
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n4) {
        f32x4 acc = v[u][0];
#pragma unroll
        for (int r = 1; r < NR; ++r) acc = acc * 0.999f + v[u][r];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          if (NT) __builtin_nontemporal_store(acc + (float)w, arr[w] + i);
          else arr[w][i] = acc + (float)w;
        }
        if (NS >= 1) {
          bf16x4 o = {(__bf16)acc[0], (__bf16)acc[1], (__bf16)acc[2], (__bf16)acc[3]};
          if (NT) __builtin_nontemporal_store(o, s0 + i); else s0[i] = o;
          if (NS >= 2) { if (NT) __builtin_nontemporal_store(o, s1 + i); else s1[i] = o; }
        }
      }
    }
  }
}

// tiled walk: block = TH x TW tile of a [R][C] fp32 matrix (4 streams read, 4 written, bf16 row shadow + optional transposed shadow)
template <int TH, int TW, bool NT, bool TSH, int MATH = 0>
__global__ __launch_bounds__(256) void kt(f32x4* a0, f32x4* a1, f32x4* a2, f32x4* a3, __bf16* s0, __bf16* s1, int R, int C) {
  __shared__ float tile[TSH ? TH : 1][TW + 1];
  const int tc = C / TW;
  const int t = blockIdx.x, r0 = (t / tc) * TH, c0 = (t % tc) * TW;
  constexpr int CQ = TW / 4, RP = 256 / CQ;
  const int cx = (threadIdx.x % CQ) * 4, ry = threadIdx.x / CQ;
  float* arr[4] = {(float*)a0, (float*)a1, (float*)a2, (float*)a3};
#pragma unroll
  for (int g0 = 0; g0 < TH; g0 += 4 * RP) {
    f32x4 v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t o = (size_t)(r0 + ry + g0 + u * RP) * C + c0 + cx;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[u][q] = NT ? __builtin_nontemporal_load((f32x4*)(arr[q] + o)) : *(f32x4*)(arr[q] + o);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = ry + g0 + u * RP;
      const size_t o = (size_t)(r0 + rr) * C + c0 + cx;
      f32x4 acc = v[u][0] * 0.999f + v[u][1] + v[u][2] * 0.5f + v[u][3];
      f32x4 outv[4] = {acc, acc + 1.f, acc + 2.f, acc + 3.f};
      if (MATH) {  // Keras Adam: g = v[0], m = v[1], vv = v[2], p = v[3]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = v[u][0][j];
          float m = 0.9f * v[u][1][j] + 0.1f * g, vv = 0.999f * v[u][2][j] + 0.001f * g * g, pp = v[u][3][j];
          if (MATH == 1) pp -= 1e-3f * m / (sqrtf(vv) + 1e-7f);
          else pp -= 1e-3f * m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv) + 1e-7f);
          outv[0][j] = 0.f; outv[1][j] = m; outv[2][j] = vv; outv[3][j] = pp; acc[j] = pp;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { if (NT) __builtin_nontemporal_store(outv[q], (f32x4*)(arr[q] + o)); else *(f32x4*)(arr[q] + o) = outv[q]; }
      bf16x4 o16 = {(__bf16)acc[0], (__bf16)acc[1], (__bf16)acc[2], (__bf16)acc[3]};
      *(bf16x4*)(s0 + o) = o16;
      if (TSH) { tile[rr][cx] = acc[0]; tile[rr][cx + 1] = acc[1]; tile[rr][cx + 2] = acc[2]; tile[rr][cx + 3] = acc[3]; }
    }
  }
  if (TSH) {
    __syncthreads();
    constexpr int RQ = TH / 4;             // row quads per column
    const int rq = (threadIdx.x % RQ) * 4, cy = threadIdx.x / RQ;
    for (int cc = cy; cc < TW; cc += 256 / RQ) {
      bf16x4 o = {(__bf16)tile[rq][cc], (__bf16)tile[rq + 1][cc], (__bf16)tile[rq + 2][cc], (__bf16)tile[rq + 3][cc]};
      *(bf16x4*)(s1 + (size_t)(c0 + cc) * R + r0 + rq) = o;
    }
  }
}
template <int TH, int TW, bool NT, bool TSH, int MATH = 0>
void runt(const char* name, f32x4** a, bf16x4** s, int R, int C) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = (R / TH) * (C / TW);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((kt<TH, TW, NT, TSH, MATH>), dim3(blocks), dim3(256), 0, 0, a[0], a[1], a[2], a[3], (__bf16*)s[0], (__bf16*)s[1], R, C);
  hipEventRecord(e0);
  const int iters = 5;
  for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((kt<TH, TW, NT, TSH, MATH>), dim3(blocks), dim3(256), 0, 0, a[0], a[1], a[2], a[3], (__bf16*)s[0], (__bf16*)s[1], R, C);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  const double bytes = (double)R * C * (32.0 + 2.0 + (TSH ? 2.0 : 0.0));
  printf("tiled %3dx%3d %-22s R %d C %d: %7.3f ms  %6.0f GB/s\n", TH, TW, name, R, C, ms, bytes / ms / 1e6);
}

template <int NR, int NW, int NS, bool NT, int UNR>
void run(const char* name, f32x4** a, bf16x4** s, size_t n4, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<NR, NW, NS, NT, UNR>), dim3(blocks), dim3(256), 0, 0, a[0], a[1], a[2], a[3], s[0], s[1], n4);
  hipEventRecord(e0);
  const int iters = 5;
  for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((k<NR, NW, NS, NT, UNR>), dim3(blocks), dim3(256), 0, 0, a[0], a[1], a[2], a[3], s[0], s[1], n4);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  const double bytes = (double)n4 * (16.0 * (NR + NW) + 8.0 * NS);
  printf("%-28s blocks %6d unr %d: %7.3f ms  %6.0f GB/s  (%.2f GB)\n", name, blocks, UNR, ms, bytes / ms / 1e6, bytes / 1e9);
}

int main() {
  const size_t n = 120406977 / 4 * 4, n4 = n / 4;
  f32x4* a[4]; bf16x4* s[2];
  for (int i = 0; i < 4; ++i) { hipMalloc((void**)&a[i], n * 4); hipMemset(a[i], 0, n * 4); }
  for (int i = 0; i < 2; ++i) { hipMalloc((void**)&s[i], n * 2); hipMemset(s[i], 0, n * 2); }
  for (int blocks : {2048, 8192, 32768}) {
    run<4, 4, 2, false, 1>("R4 W4 S2 (adam today)", a, s, n4, blocks);
    run<4, 4, 2, true, 1>("R4 W4 S2 nt", a, s, n4, blocks);
    run<4, 3, 2, false, 1>("R4 W3 S2 (no zeroing)", a, s, n4, blocks);
    run<4, 3, 2, true, 1>("R4 W3 S2 nt", a, s, n4, blocks);
    run<4, 3, 2, false, 2>("R4 W3 S2 unr2", a, s, n4, blocks);
    run<4, 3, 2, true, 4>("R4 W3 S2 nt unr4", a, s, n4, blocks);
    run<1, 1, 0, false, 4>("copy R1 W1 unr4", a, s, n4, blocks);
    run<4, 0, 0, false, 2>("read only R4", a, s, n4, blocks);
  }
  // one big [R][C] matrix of the same total size: C = 3072, R = n / C rounded down to a multiple of 256
  const int C = 3072, R = (int)(n / C) / 256 * 256;
  runt<64, 64, false, false>("plain, no t-shadow", a, s, R, C);
  runt<64, 64, false, true>("plain + t-shadow", a, s, R, C);
  runt<64, 64, true, true>("nt + t-shadow", a, s, R, C);
  runt<64, 128, false, true>("plain + t-shadow", a, s, R, C);
  runt<64, 128, true, true>("nt + t-shadow", a, s, R, C);
  runt<64, 256, false, true>("plain + t-shadow", a, s, R, C);
  runt<64, 256, true, true>("nt + t-shadow", a, s, R, C);
  runt<128, 64, true, true>("nt + t-shadow", a, s, R, C);
  runt<64, 64, false, true, 1>("plain+tsh ADAM ieee", a, s, R, C);
  runt<64, 64, false, true, 2>("plain+tsh ADAM fast", a, s, R, C);
  runt<64, 128, true, true, 1>("nt+tsh ADAM ieee", a, s, R, C);
  runt<64, 128, true, true, 2>("nt+tsh ADAM fast", a, s, R, C);
  runt<16, 256, true, false>("nt no t-shadow", a, s, R, C);
  runt<64, 256, true, false>("nt no t-shadow", a, s, R, C);
  return 0;
}
