// Which XCDs / CUs does a CU-masked stream (hipExtStreamCreateWithCUMask) run on?  Prints per-XCC block counts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void who(unsigned* out) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hwid; }
  // hold the CU for a while so that blocks spread over every CU the mask allows
  for (volatile int i = 0; i < 20000; ++i) {}
}
int main() {
  unsigned* d; hipMalloc(&d, 4096 * 8);
  const int nblk = 512;
  for (int pat = 0; pat < 4; ++pat) {
    std::vector<uint32_t> mask(8, 0);
    for (int i = 0; i < 256; ++i) {
      bool on = pat == 0 ? true : pat == 1 ? (i < 96) : pat == 2 ? ((i % 8) < 3) : (i >= 160);
      if (on) mask[i / 32] |= 1u << (i % 32);
    }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
    if (e != hipSuccess) { printf("pattern %d: create failed %d\n", pat, (int)e); continue; }
    hipLaunchKernelGGL(who, dim3(nblk), dim3(256), 65536, s, d);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * nblk);
    hipMemcpy(h.data(), d, 2 * nblk * 4, hipMemcpyDeviceToHost);
    int cnt[16] = {0};
    std::vector<int> cus;
    for (int b = 0; b < nblk; ++b) {
      cnt[h[2 * b] & 15]++;
      int key = (h[2 * b] & 15) * 1024 + ((h[2 * b + 1] >> 8) & 0xf) + 16 * ((h[2 * b + 1] >> 13) & 7);  // xcc, cu, se
      bool seen = false; for (int c : cus) if (c == key) seen = true;
      if (!seen) cus.push_back(key);
    }
    const char* names[] = {"all 256 bits", "bits 0..95", "bits with i%8 < 3", "bits 160..255"};
    printf("%-18s distinct (xcc,se,cu): %3zu   blocks per XCC:", names[pat], cus.size());
    for (int x = 0; x < 8; ++x) printf(" %3d", cnt[x]);
    printf("\n");
    hipStreamDestroy(s);
  }
  return 0;
}
