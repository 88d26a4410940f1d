// Does s_store_dwordx2 (+ s_dcache_wb) work on gfx950?  Each wave stores 4 ballots through the scalar path; the host checks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned long long* out, int n) {
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  for (int r = 0; r < 4; ++r) {
    const unsigned long long m = __ballot(((lane * 2654435761u + wave * 40503u + r * 7u) >> 7) & 1);
    unsigned long long* p = out + (size_t)wave * 4 + r;
    asm volatile("s_store_dwordx2 %0, %1, 0x0" :: "s"(m), "s"(p) : "memory");
  }
  asm volatile("s_dcache_wb" ::: "memory");
}
int main() {
  const int blocks = 4096, threads = 256, waves = blocks * threads / 64;
  unsigned long long* d;
  (void)hipMalloc(&d, sizeof(unsigned long long) * waves * 4);
  (void)hipMemset(d, 0xff, sizeof(unsigned long long) * waves * 4);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, waves);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(waves * 4);
  (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int w = 0; w < waves; ++w)
    for (int r = 0; r < 4; ++r) {
      unsigned long long m = 0;
      for (int lane = 0; lane < 64; ++lane)
        if (((unsigned)(lane * 2654435761u + w * 40503u + r * 7u) >> 7) & 1) m |= 1ull << lane;
      if (h[w * 4 + r] != m) ++bad;
    }
  printf("scalar-store test: %ld wrong of %d\n", bad, waves * 4);
  return bad != 0;
}
