// cycles per MFMA instruction on one SIMD (one wave per SIMD, 4 independent accumulators): is the legacy K=8 form half the time of K=16?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  bf16x8 a, b; s16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * i); }
  for (int i = 0; i < 4; ++i) { a4[i] = (short)(threadIdx.x + i); b4[i] = (short)(3 * i); }
  f32x16 acc[4]; f32x4 c4[4];
  for (int j = 0; j < 4; ++j) { for (int i = 0; i < 16; ++i) acc[j][i] = 0.f; for (int i = 0; i < 4; ++i) c4[j][i] = 0.f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      if (KIND == 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
      if (KIND == 1) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[m & 3], 0, 0, 0);
      if (KIND == 2) c4[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4[m & 3], 0, 0, 0);
      if (KIND == 3) c4[m & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c4[m & 3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) { for (int i = 0; i < 16; ++i) s += acc[j][i]; for (int i = 0; i < 4; ++i) s += c4[j][i]; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> static void run(const char* name) {
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, 4000);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %.2f ns per MFMA per SIMD\n", name, ms * 1e6 / 4000 / 32);
}
int main() {
  run<0>("v_mfma_f32_32x32x16_bf16");
  run<1>("v_mfma_f32_32x32x8_bf16_1k");
  run<2>("v_mfma_f32_16x16x32_bf16");
  run<3>("v_mfma_f32_16x16x16_bf16_1k");
  return 0;
}
