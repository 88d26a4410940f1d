// Micro-benchmark: do the MFMA phase of one wave and the VALU phase of ANOTHER wave on the same SIMD overlap?
// Each wave loops { NM dependent-free MFMAs ; NV plain VALU (optionally NE v_exp among them) }.  Launched with
// 1 wave per SIMD (256 threads, 1 WG/CU via LDS) and 2 waves per SIMD (two WGs per CU), same total work per wave.
// hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip && ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NM, int NV, int NE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int lds_touch) {
  extern __shared__ float sh[];
  if (lds_touch) sh[threadIdx.x] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * i); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.01f * i + 0.001f * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      if (n < NE) asm volatile("v_exp_f32 %0, %0" : "+v"(v[n & 7]));
      else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[n & 7]) : "v"(v[(n + 1) & 7]));
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// dependent form: the VALU block reads the accumulators (waits for the MFMA chain) and the next iteration's MFMA operand
// is produced by the VALU block -- the S -> softmax -> P.V dependency chain of an attention tile
template <int NM, int NV, int NE>
__global__ __launch_bounds__(256) void kd(float* out, int iters, int lds_touch) {
  extern __shared__ float sh[];
  if (lds_touch) sh[threadIdx.x] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * i); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.01f * i + 0.001f * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += acc[i & 3][i];            // waits for the chain
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      if (n < NE) asm volatile("v_exp_f32 %0, %0" : "+v"(v[n & 7]));
      else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[n & 7]) : "v"(v[(n + 1) & 7]));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)v[i];               // next chain's operand
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the real arithmetic of the bf16x3 dQ tile between its MFMA segments (no LDS, no loads): S^T and dP^T chains (24 MFMAs),
// p = exp2(s*c - L), dS = p*(dp*keep - delta), split into hi / lo bf16, 12 MFMAs with the split operands
__device__ __forceinline__ void split8(const f32x16& x, int s, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float v = x[8 * s + j]; const __bf16 h = (__bf16)v; hi[j] = h; lo[j] = (__bf16)(v - (float)h); }
}
template <int MODE>   // 0 full, 1 no split (plain cvt), 2 no exp
__global__ __launch_bounds__(256) void kreal(float* out, int iters, int lds_touch) {
  extern __shared__ float sh[];
  if (lds_touch) sh[threadIdx.x] = 0.f;
  bf16x8 a, b, d0h, d0l, d1h, d1l;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * i); d0h[i] = a[i]; d0l[i] = b[i]; d1h[i] = a[i]; d1l[i] = b[i]; }
  f32x16 dq0, dq1;
  for (int i = 0; i < 16; ++i) { dq0[i] = 0.f; dq1[i] = 0.f; }
  const float c = 0.18f, L = 3.f, dl = 0.01f;
  f32x16 sp, dpp;
  for (int i = 0; i < 16; ++i) { sp[i] = 0.1f * i; dpp[i] = 0.2f * i; }
  for (int it = 0; it < iters; ++it) {
    f32x16 s, dp;
    for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
    for (int m = 0; m < 12; ++m) { s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, MODE >= 3 ? b : d0h, s, 0, 0, 0); dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, MODE >= 3 ? a : d1h, dp, 0, 0, 0); }
    if (MODE >= 3) { const f32x16 ts = s, td = dp; s = sp; dp = dpp; sp = ts; dpp = td; }   // arithmetic on the previous tile's scores
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = MODE == 2 ? fmaf(s[r], c, -L) : __builtin_amdgcn_exp2f(fmaf(s[r], c, -L));
      s[r] = p * (dp[r] * 1.1f - dl);
    }
    if (MODE == 1) {
      for (int j = 0; j < 8; ++j) { d0h[j] = (__bf16)s[j]; d1h[j] = (__bf16)s[8 + j]; }
    } else {
      split8(s, 0, d0h, d0l);
      split8(s, 1, d1h, d1l);
    }
    if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 24; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 7, 0); }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d0h, dq0, 0, 0, 0); dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, d0l, dq1, 0, 0, 0);
      dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d1h, dq0, 0, 0, 0); dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, d1l, dq1, 0, 0, 0);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += dq0[i] + dq1[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE>
static void run_real(const char* name) {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto kern = kreal<MODE>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int occ = 1; occ <= 2; ++occ) {
    const int shm = occ == 1 ? 100 * 1024 : 60 * 1024;
    hipLaunchKernelGGL(kern, dim3(256 * occ), dim3(256), shm, 0, out, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256 * occ), dim3(256), shm, 0, out, iters, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s 36 MFMA + real dQ arithmetic  waves/SIMD=%d  %.3f ms  %.0f ns per iteration per SIMD\n", name, occ, ms, ms * 1e6 / iters / occ);
  }
  hipFree(out);
}

// forward attention tile (bf16x3, 32-key block, 16 scores per lane): S = 12 MFMAs, online softmax + pair split (+ a 10-op hash per
// score pair when DROP), PV = 12 MFMAs.  SEQ: S(j) -> softmax(j) -> PV(j).  PIPE: one region holds PV(j-1) and S(j+1) (24 MFMAs) next
// to softmax(j), which depends on neither (three-stage software pipeline, deferred rescale of O).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x = __umul24(x, 0x7b352dU) + x; x ^= x >> 13; x = __umul24(x, 0x6ca68bU) + x; x ^= x >> 16; return x;
}
template <bool DROP>
__device__ __forceinline__ void fwd_softmax(f32x16& s, float& m, float& l, float& alpha, uint32_t ctr, bf16x8& p0h, bf16x8& p0l, bf16x8& p1h, bf16x8& p1l) {
  float mt = s[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
  mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * 0.18f;
  const float mn = fmaxf(m, mt);
  alpha = __builtin_amdgcn_exp2f(m - mn);
  m = mn;
  float ls = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { const float p = __builtin_amdgcn_exp2f(fmaf(s[r], 0.18f, -mn)); s[r] = p; ls += p; }
  l = l * alpha + ls;
  if (DROP) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const uint32_t hsh = mix32(ctr + r);
      s[r] = (hsh & 0xFFFFu) >= 6554u ? s[r] : 0.f;
      s[r + 1] = (hsh >> 16) >= 6554u ? s[r + 1] : 0.f;
    }
  }
  split8(s, 0, p0h, p0l);
  split8(s, 1, p1h, p1l);
}
template <int MODE, bool DROP>   // 0 SEQ, 1 PIPE, 2 PIPE + sched_group_barrier
__global__ __launch_bounds__(256) void kfwd(float* out, int iters, int lds_touch) {
  extern __shared__ float sh[];
  if (lds_touch) sh[threadIdx.x] = 0.f;
  bf16x8 a, b, c, d;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * i); c[i] = (__bf16)(0.01f * i); d[i] = (__bf16)(0.003f * i); }
  f32x16 o0, o1, sc, sn;
  for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; sc[i] = 0.1f * i; sn[i] = 0.f; }
  float m = -1e30f, l = 0.f, alpha = 1.f;
  bf16x8 p0h = a, p0l = b, p1h = c, p1l = d;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      f32x16 s;
      for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d, s, 0, 0, 0); s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, c, s, 0, 0, 0); s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, c, s, 0, 0, 0); }
      fwd_softmax<DROP>(s, m, l, alpha, it * 64 + threadIdx.x, p0h, p0l, p1h, p1l);
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p0l, o0, 0, 0, 0); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, p0h, o0, 0, 0, 0); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p0h, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, p0l, o1, 0, 0, 0); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d, p0h, o1, 0, 0, 0); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, p0h, o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p1l, o0, 0, 0, 0); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, p1h, o0, 0, 0, 0); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p1h, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, p1l, o1, 0, 0, 0); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d, p1h, o1, 0, 0, 0); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, p1h, o1, 0, 0, 0);
    } else {
      // deferred rescale of O by the factor of the block whose P is about to be accumulated
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
      __builtin_amdgcn_sched_barrier(0);
      // region: PV(j-1) with the splits of the previous step, S(j+1) into sn, softmax(j) on sc
      const bf16x8 q0h = p0h, q0l = p0l, q1h = p1h, q1l = p1l;
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q0l, o0, 0, 0, 0); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, q0h, o0, 0, 0, 0); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q0h, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, q0l, o1, 0, 0, 0); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d, q0h, o1, 0, 0, 0); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, q0h, o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q1l, o0, 0, 0, 0); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, q1h, o0, 0, 0, 0); o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q1h, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, q1l, o1, 0, 0, 0); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d, q1h, o1, 0, 0, 0); o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, q1h, o1, 0, 0, 0);
      for (int i = 0; i < 16; ++i) sn[i] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d, sn, 0, 0, 0); sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, c, sn, 0, 0, 0); sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, c, sn, 0, 0, 0); }
      fwd_softmax<DROP>(sc, m, l, alpha, it * 64 + threadIdx.x, p0h, p0l, p1h, p1l);
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 24; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, DROP ? 10 : 6, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);
      sc = sn;
    }
  }
  float r = m + l;
  for (int i = 0; i < 16; ++i) r += o0[i] + o1[i] + sc[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE, bool DROP>
static void run_fwd(const char* name) {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto kern = kfwd<MODE, DROP>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int occ = 1; occ <= 2; ++occ) {
    const int shm = occ == 1 ? 100 * 1024 : 60 * 1024;
    hipLaunchKernelGGL(kern, dim3(256 * occ), dim3(256), shm, 0, out, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256 * occ), dim3(256), shm, 0, out, iters, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s 24 MFMA + forward-tile arithmetic%s  waves/SIMD=%d  %.3f ms  %.0f ns per block per SIMD\n", name, DROP ? " + dropout" : "", occ, ms, ms * 1e6 / iters / occ);
  }
  hipFree(out);
}

template <int NM, int NV, int NE, bool DEP = false>
static void run(const char* name) {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto kern = DEP ? kd<NM, NV, NE> : k<NM, NV, NE>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int occ = 1; occ <= 4; ++occ) {
    const int shm = occ == 1 ? 100 * 1024 : occ == 2 ? 60 * 1024 : occ == 3 ? 50 * 1024 : 36 * 1024;   // workgroups per CU
    const int grid = 256 * occ;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shm, 0, out, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shm, 0, out, iters, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // cycles per wave-iteration at a nominal 2.4 GHz (the clock may be lower under load: compare rows, not absolutes)
    printf("%-28s NM=%2d NV=%3d NE=%2d  waves/SIMD=%d  %.3f ms  %.0f ns per iteration per SIMD (%.0f per wave-iteration)\n", name, NM, NV, NE, occ,
           ms, ms * 1e6 / iters / occ, ms * 1e6 / iters);
  }
  hipFree(out);
}

int main() {
  run_fwd<0, false>("fwd block, sequential");
  run_fwd<1, false>("fwd block, 3-stage pipeline");
  run_fwd<2, false>("fwd block, pipeline + sgb");
  run_fwd<0, true>("fwd block, sequential");
  run_fwd<1, true>("fwd block, 3-stage pipeline");
  run_fwd<2, true>("fwd block, pipeline + sgb");
  run_real<0>("real dq tile");
  run_real<1>("real dq tile, no split");
  run_real<2>("real dq tile, no exp");
  run_real<3>("real dq tile, skewed");
  run_real<4>("real dq tile, skewed+sgb");
  run<36, 0, 0>("mfma only");
  run<0, 200, 0>("valu only");
  run<0, 200, 16>("valu+exp only");
  run<24, 200, 0>("mfma then valu");
  run<24, 200, 16>("mfma then valu+exp");
  run<36, 170, 16>("x3 dq-like");
  run<16, 400, 32>("bf16 fwd-like");
  run<24, 200, 16, true>("dependent mfma->valu->mfma");
  run<36, 170, 16, true>("dependent x3 dq-like");
  run<16, 400, 32, true>("dependent bf16 fwd-like");
  return 0;
}
