// Micro-benchmarks behind the round-3 attention restructuring (register-only: no LDS, no global traffic in the loops).
//   1. VALU issue rate per SIMD at 1 / 2 / 3 waves per SIMD for the instruction kinds of the softmax (is the vector pipe 2 or 4
//      cycles per wave64 instruction once two waves feed it?)
//   2. wave-specialised co-issue: waves 0-3 only MFMA, waves 4-7 only VALU, against each alone
//   3. the single-pass forward tile (64 keys x 32 queries per wave: 16 MFMA + softmax of 32 scores per lane), as
//        SEQ    S -> softmax -> PV in one wave, 1 / 2 / 3 waves per SIMD (what k_attn_fwd_mfma does)
//        PING   512-thread workgroup, the two waves of a SIMD in opposite phases separated by s_barrier
//        STAG   512-thread workgroup, no barriers, waves 4-7 start half a tile late
//        PIPE   one wave: softmax(j) in the same region as PV(j-1) and S(j+1), compiler-scheduled / sched_group_barrier
//      each with the current arithmetic (hash dropout, fma+exp, unconditional rescale) and the reduced one (keep bits as SGPR
//      masks, pre-scaled scores, rescale skipped)
// hipcc --offload-arch=gfx950 -O3 -o pingpong pingpong.hip && ./pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned long long u64;
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// ------------------------------------------------------------------------------------------------ 1. VALU rates
template <int OP>
__global__ __launch_bounds__(256) void k_valu(u64* cyc, float* out, int iters) {
  extern __shared__ float sh[];
  if (iters < 0) sh[threadIdx.x] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.01f * i + 0.001f * threadIdx.x;
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 977u + i;
  const u64 t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < 64; ++n) {
      const int r = n & 7, r2 = (n + 3) & 7;
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(v[r2]));
      if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
      if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[r2]));
      if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(v[r2]));
      if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[r]) : "v"(u[r2]));
      if (OP == 5) asm volatile("v_lshrrev_b32 %0, 13, %1\n\tv_xor_b32 %1, %0, %1" : "+v"(u[r]), "+v"(u[r2]));
      if (OP == 6) asm volatile("v_cmp_ge_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, 0, %1, vcc" : "+v"(u[r]) : "v"(u[r2]) : "vcc");
      if (OP == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[r2]));
      if (OP == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&v[2 * (r & 3)]) : "v"(*(double*)&v[2 * (r2 & 3)]));
      if (OP == 9) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[r2]));
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ softmax tile arithmetic
struct Tile {
  f32x16 s0, s1;      // scores of the two 32-key blocks (transposed: keys in registers)
};
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x = __umul24(x, 0x7b352dU) + x; x ^= x >> 13; x = __umul24(x, 0x6ca68bU) + x; x ^= x >> 16; return x;
}
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
__device__ __forceinline__ bf16x8 cvt8(const f32x16& x, int s) {
  return (bf16x8){(__bf16)x[8 * s + 0], (__bf16)x[8 * s + 1], (__bf16)x[8 * s + 2], (__bf16)x[8 * s + 3],
                  (__bf16)x[8 * s + 4], (__bf16)x[8 * s + 5], (__bf16)x[8 * s + 6], (__bf16)x[8 * s + 7]};
}
// ARITH 0: the current kernel's arithmetic (fma + exp2, hash dropout of every pair, unconditional rescale of O)
// ARITH 1: reduced: scores arrive pre-scaled with -m in the accumulator (p = exp2(s)), keep bits as 64-bit lane masks in SGPRs,
//          O rescaled only when some row maximum moved by more than 8 (wave-uniform branch)
// ARITH 2: reduced, but still hashing the dropout
template <int ARITH>
__device__ __forceinline__ void softmax_tile(Tile& t, float& m, float& l, f32x16& o0, f32x16& o1, unsigned ctr, const u64* __restrict__ gm,
                                             bf16x8 (&p)[4]) {
  float mt = fmaxf(t.s0[0], t.s1[0]);
#pragma unroll
  for (int r = 1; r < 16; ++r) mt = max3_raw(mt, t.s0[r], t.s1[r]);
  mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
  float ls = 0.f;
  if (ARITH == 0) {
    mt *= 0.18f;
    const float mn = fmaxf(m, mt);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) { t.s0[r] = __builtin_amdgcn_exp2f(fmaf(t.s0[r], 0.18f, -mn)); ls += t.s0[r]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) { t.s1[r] = __builtin_amdgcn_exp2f(fmaf(t.s1[r], 0.18f, -mn)); ls += t.s1[r]; }
    l = l * alpha + ls;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
  } else {
    // scores are s - m_old already; rescale only if the block maximum exceeds the running one by > 8
    if (__builtin_expect(__any(mt > 8.f), 0)) {
      const float mn = fmaxf(mt, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-mn);
      m += mn;
#pragma unroll
      for (int r = 0; r < 16; ++r) { t.s0[r] -= mn; t.s1[r] -= mn; }
      l *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { t.s0[r] = __builtin_amdgcn_exp2f(t.s0[r]); ls += t.s0[r]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) { t.s1[r] = __builtin_amdgcn_exp2f(t.s1[r]); ls += t.s1[r]; }
    l += ls;
  }
  if (ARITH == 0 || ARITH == 2) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const unsigned h0 = mix32(ctr + r), h1 = mix32(ctr + 64 + r);
      t.s0[r] = (h0 & 0xFFFFu) >= 6554u ? t.s0[r] : 0.f;
      t.s0[r + 1] = (h0 >> 16) >= 6554u ? t.s0[r + 1] : 0.f;
      t.s1[r] = (h1 & 0xFFFFu) >= 6554u ? t.s1[r] : 0.f;
      t.s1[r + 1] = (h1 >> 16) >= 6554u ? t.s1[r + 1] : 0.f;
    }
  } else {
    const u64* mp = gm + (__builtin_amdgcn_readfirstlane(ctr) >> 7 & 3) * 32;    // wave-uniform: s_load per tile, as the real kernel would
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(u64)mp), hi = __builtin_amdgcn_readfirstlane((unsigned)((u64)mp >> 32));
    const u64* sp = (const u64*)(((u64)hi << 32) | lo);
    u32x16 ma, mb, mc, md;
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx16 %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(ma), "=&s"(mb), "=&s"(mc), "=&s"(md) : "s"(sp) : "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const u32x16& va = r < 8 ? ma : mb;
      const u32x16& vb = r < 8 ? mc : md;
      const u64 k0 = ((u64)va[2 * (r & 7) + 1] << 32) | va[2 * (r & 7)], k1 = ((u64)vb[2 * (r & 7) + 1] << 32) | vb[2 * (r & 7)];
      float y0, y1;
      asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(y0) : "v"(t.s0[r]), "s"(k0));
      asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(y1) : "v"(t.s1[r]), "s"(k1));
      t.s0[r] = y0; t.s1[r] = y1;
    }
  }
  p[0] = cvt8(t.s0, 0); p[1] = cvt8(t.s0, 1); p[2] = cvt8(t.s1, 0); p[3] = cvt8(t.s1, 1);
}
__device__ __forceinline__ void qk_tile(Tile& t, const bf16x8 (&k)[4], const bf16x8 (&q)[4], float init) {
#pragma unroll
  for (int i = 0; i < 16; ++i) { t.s0[i] = init; t.s1[i] = init; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { t.s0 = MFMA(k[ks], q[ks], t.s0); t.s1 = MFMA(k[(ks + 1) & 3], q[ks], t.s1); }
}
__device__ __forceinline__ void pv_tile(f32x16& o0, f32x16& o1, const bf16x8 (&v)[4], const bf16x8 (&p)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { o0 = MFMA(v[i], p[i], o0); o1 = MFMA(v[(i + 1) & 3], p[i], o1); }
}

enum { SEQ = 0, PING = 1, STAG = 2, PIPE = 3, PIPE_SGB = 4 };
template <int STRUCT, int ARITH, int THREADS>
__global__ __launch_bounds__(THREADS) void k_fwd(u64* cyc, float* out, int iters, const u64* gmasks) {
  extern __shared__ float sh[];
  if (iters < 0) sh[threadIdx.x] = 0.f;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16x8 kf[4], qf[4], vf[4], p[4];
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) {
      kf[s][i] = (__bf16)(0.01f * ((threadIdx.x + i + s) % 17) - 0.08f);
      qf[s][i] = (__bf16)(0.02f * ((threadIdx.x * 3 + i + s) % 13) - 0.1f);
      vf[s][i] = (__bf16)(0.01f * ((threadIdx.x + 2 * i + s) % 11));
      p[s][i] = (__bf16)0.01f;
    }
  f32x16 o0, o1;
  for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
  float m = ARITH == 0 ? -1e30f : 0.f, l = 0.f;
  Tile t;
  const float init = ARITH == 0 ? 0.f : -1.0f;
  const u64 t0 = __builtin_amdgcn_s_memtime();
  if (STRUCT == SEQ) {
    _Pragma("unroll 1") for (int it = 0; it < iters; ++it) {
      qk_tile(t, kf, qf, init);
      softmax_tile<ARITH>(t, m, l, o0, o1, it * 128 + threadIdx.x, gmasks, p);
      pv_tile(o0, o1, vf, p);
    }
  } else if (STRUCT == PING || STRUCT == STAG) {
    // M phase: PV of the previous tile's probabilities + S of the next tile; V phase: softmax.  Waves 4-7 run half a period late.
    const bool late = w >= 4;
    if (STRUCT == PING) {
      if (late) __builtin_amdgcn_s_setprio(1);
      qk_tile(t, kf, qf, init);
      if (late) __builtin_amdgcn_s_barrier();
      _Pragma("unroll 1") for (int it = 0; it < iters; ++it) {
        softmax_tile<ARITH>(t, m, l, o0, o1, it * 128 + threadIdx.x, gmasks, p);
        __builtin_amdgcn_s_barrier();
        pv_tile(o0, o1, vf, p);
        qk_tile(t, kf, qf, init);
        __builtin_amdgcn_s_barrier();
      }
      if (!late) __builtin_amdgcn_s_barrier();
    } else {
      if (late) {   // half a tile of delay: one softmax on dummy data
        qk_tile(t, kf, qf, init);
        softmax_tile<ARITH>(t, m, l, o0, o1, threadIdx.x, gmasks, p);
      }
      qk_tile(t, kf, qf, init);
      _Pragma("unroll 1") for (int it = 0; it < iters; ++it) {
        softmax_tile<ARITH>(t, m, l, o0, o1, it * 128 + threadIdx.x, gmasks, p);
        pv_tile(o0, o1, vf, p);
        qk_tile(t, kf, qf, init);
      }
    }
  } else {
    // software pipeline in ONE wave: region = { PV(j-1), S(j+1) } next to softmax(j)
    Tile tn;
    bf16x8 pp[4] = {p[0], p[1], p[2], p[3]};
    qk_tile(t, kf, qf, init);
    _Pragma("unroll 1") for (int it = 0; it < iters; ++it) {
      __builtin_amdgcn_sched_barrier(0);
      pv_tile(o0, o1, vf, pp);
      qk_tile(tn, kf, qf, init);
      f32x16 d0 = o0, d1 = o1;      // (the rescale of softmax(j) applies to O after PV(j-1): model it on copies' registers)
      softmax_tile<ARITH>(t, m, l, d0, d1, it * 128 + threadIdx.x, gmasks, p);
      if (STRUCT == PIPE_SGB) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, ARITH == 1 ? 9 : 28, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);
      o0 = d0; o1 = d1;
      t = tn;
#pragma unroll
      for (int i = 0; i < 4; ++i) pp[i] = p[i];
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  float r = m + l;
  for (int i = 0; i < 16; ++i) r += o0[i] + o1[i] + t.s0[i];
  out[blockIdx.x * THREADS + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (THREADS / 64) + w] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ 2. wave-specialised co-issue
template <int ARITH>
__global__ __launch_bounds__(512) void k_split(u64* cyc, float* out, int iters, int mode, const u64* gmasks) {   // mode 1 MFMA waves only, 2 VALU waves only, 3 both
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16x8 kf[4], qf[4], p[4];
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) { kf[s][i] = (__bf16)(0.01f * ((threadIdx.x + i + s) % 17)); qf[s][i] = (__bf16)(0.02f * ((i + s) % 13)); p[s][i] = (__bf16)0.f; }
  f32x16 o0, o1;
  for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
  Tile t;
  for (int i = 0; i < 16; ++i) { t.s0[i] = 0.1f * i; t.s1[i] = 0.05f * i; }
  float m = 0.f, l = 0.f;
  if ((mode & 4) && w < 4) __builtin_amdgcn_s_setprio(3);
  if ((mode & 8) && w >= 4) __builtin_amdgcn_s_setprio(3);
  const u64 t0 = __builtin_amdgcn_s_memtime();
  if (w < 4) {
    if (mode & 1)
      _Pragma("unroll 1") for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { o0 = MFMA(kf[i], qf[i], o0); o1 = MFMA(kf[(i + 1) & 3], qf[i], o1); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { o0 = MFMA(qf[i], kf[i], o0); o1 = MFMA(qf[(i + 1) & 3], kf[i], o1); }
      }
  } else {
    if (mode & 2)
      _Pragma("unroll 1") for (int it = 0; it < iters; ++it) {
        Tile u = t;
        softmax_tile<ARITH>(u, m, l, o0, o1, it * 128 + threadIdx.x, gmasks, p);
        asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]));
      }
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  float r = m + l;
  for (int i = 0; i < 16; ++i) r += o0[i] + o1[i];
  for (int i = 0; i < 8; ++i) r += (float)p[0][i] + (float)p[3][i];
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ host
static u64* d_cyc; static float* d_out; static u64* d_masks;
static double median(std::vector<u64>& v) { std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }

template <typename K, typename... A>
static void run(const char* name, K kern, int threads, int occ, int iters, double units_per_wave_iter, const char* unit, A... args) {
  const int shm = occ == 1 ? 100 * 1024 : occ == 2 ? 60 * 1024 : 40 * 1024;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int grid = 256 * occ;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), shm, 0, d_cyc, d_out, 10, args...);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), shm, 0, d_cyc, d_out, iters, args...);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const int nw = grid * threads / 64;
  std::vector<u64> c(nw);
  hipMemcpy(c.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost);
  const double cy = median(c);
  const int waves_per_simd = occ * threads / 256;
  // cycles of SIMD time per unit: a SIMD hosts waves_per_simd waves that each did iters * units
  printf("%-44s thr=%d WG/CU=%d waves/SIMD=%d  %8.3f ms  wave-cycles/iter %8.1f  SIMD-cycles per %s %7.2f  (clock %.2f GHz)\n", name, threads, occ,
         waves_per_simd, best, cy / iters, unit, cy / iters / units_per_wave_iter / waves_per_simd, cy / (best * 1e6));
  fflush(stdout);
}

int main() {
  hipMalloc(&d_cyc, 8192 * 8 * 8); hipMalloc(&d_out, 4096 * 512 * 4); hipMalloc(&d_masks, 128 * 8);
  u64 hm[128];
  for (int i = 0; i < 128; ++i) hm[i] = 0xFFFFFFFFFFFFFFFFull ^ (0x0123456789ABCDEFull * (i + 1));
  hipMemcpy(d_masks, hm, sizeof hm, hipMemcpyHostToDevice);
  const char* opn[] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "v_max3_f32", "v_mad_u32_u24", "lshr+xor (2)", "cmp+cndmask (2)", "v_add_f32", "v_pk_mul_f32", "v_cvt_pk_f16_f32"};
  const bool split_only = getenv("PP_SPLIT_ONLY") != nullptr;
  printf("== 1. VALU issue: SIMD-cycles per instruction (64 instructions per iteration)\n");
#define VAL(OP, MULT) for (int occ = 1; occ <= 3 && !split_only; ++occ) run(opn[OP], k_valu<OP>, 256, occ, 2000, 64.0 * MULT, "instr");
  VAL(0, 1) VAL(1, 1) VAL(2, 1) VAL(3, 1) VAL(4, 1) VAL(5, 2) VAL(6, 2) VAL(7, 1) VAL(8, 1) VAL(9, 1)
  printf("== 2. wave-specialised: waves 0-3 16 MFMA per iteration, waves 4-7 one softmax tile per iteration (median wave cycles: of all 8 waves)\n");
  for (int mode = 1; mode <= 3; ++mode) {
    run(mode == 1 ? "split A0: MFMA waves only" : mode == 2 ? "split A0: VALU waves only" : "split A0: both", k_split<0>, 512, 1, 1000, 1.0, "iter", mode, (const u64*)d_masks);
  }
  for (int mode = 2; mode <= 3; ++mode) run(mode == 2 ? "split A1: VALU waves only" : "split A1: both", k_split<1>, 512, 1, 1000, 1.0, "iter", mode, (const u64*)d_masks);
  run("split A0: both, MFMA waves prio 3", k_split<0>, 512, 1, 1000, 1.0, "iter", 3 | 4, (const u64*)d_masks);
  run("split A0: both, VALU waves prio 3", k_split<0>, 512, 1, 1000, 1.0, "iter", 3 | 8, (const u64*)d_masks);
  run("split A1: both, MFMA waves prio 3", k_split<1>, 512, 1, 1000, 1.0, "iter", 3 | 4, (const u64*)d_masks);
  run("split A1: both, VALU waves prio 3", k_split<1>, 512, 1, 1000, 1.0, "iter", 3 | 8, (const u64*)d_masks);
  run("split A2: VALU waves only", k_split<2>, 512, 1, 1000, 1.0, "iter", 2, (const u64*)d_masks);
  run("split A2: both", k_split<2>, 512, 1, 1000, 1.0, "iter", 3, (const u64*)d_masks);
  run("split A2: both, MFMA waves prio 3", k_split<2>, 512, 1, 1000, 1.0, "iter", 3 | 4, (const u64*)d_masks);
  if (getenv("PP_SPLIT_ONLY")) return 0;
  printf("== 3. forward tile (16 MFMA + softmax of 32 scores / lane): SIMD-cycles per tile; MFMA floor = 512\n");
#define FWD(S, A, T, NAME) run(NAME, k_fwd<S, A, T>, T, occ, 1000, 1.0, "tile", (const u64*)d_masks)
  for (int occ = 1; occ <= 3; ++occ) FWD(SEQ, 0, 256, "SEQ  current arithmetic");
  for (int occ = 1; occ <= 3; ++occ) FWD(SEQ, 2, 256, "SEQ  reduced + hash");
  for (int occ = 1; occ <= 3; ++occ) FWD(SEQ, 1, 256, "SEQ  reduced + mask bits");
  { int occ = 1;
    FWD(PING, 0, 512, "PING current arithmetic"); FWD(PING, 2, 512, "PING reduced + hash"); FWD(PING, 1, 512, "PING reduced + mask bits");
    FWD(STAG, 0, 512, "STAG current arithmetic"); FWD(STAG, 2, 512, "STAG reduced + hash"); FWD(STAG, 1, 512, "STAG reduced + mask bits");
  }
  for (int occ = 1; occ <= 2; ++occ) { FWD(PIPE, 0, 256, "PIPE current arithmetic"); FWD(PIPE, 2, 256, "PIPE reduced + hash"); FWD(PIPE, 1, 256, "PIPE reduced + mask bits"); }
  for (int occ = 1; occ <= 2; ++occ) { FWD(PIPE_SGB, 0, 256, "PIPE+sgb current arithmetic"); FWD(PIPE_SGB, 1, 256, "PIPE+sgb reduced + mask bits"); }
  return 0;
}
