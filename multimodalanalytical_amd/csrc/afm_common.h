// Shared device/host helpers for libafm_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "afm_hip.h"
// padded-row hint bookkeeping of the last call on this thread (afm_last_hint): 0 no hint given, 1 honoured, -1 given but ignored
extern "C" void afm_note_hint(int state);

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef _Float16 f16;                                       // AFM_F16: IEEE half, conversions round to nearest even (v_cvt_pk_f16_f32)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
// The single-pass MFMA kernels (afm_gemm_mfma_impl.h, afm_attn_mfma_impl.h) are written once against `e16`, a 16-bit floating
// element, and compiled twice: bf16 (v_mfma_*_bf16) and, with AFM_E16_F16 defined, fp16 (v_mfma_*_f16).  Same tiles, same LDS
// images, same instruction counts: the two formats differ only in the MFMA opcode and the conversion instructions.
#ifdef AFM_E16_F16
typedef f16 e16;
typedef f16x8 e16x8;
typedef f16x4 e16x4;
#define AFM_E16 AFM_F16
#define AFM_E16_NS afm_f16
#define AFM_E16_FN(name) name##_f16
#define AFM_E16_NAME "f16"
#else
typedef bf16 e16;
typedef bf16x8 e16x8;
typedef bf16x4 e16x4;
#define AFM_E16 AFM_BF16
#define AFM_E16_NS afm_bf16
#define AFM_E16_FN(name) name
#define AFM_E16_NAME "bf16"
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef f32x16 afm_f32x16_;
typedef f32x4 afm_f32x4_;
// (host pass: the builtins do not exist there, the bodies are never run)
#if defined(__HIP_DEVICE_COMPILE__)
#define AFM_MFMA_BODY(expr) return expr
#else
#define AFM_MFMA_BODY(expr) return c
#endif
__device__ __forceinline__ afm_f32x4_ mfma16(bf16x8 a, bf16x8 b, afm_f32x4_ c) { AFM_MFMA_BODY(__builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)); }
__device__ __forceinline__ afm_f32x4_ mfma16(f16x8 a, f16x8 b, afm_f32x4_ c) { AFM_MFMA_BODY(__builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)); }
__device__ __forceinline__ afm_f32x16_ mfma32_raw(bf16x8 a, bf16x8 b, afm_f32x16_ c) { AFM_MFMA_BODY(__builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)); }
__device__ __forceinline__ afm_f32x16_ mfma32_raw(f16x8 a, f16x8 b, afm_f32x16_ c) { AFM_MFMA_BODY(__builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)); }

#define AFM_WAVE 64

// Launch + check.  hipGetLastError() is per-thread and sticky: torch's own runtime calls (e.g. an
// event query answering hipErrorNotReady) leave stale codes behind, so clear it before launching
// and only then read it back.
#define AFM_LAUNCH(kern, grid, block, shm, st, ...)                        \
  do {                                                                     \
    (void)hipGetLastError();                                               \
    hipLaunchKernelGGL(kern, grid, block, shm, st, __VA_ARGS__);           \
    if (hipGetLastError() != hipSuccess) return AFM_ERR_LAUNCH;            \
  } while (0)

// hipFuncSetAttribute (the > 64 KiB dynamic LDS opt-in) is PER DEVICE: a process-wide `static bool` would leave the kernels of a
// second GPU driven by the same process at the default limit.  need() is true the first time it is called on each device.
#include <atomic>
struct AfmOncePerDevice {
  std::atomic<unsigned long long> mask{0};
  bool need() {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long b = 1ull << (d & 63);
    if (mask.load(std::memory_order_relaxed) & b) return false;
    mask.fetch_or(b, std::memory_order_relaxed);
    return true;
  }
};

// thread-local name of the kernel family last dispatched (afm_last_algo)
extern "C" void afm_set_last_algo(const char* name);

// ---------------------------------------------------------------- dtype access
template <typename T> __device__ __forceinline__ float ld_f32(const T* p, int64_t i);
template <> __device__ __forceinline__ float ld_f32<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_f32<bf16>(const bf16* p, int64_t i) { return (float)p[i]; }
template <> __device__ __forceinline__ float ld_f32<f16>(const f16* p, int64_t i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void st_f32(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void st_f32<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st_f32<bf16>(bf16* p, int64_t i, float v) { p[i] = (bf16)v; }
template <> __device__ __forceinline__ void st_f32<f16>(f16* p, int64_t i, float v) { p[i] = (f16)v; }

// ---------------------------------------------------------------- split bf16 pairs (AFM_BF16X2)
// value = hi + lo; hi(r, c) at base[r*ld + c], lo(r, c) at base[r*ld + ld/2 + c] (include/afm_hip.h).
// `x2` is a tag type: an x2* points at the hi plane.
struct x2 { bf16 v; };
__device__ __forceinline__ void afm_split(float v, bf16& hi, bf16& lo) { hi = (bf16)v; lo = (bf16)(v - (float)hi); }
template <typename T> struct RowMul { static constexpr int v = 1; };
template <> struct RowMul<x2> { static constexpr int v = 2; };   // a contiguous split-pair row is [hi(n) | lo(n)]
// element (r, c) of a matrix with row stride ld, any dtype
template <typename T> __device__ __forceinline__ float ld_rc(const T* p, int64_t r, int c, int ld);
template <> __device__ __forceinline__ float ld_rc<float>(const float* p, int64_t r, int c, int ld) { return p[r * ld + c]; }
template <> __device__ __forceinline__ float ld_rc<bf16>(const bf16* p, int64_t r, int c, int ld) { return (float)p[r * ld + c]; }
template <> __device__ __forceinline__ float ld_rc<f16>(const f16* p, int64_t r, int c, int ld) { return (float)p[r * ld + c]; }
template <> __device__ __forceinline__ float ld_rc<x2>(const x2* p, int64_t r, int c, int ld) {
  const bf16* q = (const bf16*)p + r * ld + c;
  return (float)q[0] + (float)q[ld >> 1];
}
template <typename T> __device__ __forceinline__ void st_rc(T* p, int64_t r, int c, int ld, float v);
template <> __device__ __forceinline__ void st_rc<float>(float* p, int64_t r, int c, int ld, float v) { p[r * ld + c] = v; }
template <> __device__ __forceinline__ void st_rc<bf16>(bf16* p, int64_t r, int c, int ld, float v) { p[r * ld + c] = (bf16)v; }
template <> __device__ __forceinline__ void st_rc<f16>(f16* p, int64_t r, int c, int ld, float v) { p[r * ld + c] = (f16)v; }
template <> __device__ __forceinline__ void st_rc<x2>(x2* p, int64_t r, int c, int ld, float v) {
  bf16* q = (bf16*)p + r * ld + c;
  bf16 hi, lo;
  afm_split(v, hi, lo);
  q[0] = hi; q[ld >> 1] = lo;
}
// two values -> packed hi pair and packed lo pair (element 0 in the low half): 6 vector instructions for 2 elements (the compiler's
// per-element form converts and packs each bf16 by itself: 8+).  Same rounding as afm_split (v_cvt_pk_bf16_f32 is RNE).  The inputs
// must be VALU results, not MFMA accumulators read in place (no hazard wait states around inline asm).
typedef uint32_t afm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void afm_split2(float v0, float v1, uint32_t& hi2, uint32_t& lo2) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi2) : "v"(v0), "v"(v1));
  const float l0 = v0 - __uint_as_float(hi2 << 16), l1 = v1 - __uint_as_float(hi2 & 0xffff0000u);
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo2) : "v"(l0), "v"(l1));
#else
  hi2 = lo2 = 0; (void)v0; (void)v1;
#endif
}
// 8 consecutive elements (16-byte accesses per plane)
__device__ __forceinline__ void afm_split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
  afm_u32x4 h, l;
#pragma unroll
  for (int k = 0; k < 4; ++k) { uint32_t a, b; afm_split2(x[2 * k], x[2 * k + 1], a, b); h[k] = a; l[k] = b; }
  hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);
}
// dtype dispatch of a templated launch: DT_SWITCH(code, T, stmt) runs stmt with T = float / bf16 / x2 / f16
#define AFM_DT_SWITCH(code, T, ...)                                        \
  do {                                                                     \
    if ((code) == AFM_F32) { typedef float T; __VA_ARGS__; }               \
    else if ((code) == AFM_BF16) { typedef bf16 T; __VA_ARGS__; }          \
    else if ((code) == AFM_BF16X2) { typedef x2 T; __VA_ARGS__; }          \
    else if ((code) == AFM_F16) { typedef f16 T; __VA_ARGS__; }            \
    else return AFM_ERR_ARG;                                               \
  } while (0)

// ---------------------------------------------------------------- dropout stream
// keep(i) = mix32(lo(i) ^ key ^ hi(i)*phi) >= thresh ; documented in DESIGN.md and
// re-implemented in tests (numpy) so parity tests run WITH dropout against the oracle.
struct DropDev {
  uint32_t key;
  uint32_t thresh;    // 0 => keep everything
  float scale;        // 1/(1-p)
  uint32_t thresh16;  // attention-probability stream: 16 random bits per element
  float scale16;      // 1/(1 - thresh16/65536)
};
// 32-bit mixer of the dropout stream: xorshift / 24-bit multiply-add rounds.  v_mad_u32_u24 issues at
// full rate on CDNA4 whereas v_mul_lo_u32 is quarter rate: the attention kernels evaluate this once
// per probability pair, where two 32-bit multiplies per hash cost as much as the tile's MFMAs.
// Avalanche on sequential counters measured equal to lowbias32 (every output bit flips with
// probability 0.49-0.51 per input bit; tools/hash_quality.py).
__host__ __device__ __forceinline__ uint32_t afm_mad24(uint32_t x, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(x, c) + x;
#else
  return (uint32_t)(((uint64_t)(x & 0xFFFFFFu) * (uint64_t)(c & 0xFFFFFFu)) + x);
#endif
}
__host__ __device__ __forceinline__ uint32_t afm_lowbias32(uint32_t x) {   // name kept: the stream's mixer
  x ^= x >> 16; x = afm_mad24(x, 0x7b352dU); x ^= x >> 13; x = afm_mad24(x, 0x6ca68bU); x ^= x >> 16;
  return x;
}
static inline DropDev afm_make_drop(const afm_dropout* d) {
  DropDev r;
  if (!d || d->p <= 0.f) { r.key = 0; r.thresh = 0; r.scale = 1.f; r.thresh16 = 0; r.scale16 = 1.f; return r; }
  double t = (double)d->p * 4294967296.0;
  r.thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
  r.scale = 1.0f / (1.0f - d->p);
  double t16 = (double)d->p * 65536.0 + 0.5;
  r.thresh16 = t16 >= 65535.0 ? 65535u : (uint32_t)t16;
  r.scale16 = (float)(1.0 / (1.0 - (double)r.thresh16 / 65536.0));
  uint32_t k = afm_lowbias32((uint32_t)d->seed ^ 0x9E3779B9u);
  k = afm_lowbias32(k ^ (uint32_t)(d->seed >> 32));
  k = afm_lowbias32(k ^ (d->site * 0x85EBCA6Bu + 0x1234567u));
  r.key = k;
  return r;
}
// Two-level hashing (round 4): the full mixer once per GROUP -- a score-matrix row for the attention stream (below), a block of 64
// consecutive elements for the element-wise stream -- and a four-instruction mix per element PAIR of the group (add of a pair stride,
// 24-bit multiply-add, xor-shift by 16, 24-bit multiply-add); the pair's two elements take the low / high 16 bits and compare them
// with thresh16.  Keep rate, serial correlations along and across groups and field uniformity measured equal to the full mixer per
// pair (tools/hash_quality.py); before, every element (pair, for attention) ran the eight-instruction mixer: 46 % of the attention
// forward's vector work, a quarter of the FFN-up epilogue's.  tests/dropmask.py restates both streams.
#define AFM_PAIR_STRIDE 0x9E3779u      // odd, 24 bits: pair index times stride is one v_mul_u32_u24 (Tk / 2 < 2^24)
__host__ __device__ __forceinline__ uint32_t afm_row_hash(const DropDev& d, uint64_t row) {
  return afm_lowbias32((uint32_t)row ^ d.key ^ ((uint32_t)(row >> 32) * 0x9E3779B1u));
}
__host__ __device__ __forceinline__ uint32_t afm_pair_mix(uint32_t x) {       // x = row hash + pair index * AFM_PAIR_STRIDE
  x = afm_mad24(x, 0x7b352dU); x ^= x >> 16;
  return afm_mad24(x, 0x6ca68bU);
}
__host__ __device__ __forceinline__ uint32_t afm_pair_offset(uint32_t pair) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(pair, AFM_PAIR_STRIDE);
#else
  return (uint32_t)((uint64_t)(pair & 0xFFFFFFu) * AFM_PAIR_STRIDE);
#endif
}
// Element-wise stream: element idx (row-major in the tensor the mask applies to) belongs to block idx >> 6, pair (idx & 63) >> 1.
// The keep probability is 1 - thresh16 / 65536 (within 2^-17 of 1 - p); kept values are scaled by 1 / (1 - p) as before.
__device__ __forceinline__ bool afm_keep(const DropDev& d, uint64_t idx) {
  const uint32_t h = afm_pair_mix(afm_row_hash(d, idx >> 6) + afm_pair_offset(((uint32_t)idx & 63u) >> 1));
  return ((idx & 1) ? (h >> 16) : (h & 0xFFFFu)) >= d.thresh16;
}
// N = 2, 4 or 8 consecutive elements from idx0 (a multiple of N): ONE block hash and N / 2 pair mixes; out[k] = scale or 0
template <int N>
__device__ __forceinline__ void afm_keep_scale(const DropDev& d, uint64_t idx0, float (&out)[N]) {
  static_assert(N == 2 || N == 4 || N == 8, "whole pairs inside one 64-element block");
  const uint32_t base = afm_row_hash(d, idx0 >> 6) + afm_pair_offset(((uint32_t)idx0 & 63u) >> 1);
#pragma unroll
  for (int j = 0; j < N / 2; ++j) {
    const uint32_t h = afm_pair_mix(base + (uint32_t)j * AFM_PAIR_STRIDE);
    out[2 * j] = (h & 0xFFFFu) >= d.thresh16 ? d.scale : 0.f;
    out[2 * j + 1] = (h >> 16) >= d.thresh16 ? d.scale : 0.f;
  }
}
// Attention probabilities (B*H*Tq*Tk of them per layer): the group is the score-matrix ROW (row = (b H + h) Tq + q: in the flash kernels
// one full mixer per lane and kernel), the pair (2i, 2i+1) of that row's keys; kept values are scaled by scale16.
__device__ __forceinline__ bool afm_keep16(const DropDev& d, uint64_t row, uint32_t key) {
  const uint32_t h = afm_pair_mix(afm_row_hash(d, row) + afm_pair_offset(key >> 1));
  return ((key & 1) ? (h >> 16) : (h & 0xFFFFu)) >= d.thresh16;
}
__device__ __forceinline__ float afm_drop16(const DropDev& d, uint64_t row, uint32_t key, float x) {
  if (d.thresh16 == 0) return x;
  return afm_keep16(d, row, key) ? x * d.scale16 : 0.f;
}
__device__ __forceinline__ float afm_drop(const DropDev& d, uint64_t idx, float x) {
  if (d.thresh == 0) return x;
  return afm_keep(d, idx) ? x * d.scale : 0.f;
}

// ---------------------------------------------------------------- math
// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 rounding level for GELU):
// one v_rcp, one v_exp and a 5-term Horner chain instead of libm's branchy ~40-instruction erff,
// which dominated the GEMM epilogues it is fused into.
__device__ __forceinline__ float afm_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float afm_gelu(float x) {
  return 0.5f * x * (1.0f + afm_erf(x * 0.70710678118654752440f));
}
// gelu(x) and gelu'(x) together: the exponential of the erf approximation IS exp(-x^2/2), the pdf term
__device__ __forceinline__ void afm_gelu_both(float x, float& g, float& gp) {
  const float z = x * 0.70710678118654752440f, az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * az * az);
  const float cdf = 0.5f * (1.0f + copysignf(fmaf(-p * t, e, 1.0f), z));
  g = x * cdf;
  gp = fmaf(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float afm_gelu_grad(float x) {
  // d/dx [x Phi(x)] = Phi(x) + x phi(x)
  const float cdf = 0.5f * (1.0f + afm_erf(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
  return cdf + x * pdf;
}

// The same three functions for epilogues whose OUTPUT is a 16-bit float (fp16 / bf16 modes): the results are
// rounded to 11 / 8 significant bits anyway.  Same form as Abramowitz & Stegun 7.1.26 with FOUR terms refitted (weighted minimax over
// |x| <= 9 of Phi(-|x|) = P(t) exp(-x^2 / 2), t = 1 / (1 + 0.27 |x|): |Phi error| <= 8.6e-7), the 1/sqrt(2) of the argument and the 0.5
// of Phi folded into the constants, and Phi = 0.5 + copysign(0.5 - P e, x) in three instructions: 17 issue slots for the pair instead
// of 20 (rcp and exp count two).  Measured in fp32 against fp64 over [-12, 12]: |g error| <= 2.5e-6, |g' error| <= 1.0e-6 -- a
// hundredth of an fp16 half-ulp at |g| = 1.  (The three-term form 7.1.25, one slot cheaper, is 2.6e-5 off: the same size as the fp16
// rounding of a typical activation, and the c2 logits error against the CPU reference went from 7.2e-4 to 8.1e-4 with it; rejected.)
// fp32 / split-pair outputs keep the 1.5e-7 forms above.
__device__ __forceinline__ void afm_gelu_both16(float x, float& g, float& gp) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.27f, fabsf(x), 1.0f));
  const float p = t * fmaf(fmaf(fmaf(0.42752638f, t, -0.30561537f), t, 0.30629668f), t, 0.07179219f);
  const float u = x * 0.84932180f;                                                            // sqrt(log2(e) / 2)
  const float e = __builtin_amdgcn_exp2f(-(u * u));                                           // exp(-x^2 / 2)
  const float cdf = 0.5f + copysignf(fmaf(-p, e, 0.5f), x);
  g = x * cdf;
  gp = fmaf(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float afm_gelu16(float x) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.27f, fabsf(x), 1.0f));
  const float p = t * fmaf(fmaf(fmaf(0.42752638f, t, -0.30561537f), t, 0.30629668f), t, 0.07179219f);
  const float u = x * 0.84932180f;
  return x * (0.5f + copysignf(fmaf(-p, __builtin_amdgcn_exp2f(-(u * u)), 0.5f), x));
}
__device__ __forceinline__ float afm_gelu_grad16(float x) { float g, gp; afm_gelu_both16(x, g, gp); return gp; }
// dispatch on the output width of an epilogue: 16-bit outputs take the cheaper forms
template <bool OUT16> __device__ __forceinline__ float afm_gelu_o(float x) { return OUT16 ? afm_gelu16(x) : afm_gelu(x); }
template <bool OUT16> __device__ __forceinline__ float afm_gelu_grad_o(float x) { return OUT16 ? afm_gelu_grad16(x) : afm_gelu_grad(x); }

// ---------------------------------------------------------------- reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
