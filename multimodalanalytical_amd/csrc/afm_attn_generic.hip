// Exact-fp32 masked attention (any head size), forward and backward, one 64-lane wave per
// query row (forward, dQ) or per key row (dK/dV).  Scores never reach HBM: a row of them lives
// in the wave's LDS slice.  This is the numerics yardstick for the MFMA attention kernels and
// the path for head sizes they do not take (the tiny test/plumbing configs: dh = 16).
#include "afm_common.h"

struct AttnArgs {
  int B, H, Tq, Tk, dh;
  int ldq, ldk, ldv, ldo;
  int64_t sqb, skb, svb, sob;   // batch strides (elements)
  int causal;
  float scale;
  const uint8_t* key_pad;
  DropDev dd;
};

// row-pointer accessors: element j of a row; `lo` = offset of the lo plane of a split-pair tensor (ld / 2)
template <typename T> __device__ __forceinline__ float ldp(const T* p, int j, int) { return (float)p[j]; }
template <> __device__ __forceinline__ float ldp<x2>(const x2* p, int j, int lo) { return (float)((const bf16*)p)[j] + (float)((const bf16*)p)[j + lo]; }
template <typename T> __device__ __forceinline__ void stp(T* p, int j, int, float v) { p[j] = (T)v; }
template <> __device__ __forceinline__ void stp<x2>(x2* p, int j, int lo, float v) {
  bf16 h, l;
  afm_split(v, h, l);
  ((bf16*)p)[j] = h; ((bf16*)p)[j + lo] = l;
}

__device__ __forceinline__ bool attn_masked(const AttnArgs& a, int b, int q, int k) {
  if (a.causal && k > q) return true;
  if (a.key_pad && a.key_pad[(int64_t)b * a.Tk + k]) return true;
  return false;
}

template <typename T>
__global__ __launch_bounds__(256) void k_attn_fwd_generic(AttnArgs a, const T* __restrict__ Q,
                                                          const T* __restrict__ K,
                                                          const T* __restrict__ V, T* __restrict__ O,
                                                          float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* qv = sm + (size_t)w * (a.dh + a.Tk);
  float* sc = qv + a.dh;
  const int64_t row = (int64_t)blockIdx.x * 4 + w;  // (b, h, q) flattened
  const int64_t nrows = (int64_t)a.B * a.H * a.Tq;
  if (row >= nrows) return;
  const int q = (int)(row % a.Tq);
  const int h = (int)((row / a.Tq) % a.H);
  const int b = (int)(row / ((int64_t)a.Tq * a.H));
  const T* qp = Q + ((int64_t)b * a.sqb + (int64_t)q * a.ldq) + (int64_t)h * a.dh;
  for (int j = lane; j < a.dh; j += 64) qv[j] = ldp(qp, j, a.ldq >> 1);
  // scores
  float mx = -INFINITY;
  for (int k = lane; k < a.Tk; k += 64) {
    float s = -INFINITY;
    if (!attn_masked(a, b, q, k)) {
      const T* kp = K + ((int64_t)b * a.skb + (int64_t)k * a.ldk) + (int64_t)h * a.dh;
      float acc = 0.f;
      for (int j = 0; j < a.dh; ++j) acc = fmaf(qv[j], ldp(kp, j, a.ldk >> 1), acc);
      s = acc * a.scale;
    }
    sc[k] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  T* op = O + ((int64_t)b * a.sob + (int64_t)q * a.ldo) + (int64_t)h * a.dh;
  if (mx == -INFINITY) {  // every key masked: zeros (torch _safe_softmax)
    for (int j = lane; j < a.dh; j += 64) stp(op, j, a.ldo >> 1, 0.f);
    if (lane == 0) lse[row] = INFINITY;
    return;
  }
  float l = 0.f;
  for (int k = lane; k < a.Tk; k += 64) {
    const float p = expf(sc[k] - mx);  // exp(-inf) = 0 for masked keys
    sc[k] = p;
    l += p;
  }
  l = wave_sum(l);
  if (lane == 0) lse[row] = mx + logf(l);
  const float inv_l = 1.0f / l;
  for (int j = lane; j < a.dh; j += 64) {
    float acc = 0.f;
    for (int k = 0; k < a.Tk; ++k) {
      const float p = afm_drop16(a.dd, (uint64_t)row, (uint32_t)k, sc[k]);
      acc = fmaf(p, ldp(V + ((int64_t)b * a.svb + (int64_t)k * a.ldv) + (int64_t)h * a.dh, j, a.ldv >> 1), acc);
    }
    stp(op, j, a.ldo >> 1, acc * inv_l);
  }
}

// dQ and delta: wave per query row
template <typename T>
__global__ __launch_bounds__(256) void k_attn_bwd_q_generic(
    AttnArgs a, const T* __restrict__ Q, const T* __restrict__ K, const T* __restrict__ V,
    const T* __restrict__ O, const T* __restrict__ dO, const float* __restrict__ lse,
    float* __restrict__ delta, T* __restrict__ dQ, int lddq) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* qv = sm + (size_t)w * (2 * a.dh + a.Tk);
  float* dov = qv + a.dh;
  float* ds = dov + a.dh;
  const int64_t row = (int64_t)blockIdx.x * 4 + w;
  const int64_t nrows = (int64_t)a.B * a.H * a.Tq;
  if (row >= nrows) return;
  const int q = (int)(row % a.Tq);
  const int h = (int)((row / a.Tq) % a.H);
  const int b = (int)(row / ((int64_t)a.Tq * a.H));
  const T* qp = Q + ((int64_t)b * a.sqb + (int64_t)q * a.ldq) + (int64_t)h * a.dh;
  const T* op = O + ((int64_t)b * a.sob + (int64_t)q * a.ldo) + (int64_t)h * a.dh;
  const T* dop = dO + ((int64_t)b * a.sob + (int64_t)q * a.ldo) + (int64_t)h * a.dh;
  float dl = 0.f;
  for (int j = lane; j < a.dh; j += 64) {
    qv[j] = ldp(qp, j, a.ldq >> 1);
    const float g = ldp(dop, j, a.ldo >> 1);
    dov[j] = g;
    dl += g * ldp(op, j, a.ldo >> 1);
  }
  dl = wave_sum(dl);
  if (lane == 0) delta[row] = dl;
  const float L = lse[row];
  for (int k = lane; k < a.Tk; k += 64) {
    float d = 0.f;
    if (!attn_masked(a, b, q, k) && L != INFINITY) {
      const T* kp = K + ((int64_t)b * a.skb + (int64_t)k * a.ldk) + (int64_t)h * a.dh;
      const T* vp = V + ((int64_t)b * a.svb + (int64_t)k * a.ldv) + (int64_t)h * a.dh;
      float s = 0.f, dp = 0.f;
      for (int j = 0; j < a.dh; ++j) {
        s = fmaf(qv[j], ldp(kp, j, a.ldk >> 1), s);
        dp = fmaf(dov[j], ldp(vp, j, a.ldv >> 1), dp);
      }
      const float p = expf(s * a.scale - L);
      dp = afm_drop16(a.dd, (uint64_t)row, (uint32_t)k, dp);
      d = p * (dp - dl) * a.scale;
    }
    ds[k] = d;
  }
  T* dqp = dQ + ((int64_t)b * a.Tq + q) * lddq + (int64_t)h * a.dh;
  for (int j = lane; j < a.dh; j += 64) {
    float acc = 0.f;
    for (int k = 0; k < a.Tk; ++k)
      acc = fmaf(ds[k], ldp(K + ((int64_t)b * a.skb + (int64_t)k * a.ldk) + (int64_t)h * a.dh, j, a.ldk >> 1), acc);
    stp(dqp, j, lddq >> 1, acc);
  }
}

// dK and dV: wave per key row
template <typename T>
__global__ __launch_bounds__(256) void k_attn_bwd_kv_generic(
    AttnArgs a, const T* __restrict__ Q, const T* __restrict__ K, const T* __restrict__ V,
    const T* __restrict__ dO, const float* __restrict__ lse, const float* __restrict__ delta,
    T* __restrict__ dK, T* __restrict__ dV, int lddk, int lddv) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* kv = sm + (size_t)w * (2 * a.dh + 2 * a.Tq);
  float* vv = kv + a.dh;
  float* pd = vv + a.dh;   // dropped probabilities (for dV)
  float* ds = pd + a.Tq;   // score gradients * scale (for dK)
  const int64_t row = (int64_t)blockIdx.x * 4 + w;  // (b, h, k)
  const int64_t nrows = (int64_t)a.B * a.H * a.Tk;
  if (row >= nrows) return;
  const int k = (int)(row % a.Tk);
  const int h = (int)((row / a.Tk) % a.H);
  const int b = (int)(row / ((int64_t)a.Tk * a.H));
  const T* kp = K + ((int64_t)b * a.skb + (int64_t)k * a.ldk) + (int64_t)h * a.dh;
  const T* vp = V + ((int64_t)b * a.svb + (int64_t)k * a.ldv) + (int64_t)h * a.dh;
  for (int j = lane; j < a.dh; j += 64) { kv[j] = ldp(kp, j, a.ldk >> 1); vv[j] = ldp(vp, j, a.ldv >> 1); }
  for (int q = lane; q < a.Tq; q += 64) {
    float pdv = 0.f, dsv = 0.f;
    const int64_t qrow = ((int64_t)b * a.H + h) * a.Tq + q;
    const float L = lse[qrow];
    if (!attn_masked(a, b, q, k) && L != INFINITY) {
      const T* qp = Q + ((int64_t)b * a.sqb + (int64_t)q * a.ldq) + (int64_t)h * a.dh;
      const T* dop = dO + ((int64_t)b * a.sob + (int64_t)q * a.ldo) + (int64_t)h * a.dh;
      float s = 0.f, dp = 0.f;
      for (int j = 0; j < a.dh; ++j) {
        s = fmaf(ldp(qp, j, a.ldq >> 1), kv[j], s);
        dp = fmaf(ldp(dop, j, a.ldo >> 1), vv[j], dp);
      }
      const float p = expf(s * a.scale - L);
      pdv = afm_drop16(a.dd, (uint64_t)qrow, (uint32_t)k, p);
      dp = afm_drop16(a.dd, (uint64_t)qrow, (uint32_t)k, dp);
      dsv = p * (dp - delta[qrow]) * a.scale;
    }
    pd[q] = pdv;
    ds[q] = dsv;
  }
  T* dkp = dK + ((int64_t)b * a.Tk + k) * lddk + (int64_t)h * a.dh;
  T* dvp = dV + ((int64_t)b * a.Tk + k) * lddv + (int64_t)h * a.dh;
  for (int j = lane; j < a.dh; j += 64) {
    float ak = 0.f, av = 0.f;
    for (int q = 0; q < a.Tq; ++q) {
      ak = fmaf(ds[q], ldp(Q + (int64_t)b * a.sqb + (int64_t)q * a.ldq + (int64_t)h * a.dh, j, a.ldq >> 1), ak);
      av = fmaf(pd[q], ldp(dO + (int64_t)b * a.sob + (int64_t)q * a.ldo + (int64_t)h * a.dh, j, a.ldo >> 1), av);
    }
    stp(dkp, j, lddk >> 1, ak);
    stp(dvp, j, lddv >> 1, av);
  }
}

// Keep-bit tensor (afm_attn_shape.drop_bits) from the dropout stream, for forwards that run the generic kernel: the layout the
// MFMA forward kernels emit, so a backward that takes the MFMA path finds valid bits whatever the forward dispatched to.
__global__ __launch_bounds__(256) void k_drop_bits(DropDev dd, int B, int H, int Tq, int Tk, int nq32, int nk32,
                                                   unsigned long long* __restrict__ bits) {
  const int lane = threadIdx.x & 63;
  const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nblk = (int64_t)B * H * nq32 * nk32;
  if (blk >= nblk) return;
  const int kb = (int)(blk % nk32), qb = (int)((blk / nk32) % nq32);
  const int64_t bh = blk / ((int64_t)nk32 * nq32);
  const int q = qb * 32 + (lane & 31);
  for (int r = 0; r < 16; ++r) {
    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const bool in = q < Tq && key < Tk;
    const bool keep = in && afm_keep16(dd, (uint64_t)bh * Tq + q, (uint32_t)key);
    const unsigned long long m = __ballot(keep);
    if (lane == 0) bits[blk * 16 + r] = m;
  }
}

static AttnArgs make_args(const afm_attn_shape* s) {
  AttnArgs a;
  a.B = s->B; a.H = s->H; a.Tq = s->Tq; a.Tk = s->Tk; a.dh = s->dh;
  a.ldq = s->ldq; a.ldk = s->ldk; a.ldv = s->ldv; a.ldo = s->ldo;
  a.sqb = s->sqb ? s->sqb : (int64_t)s->Tq * s->ldq; a.skb = s->skb ? s->skb : (int64_t)s->Tk * s->ldk;
  a.svb = s->svb ? s->svb : (int64_t)s->Tk * s->ldv; a.sob = s->sob ? s->sob : (int64_t)s->Tq * s->ldo;
  a.causal = s->causal; a.scale = s->scale; a.key_pad = s->key_pad;
  a.dd = afm_make_drop(&s->drop);
  return a;
}

static int check_shape(const afm_attn_shape* s) {
  if (!s || s->B <= 0 || s->H <= 0 || s->Tq <= 0 || s->Tk <= 0 || s->dh <= 0) return AFM_ERR_ARG;
  if (s->dtype < AFM_F32 || s->dtype > AFM_F16) return AFM_ERR_ARG;
  const int w = s->H * s->dh * (s->dtype == AFM_BF16X2 ? 2 : 1);   // split-pair rows hold two planes
  if (s->ldq < w || s->ldk < w || s->ldv < w || s->ldo < w) return AFM_ERR_ARG;
  return AFM_OK;
}

// defined in afm_attn_x3.hip (split-pair operands) and afm_attn_mfma.hip
int afm_attn_fwd_x3_try(const afm_attn_shape* s, const void* Q, const void* K, const void* V,
                        void* O, float* lse, hipStream_t st);
int afm_attn_bwd_x3_try(const afm_attn_shape* s, const void* Q, const void* K, const void* V,
                        const void* O, const void* dO, const float* lse, float* delta, void* dQ,
                        void* dK, void* dV, int lddq, int lddk, int lddv, hipStream_t st);
int afm_attn_fwd_mfma_try(const afm_attn_shape* s, const void* Q, const void* K, const void* V,
                          void* O, float* lse, hipStream_t st);
int afm_attn_bwd_mfma_try(const afm_attn_shape* s, const void* Q, const void* K, const void* V,
                          const void* O, const void* dO, const float* lse, float* delta, void* dQ,
                          void* dK, void* dV, int lddq, int lddk, int lddv, hipStream_t st);
// the same kernels compiled for fp16 operands (afm_attn_mfma_f16.hip)
int afm_attn_fwd_mfma_try_f16(const afm_attn_shape* s, const void* Q, const void* K, const void* V,
                              void* O, float* lse, hipStream_t st);
int afm_attn_bwd_mfma_try_f16(const afm_attn_shape* s, const void* Q, const void* K, const void* V,
                              const void* O, const void* dO, const float* lse, float* delta, void* dQ,
                              void* dK, void* dV, int lddq, int lddk, int lddv, hipStream_t st);

int afm_attn_bits_fill_try(const afm_attn_shape* s, hipStream_t st);
int afm_attn_bits_fill_try_f16(const afm_attn_shape* s, hipStream_t st);

extern "C" int afm_attn_drop_bits_fill(const afm_attn_shape* s, void* stream) {
  int r = check_shape(s);
  if (r != AFM_OK) return r;
  if (!s->drop_bits) return AFM_ERR_ARG;
  if (s->algo == AFM_ALGO_GENERIC) return AFM_ERR_UNSUPPORTED;
  if (s->dtype == AFM_F16) return afm_attn_bits_fill_try_f16(s, (hipStream_t)stream);
  if (s->dtype == AFM_BF16) return afm_attn_bits_fill_try(s, (hipStream_t)stream);
  return AFM_ERR_UNSUPPORTED;
}

extern "C" int afm_attn_fwd(const afm_attn_shape* s, const void* Q, const void* K, const void* V,
                            void* O, float* lse, void* stream) {
  int r = check_shape(s);
  if (r != AFM_OK) return r;
  if (!Q || !K || !V || !O || !lse) return AFM_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const bool packed = s->q_off || s->k_off;      // packed rows exist in the single-pass MFMA kernels only (include/afm_hip.h)
  if (s->algo != AFM_ALGO_GENERIC) {
    r = (s->dtype == AFM_BF16X2 && packed) ? AFM_ERR_UNSUPPORTED
        : s->dtype == AFM_BF16X2 ? afm_attn_fwd_x3_try(s, Q, K, V, O, lse, st)
        : s->dtype == AFM_F16 ? afm_attn_fwd_mfma_try_f16(s, Q, K, V, O, lse, st) : afm_attn_fwd_mfma_try(s, Q, K, V, O, lse, st);
    if (r != AFM_ERR_UNSUPPORTED) return r;
    if (s->algo == AFM_ALGO_MFMA) return r;
  }
  if (packed) return AFM_ERR_UNSUPPORTED;      // the kernels below know nothing of q_off / k_off: refuse, never compute on the wrong rows
  const AttnArgs a = make_args(s);
  const size_t shm = sizeof(float) * 4 * (size_t)(s->dh + s->Tk);
  if (shm > 160 * 1024) return AFM_ERR_UNSUPPORTED;
  const int64_t nrows = (int64_t)s->B * s->H * s->Tq;
  const dim3 grid((unsigned)((nrows + 3) / 4));
  AFM_DT_SWITCH(s->dtype, T,
    if (shm > 64 * 1024) hipFuncSetAttribute((const void*)k_attn_fwd_generic<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    AFM_LAUNCH(k_attn_fwd_generic<T>, grid, dim3(256), shm, st, a, (const T*)Q, (const T*)K, (const T*)V, (T*)O, lse));
  if (s->drop_bits && a.dd.thresh16) {
    const int nq32 = ((s->Tq + 127) / 128) * 4, nk32 = ((s->Tk + 63) / 64) * 2;
    const int64_t nblk = (int64_t)s->B * s->H * nq32 * nk32;
    AFM_LAUNCH(k_drop_bits, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, st, a.dd, s->B, s->H, s->Tq, s->Tk, nq32, nk32,
               (unsigned long long*)s->drop_bits);
  }
  afm_set_last_algo("attn_generic");
  return AFM_OK;
}

extern "C" int afm_attn_bwd(const afm_attn_shape* s, const void* Q, const void* K, const void* V,
                            const void* O, const void* dO, const float* lse, float* delta, void* dQ,
                            void* dK, void* dV, int32_t lddq, int32_t lddk, int32_t lddv,
                            void* stream) {
  int r = check_shape(s);
  if (r != AFM_OK) return r;
  if (!Q || !K || !V || !O || !dO || !lse || !delta || !dQ || !dK || !dV) return AFM_ERR_ARG;
  const int w = s->H * s->dh * (s->dtype == AFM_BF16X2 ? 2 : 1);
  if (lddq < w || lddk < w || lddv < w) return AFM_ERR_ARG;
  if (s->sqb || s->skb || s->svb || s->sob) return AFM_ERR_UNSUPPORTED;   // strided batches: forward (decode) only
  hipStream_t st = (hipStream_t)stream;
  const bool packed = s->q_off || s->k_off;
  if (s->algo != AFM_ALGO_GENERIC) {
    r = (s->dtype == AFM_BF16X2 && packed) ? AFM_ERR_UNSUPPORTED
        : s->dtype == AFM_BF16X2 ? afm_attn_bwd_x3_try(s, Q, K, V, O, dO, lse, delta, dQ, dK, dV, lddq, lddk, lddv, st)
        : s->dtype == AFM_F16 ? afm_attn_bwd_mfma_try_f16(s, Q, K, V, O, dO, lse, delta, dQ, dK, dV, lddq, lddk, lddv, st)
                              : afm_attn_bwd_mfma_try(s, Q, K, V, O, dO, lse, delta, dQ, dK, dV, lddq, lddk, lddv, st);
    if (r != AFM_ERR_UNSUPPORTED) return r;
    if (s->algo == AFM_ALGO_MFMA) return r;
  }
  if (packed) return AFM_ERR_UNSUPPORTED;      // (as in afm_attn_fwd)
  const AttnArgs a = make_args(s);
  const size_t shm_q = sizeof(float) * 4 * (size_t)(2 * s->dh + s->Tk);
  const size_t shm_k = sizeof(float) * 4 * (size_t)(2 * s->dh + 2 * s->Tq);
  if (shm_q > 160 * 1024 || shm_k > 160 * 1024) return AFM_ERR_UNSUPPORTED;
  const dim3 gq((unsigned)(((int64_t)s->B * s->H * s->Tq + 3) / 4));
  const dim3 gk((unsigned)(((int64_t)s->B * s->H * s->Tk + 3) / 4));
#define LAUNCH_BWD(T)                                                                                 \
  do {                                                                                                \
    if (shm_q > 64 * 1024) hipFuncSetAttribute((const void*)k_attn_bwd_q_generic<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_q); \
    if (shm_k > 64 * 1024) hipFuncSetAttribute((const void*)k_attn_bwd_kv_generic<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_k); \
    AFM_LAUNCH(k_attn_bwd_q_generic<T>, gq, dim3(256), shm_q, st, a, (const T*)Q, (const T*)K, \
                       (const T*)V, (const T*)O, (const T*)dO, lse, delta, (T*)dQ, lddq);              \
    AFM_LAUNCH(k_attn_bwd_kv_generic<T>, gk, dim3(256), shm_k, st, a, (const T*)Q, (const T*)K, \
                       (const T*)V, (const T*)dO, lse, delta, (T*)dK, (T*)dV, lddk, lddv);             \
  } while (0)
  if (s->dtype == AFM_F32) LAUNCH_BWD(float); else if (s->dtype == AFM_BF16) LAUNCH_BWD(bf16); else if (s->dtype == AFM_F16) LAUNCH_BWD(f16); else LAUNCH_BWD(x2);
#undef LAUNCH_BWD
  afm_set_last_algo("attn_generic");
  return AFM_OK;
}
