// LayerNorm forward/backward for the fp32 residual stream (gfx950).  HBM-bound: one 64-lane
// wave owns a row, keeps it in registers (two-pass mean / centred variance like torch), and
// writes the normalised row once.  The forward fuses the embedding's positional add and the
// per-modality placement into the concatenated sequence; the backward fuses the residual
// gradient add.  dgamma/dbeta: per-block register partials -> workspace -> one reduce kernel
// (no atomics, deterministic).
#include "afm_common.h"

#define LN_MAXV 32  // up to d = 2048 held in registers (template NV = ceil(d/64) rounded up)

__device__ __forceinline__ int64_t ln_out_row(int64_t r, int64_t seg_len, int64_t seg_stride,
                                              int64_t off) {
  if (seg_len == 0) return r;
  const int64_t b = r / seg_len;
  return b * seg_stride + off + (r - b * seg_len);
}

template <typename TY, int NV>
__global__ void k_ln_fwd(const float* x, const float* __restrict__ gamma,
                         const float* __restrict__ beta, const float* __restrict__ pos,
                         TY* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                         int64_t rows, int d, int64_t seg_len, int64_t seg_stride, int64_t off,
                         float eps, const void* __restrict__ add, int add_bf16, float* x_sum) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const float inv_d = 1.0f / (float)d;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float* xr = x + r * (int64_t)d;
    float v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      v[i] = j < d ? xr[j] : 0.f;
      if (add && j < d) {   // fused residual add: the stream value is written back once
        v[i] += add_bf16 ? (float)((const bf16*)add)[r * (int64_t)d + j] : ((const float*)add)[r * (int64_t)d + j];
        x_sum[r * (int64_t)d + j] = v[i];
      }
      s += v[i];
    }
    const float mu = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      const float c = j < d ? v[i] - mu : 0.f;
      q += c * c;
    }
    const float rs = rsqrtf(wave_sum(q) * inv_d + eps);
    if (lane == 0) {
      if (mean) mean[r] = mu;
      if (rstd) rstd[r] = rs;
    }
    const int64_t orow = ln_out_row(r, seg_len, seg_stride, off);
    const int64_t prow = seg_len == 0 ? r : off + (r % seg_len);
    TY* yr = y + orow * (int64_t)d;
    const float* pr = pos ? pos + prow * (int64_t)d : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      if (j < d) {
        float o = (v[i] - mu) * rs * gamma[j] + beta[j];
        if (pr) o += pr[j];
        st_f32(yr, j, o);
      }
    }
  }
}

extern "C" int afm_layernorm_fwd(const afm_ln_shape* s, const float* x, const float* gamma,
                                 const float* beta, const float* pos, void* y, float* mean,
                                 float* rstd, const void* add, float* x_sum, void* stream) {
  if (!s || !x || !gamma || !beta || !y || s->rows < 0 || s->d <= 0) return AFM_ERR_ARG;
  if (add && (!x_sum || (s->add_dtype != AFM_F32 && s->add_dtype != AFM_BF16) || s->seg_len != 0)) return AFM_ERR_ARG;
  const int add_bf16 = s->add_dtype == AFM_BF16;
  if (s->d > 64 * LN_MAXV) return AFM_ERR_UNSUPPORTED;
  if (s->rows == 0) return AFM_OK;
  int64_t g = (s->rows + 3) / 4;
  if (g > 2048) g = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (s->y_dtype != AFM_F32 && s->y_dtype != AFM_BF16) return AFM_ERR_ARG;
#define LN_FWD(NV)                                                                                  \
  do {                                                                                              \
    if (s->y_dtype == AFM_F32)                                                                      \
      AFM_LAUNCH((k_ln_fwd<float, NV>), dim3((int)g), dim3(256), 0, st, x, gamma, beta, pos, \
                         (float*)y, mean, rstd, s->rows, s->d, s->seg_len, s->out_seg_stride,       \
                         s->out_off, s->eps, add, add_bf16, x_sum);                                   \
    else                                                                                            \
      AFM_LAUNCH((k_ln_fwd<bf16, NV>), dim3((int)g), dim3(256), 0, st, x, gamma, beta, pos,  \
                         (bf16*)y, mean, rstd, s->rows, s->d, s->seg_len, s->out_seg_stride,        \
                         s->out_off, s->eps, add, add_bf16, x_sum);                                   \
  } while (0)
  const int nv = (s->d + 63) / 64;
  if (nv <= 1) LN_FWD(1); else if (nv <= 2) LN_FWD(2); else if (nv <= 4) LN_FWD(4);
  else if (nv <= 8) LN_FWD(8); else if (nv <= 12) LN_FWD(12); else if (nv <= 16) LN_FWD(16);
  else LN_FWD(32);
#undef LN_FWD
  return AFM_OK;
}

static inline int ln_bwd_blocks(int64_t rows) {
  int64_t g = (rows + 3) / 4;
  if (g > 1024) g = 1024;  // 4 blocks per CU keep HBM busy; the partial rows are reduced by k_ln_bwd_reduce
  if (g < 1) g = 1;
  return (int)g;
}
extern "C" int64_t afm_layernorm_bwd_ws_floats(const afm_ln_shape* s) {
  if (!s) return 0;
  return (int64_t)ln_bwd_blocks(s->rows) * 2 * s->d;
}

template <typename TY, int NV>
__global__ void k_ln_bwd(const TY* __restrict__ dy, const float* __restrict__ x,
                         const float* __restrict__ gamma, const float* __restrict__ mean,
                         const float* __restrict__ rstd, const float* __restrict__ dres,
                         float* __restrict__ dx, float* __restrict__ partial, int64_t rows, int d,
                         int64_t seg_len, int64_t seg_stride, int64_t off, TY* __restrict__ dx_drop,
                         DropDev dd) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [4 waves][2][d]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const float inv_d = 1.0f / (float)d;
  float ag[NV], ab[NV], gm[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    ag[i] = 0.f; ab[i] = 0.f;
    gm[i] = j < d ? gamma[j] : 0.f;
  }
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float mu = mean[r], rs = rstd[r];
    const float* xr = x + r * (int64_t)d;
    const TY* dyr = dy + ln_out_row(r, seg_len, seg_stride, off) * (int64_t)d;
    float xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      if (j < d) {
        const float dyv = ld_f32(dyr, j);
        xh[i] = (xr[j] - mu) * rs;
        g[i] = dyv * gm[i];
        ag[i] += dyv * xh[i];
        ab[i] += dyv;
      } else {
        xh[i] = 0.f; g[i] = 0.f;
      }
      s1 += g[i];
      s2 += g[i] * xh[i];
    }
    const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
    float* dxr = dx + r * (int64_t)d;
    const float* drr = dres ? dres + r * (int64_t)d : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      if (j < d) {
        float o = rs * (g[i] - m1 - xh[i] * m2);
        if (drr) o += drr[j];
        dxr[j] = o;
        if (dx_drop) st_f32(dx_drop, r * (int64_t)d + j, afm_drop(dd, (uint64_t)r * (uint64_t)d + (uint64_t)j, o));
      }
    }
  }
  // block partials: wave w owns sm[w][0..1][d]
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    if (j < d) {
      sm[(w * 2 + 0) * d + j] = ag[i];
      sm[(w * 2 + 1) * d + j] = ab[i];
    }
  }
  __syncthreads();
  float* pb = partial + (int64_t)blockIdx.x * 2 * d;
  for (int j = threadIdx.x; j < 2 * d; j += blockDim.x) {
    pb[j] = sm[0 * 2 * d + j] + sm[1 * 2 * d + j] + sm[2 * 2 * d + j] + sm[3 * 2 * d + j];
  }
}

// 64 columns per block (lanes = consecutive columns: coalesced), the partial rows split over
// gridDim.y chunks and the 4 waves of a block; LDS-combine, then one fp32 atomic per column and chunk.
__global__ __launch_bounds__(256) void k_ln_bwd_reduce(const float* __restrict__ partial,
                                                       float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, int nblocks, int d) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
  float acc = 0.f;
  if (j < 2 * d)
    for (int b = b0 + w; b < b1; b += 4) acc += partial[(int64_t)b * 2 * d + j];
  sm[w][lane] = acc;
  __syncthreads();
  if (w == 0 && j < 2 * d) {
    const float t = sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane];
    if (j < d) { if (dgamma) atomicAdd(dgamma + j, t); }
    else if (dbeta) atomicAdd(dbeta + j - d, t);
  }
}

extern "C" int afm_layernorm_bwd(const afm_ln_shape* s, const void* dy, const float* x,
                                 const float* gamma, const float* mean, const float* rstd,
                                 const float* dres, float* dx, float* dgamma, float* dbeta,
                                 float* partial, void* dx_drop, const afm_dropout* drop, void* stream) {
  const DropDev dd = afm_make_drop(drop);
  if (!s || !dy || !x || !gamma || !mean || !rstd || !dx || !partial || s->rows < 0 || s->d <= 0)
    return AFM_ERR_ARG;
  if (s->d > 64 * LN_MAXV) return AFM_ERR_UNSUPPORTED;
  if (s->rows == 0) return AFM_OK;
  const int g = ln_bwd_blocks(s->rows);
  hipStream_t st = (hipStream_t)stream;
  const size_t shm = sizeof(float) * 8 * s->d;
  if (s->y_dtype != AFM_F32 && s->y_dtype != AFM_BF16) return AFM_ERR_ARG;
#define LN_BWD(NV)                                                                                   \
  do {                                                                                               \
    if (s->y_dtype == AFM_F32)                                                                       \
      AFM_LAUNCH((k_ln_bwd<float, NV>), dim3(g), dim3(256), shm, st, (const float*)dy, x,     \
                         gamma, mean, rstd, dres, dx, partial, s->rows, s->d, s->seg_len,            \
                         s->out_seg_stride, s->out_off, (float*)dx_drop, dd);                                             \
    else                                                                                             \
      AFM_LAUNCH((k_ln_bwd<bf16, NV>), dim3(g), dim3(256), shm, st, (const bf16*)dy, x, gamma, \
                         mean, rstd, dres, dx, partial, s->rows, s->d, s->seg_len,                   \
                         s->out_seg_stride, s->out_off, (bf16*)dx_drop, dd);                                             \
  } while (0)
  const int nv = (s->d + 63) / 64;
  if (nv <= 1) LN_BWD(1); else if (nv <= 2) LN_BWD(2); else if (nv <= 4) LN_BWD(4);
  else if (nv <= 8) LN_BWD(8); else if (nv <= 12) LN_BWD(12); else if (nv <= 16) LN_BWD(16);
  else LN_BWD(32);
#undef LN_BWD
  AFM_LAUNCH(k_ln_bwd_reduce, dim3((2 * s->d + 63) / 64, g >= 64 ? 16 : 1), dim3(256), 0, st, partial, dgamma,
                     dbeta, g, s->d);
  return AFM_OK;
}
