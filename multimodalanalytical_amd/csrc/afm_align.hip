// Encoder alignment head pieces (reference custom_modeling.py:453-475): masked mean pooling of the
// encoder output, its backward, and the sigmoid + mse / mae / sid loss with its gradient.  HBM-bound:
// the pooling reads B*S*d activations once (split over S with fp32 atomics so the chip is filled).
#include "afm_common.h"


// grid (d / 256, B, SPLIT): partial sums over an S-chunk, divided by the row's kept-token count
template <typename T>
__global__ __launch_bounds__(256) void k_masked_mean_fwd(const T* __restrict__ x, const uint8_t* __restrict__ key_pad,
                                                         int S, int d, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  const uint8_t* kp = key_pad + (int64_t)b * S;
  __shared__ int cnt_s;
  if (threadIdx.x == 0) cnt_s = 0;
  __syncthreads();
  int local = 0;
  for (int s = threadIdx.x; s < S; s += 256) local += kp[s] == 0;
  if (local) atomicAdd(&cnt_s, local);
  __syncthreads();
  const float inv = 1.0f / (float)cnt_s;        // 0 kept tokens: inf/nan, as the reference's 0/0
  const int chunk = (S + gridDim.z - 1) / gridDim.z;
  const int s0 = blockIdx.z * chunk, s1 = min(S, s0 + chunk);
  if (c >= d) return;
  float acc = 0.f;
  for (int s = s0; s < s1; ++s)
    if (kp[s] == 0) acc += ld_rc(x, (int64_t)b * S + s, c, d * RowMul<T>::v);
  atomicAdd(out + (int64_t)b * d + c, acc * inv);
}

__global__ __launch_bounds__(256) void k_masked_mean_bwd(const float* __restrict__ dy, const uint8_t* __restrict__ key_pad,
                                                         int S, int d, float* __restrict__ dx, int accumulate) {
  const int b = blockIdx.y;
  const uint8_t* kp = key_pad + (int64_t)b * S;
  __shared__ int cnt_s;
  if (threadIdx.x == 0) cnt_s = 0;
  __syncthreads();
  int local = 0;
  for (int s = threadIdx.x; s < S; s += 256) local += kp[s] == 0;
  if (local) atomicAdd(&cnt_s, local);
  __syncthreads();
  const float inv = 1.0f / (float)cnt_s;
  const int rows_per = (S + gridDim.x - 1) / gridDim.x;
  const int s0 = blockIdx.x * rows_per, s1 = min(S, s0 + rows_per);
  for (int s = s0; s < s1; ++s) {
    const bool keep = kp[s] == 0;
    float* o = dx + ((int64_t)b * S + s) * d;
    for (int c = threadIdx.x; c < d; c += 256) {
      const float g = keep ? dy[(int64_t)b * d + c] * inv : 0.f;
      o[c] = accumulate ? o[c] + g : g;
    }
  }
}

__global__ __launch_bounds__(256) void k_align_loss(const float* __restrict__ z, const float* __restrict__ target, int kind,
                                                    int B, int64_t total, float grad_scale, const float* __restrict__ scale_dev, float* __restrict__ stats,
                                                    float* __restrict__ dz) {
  const float inv_all = 1.0f / (float)total, inv_b = 1.0f / (float)B;
  float acc = 0.f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float p = 1.0f / (1.0f + expf(-z[i])), t = target[i];
    float l, g;
    if (kind == AFM_ALIGN_MSE) { const float e = p - t; l = e * e * inv_all; g = 2.f * e * inv_all; }
    else if (kind == AFM_ALIGN_MAE) { const float e = p - t; l = fabsf(e) * inv_all; g = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * inv_all; }
    else {
      const float pc = fmaxf(p, 1e-16f), tc = fmaxf(t, 1e-16f);
      const float lr = logf(pc / tc);
      l = (pc * lr - tc * lr) * inv_b;                           // p log(p/t) + t log(t/p)
      g = p > 1e-16f ? (lr + 1.f - tc / pc) * inv_b : 0.f;      // clamp passes the gradient only above eps
    }
    acc += l;
    if (dz) dz[i] = grad_scale * (scale_dev ? scale_dev[0] : 1.f) * g * p * (1.f - p);
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(stats, part[0] + part[1] + part[2] + part[3]);
}

extern "C" int afm_masked_mean_fwd(const void* x, int32_t x_dtype, const uint8_t* key_pad, int32_t B, int32_t S,
                                   int32_t d, float* out, void* stream) {
  if (!x || !key_pad || !out || B <= 0 || S <= 0 || d <= 0) return AFM_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * d, st) != hipSuccess) return AFM_ERR_LAUNCH;
  int split = 2048 / (((d + 255) / 256) * B);
  if (split < 1) split = 1;
  if (split > (S + 31) / 32) split = (S + 31) / 32;
  const dim3 grid((d + 255) / 256, B, split);
  AFM_DT_SWITCH(x_dtype, T, AFM_LAUNCH(k_masked_mean_fwd<T>, grid, dim3(256), 0, st, (const T*)x, key_pad, S, d, out));
  return AFM_OK;
}

extern "C" int afm_masked_mean_bwd(const float* dy, const uint8_t* key_pad, int32_t B, int32_t S, int32_t d,
                                   float* dx, int32_t accumulate, void* stream) {
  if (!dy || !key_pad || !dx || B <= 0 || S <= 0 || d <= 0) return AFM_ERR_ARG;
  int gx = 4096 / B;
  if (gx < 1) gx = 1;
  if (gx > S) gx = S;
  AFM_LAUNCH(k_masked_mean_bwd, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, dy, key_pad, S, d, dx, accumulate);
  return AFM_OK;
}

extern "C" int afm_align_loss(const float* z, const float* target, int32_t kind, int32_t B, int32_t n,
                              float grad_scale, const float* scale_dev, float* stats, float* dz, void* stream) {
  if (!z || !target || !stats || B <= 0 || n <= 0 || kind < 0 || kind > 2) return AFM_ERR_ARG;
  const int64_t total = (int64_t)B * n;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  AFM_LAUNCH(k_align_loss, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z, target, kind, B, total, grad_scale, scale_dev, stats, dz);
  return AFM_OK;
}
