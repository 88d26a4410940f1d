// Gradient-norm clip + Adam/AdamW over the model's single flat fp32 parameter buffer, and the dynamic loss scaler of the
// fp16 precision mode.  HBM-bound: sumsq reads g once (4 B/param); the step reads p,g,m,v and writes p,m,v (+ g zero,
// + 16-bit shadow) = 28-34 B/param.
#include "afm_common.h"

// Two stages, no atomics: every rank of a data-parallel job must derive the SAME clip coefficient from the same (all-reduced)
// gradient buffer, bit for bit, or the replicas drift apart by an ulp per step -- an atomic sum's order changes from run to run.
__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float sh[4];
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  const f32x4* g4 = (const f32x4*)g;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = g4[i];
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    acc += v * v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void k_sumsq_final(const float* __restrict__ part, int nblocks, float* __restrict__ out) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += part[i];      // fixed order per thread, fixed tree below
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] += (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

extern "C" int afm_sumsq(const float* g, int64_t n, float* out, float* partial, void* stream) {
  if (!g || !out || !partial || n < 0 || ((uintptr_t)g & 15)) return AFM_ERR_ARG;
  if (n == 0) return AFM_OK;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > AFM_SUMSQ_PARTIALS) blocks = AFM_SUMSQ_PARTIALS;
  if (blocks < 1) blocks = 1;
  AFM_LAUNCH(k_sumsq, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, g, n, partial);
  AFM_LAUNCH(k_sumsq_final, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, (int)blocks, out);
  return AFM_OK;
}

// scaler state (device, fp32): [0] loss scale S, [1] growth tracker, [2] effective optimiser steps taken, [3] skipped steps
template <typename TS>
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, int64_t n,
                                              const float* __restrict__ hyper,
                                              const float* __restrict__ sumsq,
                                              TS* __restrict__ p_lowp, int zero_grad, const float* __restrict__ scaler) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
  float bc1 = hyper[5], bc2 = hyper[6];
  const float max_norm = hyper[7];
  float gmult = hyper[8];
  const bool decoupled = hyper[9] != 0.f;
  bool skip = false;
  if (scaler) {
    // the gradient buffer holds S x the gradients; an inf / nan anywhere in it (fp16 overflow in the backward pass) skips the
    // step, as torch.amp.GradScaler.step does.  Bias corrections follow the steps actually TAKEN (torch's per-parameter `step`).
    gmult /= scaler[0];
    skip = sumsq && !isfinite(sumsq[0]);
    const float t = scaler[2] + 1.f;
    bc1 = 1.f - powf(b1, t);
    bc2 = 1.f - powf(b2, t);
  }
  float coef = gmult;
  if (max_norm > 0.f && sumsq) {
    const float c = max_norm / (sqrtf(sumsq[0]) * gmult + 1e-6f);
    coef *= fminf(c, 1.0f);
  }
  const float step = lr / bc1;
  const float inv_sqrt_bc2 = rsqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (skip) {
      if (zero_grad) g[i] = 0.f;
      continue;
    }
    float pi = p[i];
    float gi = g[i] * coef;
    if (wd != 0.f) {
      if (decoupled) pi *= 1.0f - lr * wd; else gi += wd * pi;
    }
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    pi -= step * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (zero_grad) g[i] = 0.f;
    if (p_lowp) p_lowp[i] = (TS)pi;
  }
}

extern "C" int afm_adam_step(float* p, float* g, float* m, float* v, int64_t n, const float* hyper,
                             const float* sumsq, void* p_lowp, int32_t lowp_dtype, int32_t zero_grad,
                             const float* scaler, void* stream) {
  if (!p || !g || !m || !v || !hyper || n < 0) return AFM_ERR_ARG;
  if (p_lowp && lowp_dtype != AFM_BF16 && lowp_dtype != AFM_F16) return AFM_ERR_ARG;
  if (n == 0) return AFM_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (p_lowp && lowp_dtype == AFM_F16)
    AFM_LAUNCH(k_adam<f16>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper, sumsq, (f16*)p_lowp,
               zero_grad, scaler);
  else
    AFM_LAUNCH(k_adam<bf16>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper, sumsq, (bf16*)p_lowp,
               zero_grad, scaler);
  return AFM_OK;
}

// torch.amp.GradScaler.update() on the device: after a skipped step S *= backoff and the growth tracker restarts; after
// `interval` consecutive good steps S *= growth.
__global__ void k_scaler_update(float* __restrict__ st, const float* __restrict__ sumsq, float growth, float backoff, float interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (!isfinite(sumsq[0])) {
    st[0] = fmaxf(st[0] * backoff, 1.0f);
    st[1] = 0.f;
    st[3] += 1.f;
  } else {
    st[2] += 1.f;
    st[1] += 1.f;
    if (st[1] >= interval) { st[0] = fminf(st[0] * growth, 16777216.f); st[1] = 0.f; }
  }
}
extern "C" int afm_scaler_update(float* state, const float* sumsq, float growth, float backoff, int32_t interval, void* stream) {
  if (!state || !sumsq || growth < 1.f || backoff <= 0.f || backoff > 1.f || interval <= 0) return AFM_ERR_ARG;
  AFM_LAUNCH(k_scaler_update, dim3(1), dim3(64), 0, (hipStream_t)stream, state, sumsq, growth, backoff, (float)interval);
  return AFM_OK;
}
