// LM-head cross-entropy (ignore_index = -100) fused with the teacher-forced argmax.
// One 64-lane wave per row of logits; V is the SMILES vocabulary (tens to a few hundred), so a
// row is one or two loads per lane and the kernel is a single pass over the logits.
#include "afm_common.h"

__global__ __launch_bounds__(256) void k_ce_fwd(const float* __restrict__ logits,
                                                const int64_t* __restrict__ labels, int64_t rows,
                                                int V, int ld, float* __restrict__ row_lse,
                                                int64_t* __restrict__ argmax,
                                                float* __restrict__ stats) {
  __shared__ float s_loss[4], s_cnt[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + w;
  float loss = 0.f, cnt = 0.f;
  if (r < rows) {
    const float* lg = logits + r * (int64_t)ld;
    float mx = -INFINITY;
    int mi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
      const float x = lg[v];
      if (x > mx) { mx = x; mi = v; }  // strictly greater keeps the first index per lane
    }
    // wave arg-max: larger value wins, ties go to the smaller index (torch.argmax)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ox = __shfl_xor(mx, o, 64);
      const int oi = __shfl_xor(mi, o, 64);
      if (ox > mx || (ox == mx && oi < mi)) { mx = ox; mi = oi; }
    }
    float se = 0.f;
    for (int v = lane; v < V; v += 64) se += expf(lg[v] - mx);
    se = wave_sum(se);
    const float lse = mx + logf(se);
    if (lane == 0) {
      if (row_lse) row_lse[r] = lse;
      if (argmax) argmax[r] = mi;
      const int64_t lb = labels ? labels[r] : -100;
      if (lb != -100 && lb >= 0 && lb < V) { loss = lse - lg[lb]; cnt = 1.f; }
    }
  }
  if (lane == 0) { s_loss[w] = loss; s_cnt[w] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0 && stats) {
    const float c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (c > 0.f) {
      atomicAdd(stats + 0, s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3]);
      atomicAdd(stats + 1, c);
    }
  }
}

extern "C" int afm_ce_fwd(const float* logits, const int64_t* labels, int64_t rows, int32_t V,
                          int32_t ld, float* row_lse, int64_t* argmax, float* stats, void* stream) {
  if (!logits || rows < 0 || V <= 0 || ld < V) return AFM_ERR_ARG;
  if (rows == 0) return AFM_OK;
  AFM_LAUNCH(k_ce_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     logits, labels, rows, V, ld, row_lse, argmax, stats);
  return AFM_OK;
}

template <typename T>
__global__ void k_ce_bwd(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                         const float* __restrict__ row_lse, const float* __restrict__ stats,
                         float grad_scale, const float* __restrict__ scale_dev, T* __restrict__ dl, int lddl, int64_t rows, int V,
                         int ld) {
  const float cnt = stats[1];
  const float gs = cnt > 0.f ? grad_scale * (scale_dev ? scale_dev[0] : 1.f) / cnt : 0.f;
  const int ncol = lddl / RowMul<T>::v;     // every column of the row (padding beyond V is zeroed), per plane
  const int64_t total = rows * (int64_t)ncol;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ncol;
    const int v = (int)(i - r * ncol);
    float g = 0.f;
    const int64_t lb = labels[r];
    if (v < V && lb != -100 && lb >= 0 && lb < V) {
      g = expf(logits[r * (int64_t)ld + v] - row_lse[r]);
      if (v == lb) g -= 1.f;
      g *= gs;
    }
    st_rc(dl, r, v, lddl, g);
  }
}

extern "C" int afm_ce_bwd(const float* logits, const int64_t* labels, const float* row_lse,
                          const float* stats, float grad_scale, const float* scale_dev, void* dlogits, int32_t dl_dtype,
                          int32_t lddl, int64_t rows, int32_t V, int32_t ld, void* stream) {
  if (!logits || !labels || !row_lse || !stats || !dlogits || rows < 0 || V <= 0 || ld < V ||
      lddl < V * (dl_dtype == AFM_BF16X2 ? 2 : 1))
    return AFM_ERR_ARG;
  if (rows == 0) return AFM_OK;
  int64_t g = (rows * lddl + 255) / 256;
  if (g > 2048) g = 2048;
  hipStream_t st = (hipStream_t)stream;
  AFM_DT_SWITCH(dl_dtype, T, AFM_LAUNCH(k_ce_bwd<T>, dim3((int)g), dim3(256), 0, st, logits, labels, row_lse, stats,
                                        grad_scale, scale_dev, (T*)dlogits, lddl, rows, V, ld));
  return AFM_OK;
}
