// Device-side PatchPreprocessor (reference data/preprocessing/patches.py:54-107): an HBM-bound
// elementwise map from raw spectra to standardised patches (+ gradient patches) and the patch mask.
// One thread per output element (coalesced 4-byte stores along a patch), a second small kernel for the
// per-patch mask.  Algorithmic bytes per spectrum: 4 L read + 4 P ps (+ P) written.
#include "afm_common.h"

struct PatchArgs {
  int B, L, ps, step, interp, deriv, masking, seq_first;
  int len;       // points after interpolation
  int n;         // whole patches of the trimmed spectrum
  int P0;        // spectrum patches (n, or the unfold count)
  int P;         // P0 + (deriv ? n : 0)
  float mean, std;
};

// raw fp32 value j of row b after the None -> zeros and interpolation steps
__device__ __forceinline__ float raw_at(const PatchArgs& a, const float* row, bool here, int j) {
  if (!here) return 0.f;
  if (!a.interp) return row[j];
  // scipy interp1d (linear): interval (lo, hi) = (j + 124, j + 125) of the old grid, x_new - x_lo = 2
  const int hi = j + 125 > 1 ? j + 125 : 1, lo = hi - 1;
  const double ylo = (double)row[lo], yhi = (double)row[hi];
  const double slope = (yhi - ylo) / 2.0;
  const double xn = 650.0 + 2.0 * j, xl = 400.0 + 2.0 * lo;
  return (float)(slope * (xn - xl) + ylo);
}

__global__ __launch_bounds__(256) void k_patch_values(PatchArgs a, const float* __restrict__ spectra,
                                                      const uint8_t* __restrict__ present, float* __restrict__ out) {
  // grid: x over the P * ps outputs of one spectrum, y (grid-stride) over spectra: 32-bit index math only
  const int per_row = a.P * a.ps;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= per_row) return;
  const int p = r / a.ps, k = r - p * a.ps;
  for (int b = blockIdx.y; b < a.B; b += gridDim.y) {
    const int64_t i = (int64_t)b * per_row + r;
    const float* row = spectra + (int64_t)b * a.L;
    const bool here = present == nullptr || present[b] != 0;
    float v;
    if (p < a.P0) {
      const int j = p * a.step + k;
      v = (raw_at(a, row, here, j) - a.mean) / a.std;
    } else {                      // torch.gradient of the raw spectrum, edge_order 1
      const int j = (p - a.P0) * a.ps + k;
      if (j == 0) v = raw_at(a, row, here, 1) - raw_at(a, row, here, 0);
      else if (j == a.len - 1) v = raw_at(a, row, here, j) - raw_at(a, row, here, j - 1);
      else v = (raw_at(a, row, here, j + 1) - raw_at(a, row, here, j - 1)) / 2.0f;
    }
    const int64_t o = a.seq_first ? ((int64_t)p * a.B + b) * a.ps + k : i;
    out[o] = v;
  }
}

__global__ __launch_bounds__(256) void k_patch_mask(PatchArgs a, const uint8_t* __restrict__ present,
                                                    const float* __restrict__ patches, uint8_t* __restrict__ mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (b, p)
  if (i >= a.B * a.P) return;
  const int b = i / a.P, p = i - b * a.P;
  const int64_t o = a.seq_first ? (int64_t)p * a.B + b : i;
  uint8_t m;
  if (a.masking) {
    const float* v = patches + o * a.ps;
    float s = 0.f;
    for (int k = 0; k < a.ps; ++k) s += v[k];
    m = s == 0.f;
  } else {
    m = !(present == nullptr || present[b] != 0);
  }
  mask[o] = m;
}

static bool make_args(const afm_patch_desc* d, PatchArgs& a) {
  if (!d || d->B <= 0 || d->L <= 1 || d->patch_size <= 0 || d->step <= 0 || d->step > d->patch_size) return false;
  if (d->interpolation && d->L < 1750) return false;     // needs old-grid points up to index 1749
  if (!(d->std > 0.0) && !(d->std < 0.0)) return false;
  a.B = d->B; a.L = d->L; a.ps = d->patch_size; a.step = d->step;
  a.interp = d->interpolation != 0; a.deriv = d->derivative != 0; a.masking = d->masking != 0;
  a.seq_first = d->seq_first != 0;
  a.len = a.interp ? 1625 : d->L;
  a.n = a.len / a.ps;
  if (a.n <= 0) return false;
  a.P0 = a.step == a.ps ? a.n : (a.n * a.ps - a.ps) / a.step + 1;
  a.P = a.P0 + (a.deriv ? a.n : 0);
  a.mean = (float)d->mean; a.std = (float)d->std;
  return true;
}

extern "C" int32_t afm_patch_count(const afm_patch_desc* d) {
  PatchArgs a;
  return make_args(d, a) ? a.P : -1;
}

extern "C" int afm_patch_preprocess(const afm_patch_desc* d, const float* spectra, const uint8_t* present,
                                    float* patches, uint8_t* mask, void* stream) {
  PatchArgs a;
  if (!make_args(d, a) || !spectra || !patches || !mask) return AFM_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int per_row = a.P * a.ps;
  AFM_LAUNCH(k_patch_values, dim3((per_row + 255) / 256, a.B < 4096 ? a.B : 4096), dim3(256), 0, st, a, spectra, present, patches);
  AFM_LAUNCH(k_patch_mask, dim3((a.B * a.P + 255) / 256), dim3(256), 0, st, a, present, patches, mask);
  return AFM_OK;
}


// ---------------------------------------------------------------- mixture generator (datasets.py:58-141)
// One workgroup per mixed spectrum; a thread keeps its <= 8 fp64 points in registers between the
// weighted average and the min-max normalisation (L <= 2048).
__global__ __launch_bounds__(256) void k_mix_spectra(const float* __restrict__ table, int64_t N, int L,
                                                     const int64_t* __restrict__ idx, int c, const double* __restrict__ ratio,
                                                     int normalize, int out_len, float* __restrict__ out) {
#pragma clang fp contract(off)   // numpy multiplies, then adds: no fused multiply-add in the weighted sum
  const int r = blockIdx.x, t = threadIdx.x;
  double wsum = 0.0;
  for (int k = 0; k < c; ++k) wsum += ratio[k];
  double v[8];
  double mn = INFINITY, mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int j = t + 256 * u;
    double acc = 0.0;
    if (j < L) {
      for (int k = 0; k < c; ++k) {
        int64_t row = idx[(int64_t)r * c + k];
        row = row < 0 ? 0 : (row >= N ? N - 1 : row);
        const double term = (double)table[row * L + j] * ratio[k];
        acc = k == 0 ? term : acc + term;
      }
      acc /= wsum;
      mn = fmin(mn, acc); mx = fmax(mx, acc);
    }
    v[u] = acc;
  }
  __shared__ double smn[4], smx[4];
  if (normalize) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o, 64)); mx = fmax(mx, __shfl_xor(mx, o, 64)); }
    if ((t & 63) == 0) { smn[t >> 6] = mn; smx[t >> 6] = mx; }
    __syncthreads();
    mn = fmin(fmin(smn[0], smn[1]), fmin(smn[2], smn[3]));
    mx = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
  }
  const double range = mx - mn;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int j = t + 256 * u;
    if (j >= out_len) continue;
    double x = j < L ? v[u] : 0.0;
    if (normalize && j < L) {
      x = x > 0.0 ? x : 0.0;                       // clipped AFTER min / max were taken (datasets.py:50-52)
      x = range == 0.0 ? 0.0 : (x - mn) / range;
    }
    out[(int64_t)r * out_len + j] = (float)x;
  }
}

extern "C" int afm_mix_spectra(const float* table, int64_t N, int32_t L, const int64_t* idx, int32_t n, int32_t c,
                               const double* ratio, int32_t normalize, int32_t out_len, float* out, void* stream) {
  if (!table || !idx || !ratio || !out || N <= 0 || L <= 0 || n <= 0 || c <= 0 || out_len < L) return AFM_ERR_ARG;
  if (L > 2048 || out_len > 2048) return AFM_ERR_UNSUPPORTED;
  AFM_LAUNCH(k_mix_spectra, dim3(n), dim3(256), 0, (hipStream_t)stream, table, N, L, idx, c, ratio, normalize, out_len, out);
  return AFM_OK;
}
