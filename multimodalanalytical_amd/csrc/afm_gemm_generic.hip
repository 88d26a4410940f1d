// Exact-fp32 GEMM (FMA path) with the full afm_gemm epilogue, any transpose / stride / dtype.
// This is the numerics yardstick of the library (fp32 products, fp32 accumulation in k order)
// and the path for shapes the MFMA kernels do not take (odd K, tiny N such as the SMILES
// vocabulary, patch sizes 75/125).  64x64 tile, 16-deep k-steps through LDS, 4x4 per thread.
#include "afm_common.h"

struct GemmArgs {
  int M, N, K;
  int64_t sam, sak;  // element strides of op(A)[m][k]
  int64_t sbk, sbn;  // element strides of op(B)[k][n]
  int ldc;
  int a_dt, b_dt, c_dt;      // AFM_F32 / AFM_BF16 / AFM_BF16X2 / AFM_F16
  int a_lo, b_lo, c_lo;      // lo-plane offsets of split-pair operands (ld / 2)
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  const void* residual;
  void* pre_act;
  int act, accumulate;
  int splits, kchunk;
  DropDev dd;
};

__device__ __forceinline__ float ld_any(const void* p, int dt, int64_t i, int lo) {
  if (dt == AFM_F32) return ((const float*)p)[i];
  if (dt == AFM_F16) return (float)((const f16*)p)[i];
  const float hi = (float)((const bf16*)p)[i];
  return dt == AFM_BF16X2 ? hi + (float)((const bf16*)p)[i + lo] : hi;
}
__device__ __forceinline__ void st_any(void* p, int dt, int64_t i, float v, int lo) {
  if (dt == AFM_F32) { ((float*)p)[i] = v; return; }
  if (dt == AFM_F16) { ((f16*)p)[i] = (f16)v; return; }
  bf16 hi, l2;
  afm_split(v, hi, l2);
  ((bf16*)p)[i] = hi;
  if (dt == AFM_BF16X2) ((bf16*)p)[i + lo] = l2;
}

#define GT 64
#define GK 16

__global__ __launch_bounds__(256) void k_gemm_generic(GemmArgs g) {
  __shared__ float As[GK][GT + 4];
  __shared__ float Bs[GK][GT + 4];
  const int t = threadIdx.x;
  const int tx = t & 15, ty = t >> 4;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int kbeg = blockIdx.z * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  const bool a_kfast = g.sak == 1;  // k contiguous in memory
  const bool b_kfast = g.sbk == 1;
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m, k;
      if (a_kfast) { k = t & 15; m = (t >> 4) + 16 * i; } else { m = t & 63; k = (t >> 6) + 4 * i; }
      const int gm = m0 + m, gk = k0 + k;
      As[k][m] = (gm < g.M && gk < kend) ? ld_any(g.A, g.a_dt, gm * g.sam + gk * g.sak, g.a_lo) : 0.f;
      int n, kb;
      if (b_kfast) { kb = t & 15; n = (t >> 4) + 16 * i; } else { n = t & 63; kb = (t >> 6) + 4 * i; }
      const int gn = n0 + n, gkb = k0 + kb;
      Bs[kb][n] = (gn < g.N && gkb < kend) ? ld_any(g.B, g.b_dt, gkb * g.sbk + gn * g.sbn, g.b_lo) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GK; ++k) {
      const f32x4 a = *(const f32x4*)&As[k][ty * 4];
      const f32x4 b = *(const f32x4*)&Bs[k][tx * 4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= g.N) continue;
      const int64_t ci = (int64_t)m * g.ldc + n;
      float v = acc[i][j];
      if (g.splits > 1) {  // split-K: fp32 atomics into a C the launcher prepared
        if (blockIdx.z == 0) {
          if (g.bias) v += g.bias[n];
          if (g.residual) v += ((const float*)g.residual)[ci];
        }
        atomicAdd((float*)g.C + ci, v);
        continue;
      }
      if (g.bias) v += g.bias[n];
      if (g.act == AFM_ACT_GELU_BWD) {
        v = afm_drop(g.dd, (uint64_t)m * (uint64_t)g.N + (uint64_t)n, v) * afm_gelu_grad(ld_any(g.pre_act, g.c_dt, ci, g.c_lo));
      } else if (g.act == AFM_ACT_GELU_SAVE_GRAD) {
        float y, yp;
        afm_gelu_both(v, y, yp);
        const float k = afm_drop(g.dd, (uint64_t)m * (uint64_t)g.N + (uint64_t)n, 1.0f);   // scale or 0
        st_any(g.pre_act, g.c_dt, ci, yp * k, g.c_lo);
        v = y * k;
      } else if (g.act == AFM_ACT_MUL_SAVED) {
        v *= ld_any(g.pre_act, g.c_dt, ci, g.c_lo);
      } else {
        if (g.pre_act) st_any(g.pre_act, g.c_dt, ci, v, g.c_lo);
        if (g.act == AFM_ACT_RELU) v = fmaxf(v, 0.f);
        else if (g.act == AFM_ACT_GELU) v = afm_gelu(v);
        v = afm_drop(g.dd, (uint64_t)m * (uint64_t)g.N + (uint64_t)n, v);
      }
      if (g.residual) v += ld_any(g.residual, g.c_dt, ci, g.c_lo);
      if (g.accumulate) v += ld_any(g.C, g.c_dt, ci, g.c_lo);
      st_any(g.C, g.c_dt, ci, v, g.c_lo);
    }
  }
}

// ---------------------------------------------------------------- skinny shapes of the patch embedders
// The IR-only workload embeds 992 two-point patches per sample: the "GEMM" is (B*992 x 2) @ (2 x 512),
// an HBM-bound outer product that writes 260 MB, and its weight gradient is a 512 x 2 reduction over
// 127 k rows.  The 64 x 64 tile kernel above runs both at < 1 TB/s; these two stream them.
//
// forward:  C[m][n] = act(sum_{k < K <= 8} A[m][k] B(k, n) + bias[n]),  fp32, same FMA order as k_gemm_generic
template <int KMAX>
__global__ __launch_bounds__(256) void k_gemm_skinny_k(GemmArgs g, int rows_per_block) {
  const int nq = g.N >> 2;                      // column quads
  const int cq = threadIdx.x % nq, rsub = threadIdx.x / nq, rstep = 256 / nq;
  float b[KMAX][4], bias4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      b[k][j] = k < g.K ? ((const float*)g.B)[k * g.sbk + (int64_t)(cq * 4 + j) * g.sbn] : 0.f;
  if (g.bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bias4[j] = g.bias[cq * 4 + j];
  }
  const int m0 = blockIdx.x * rows_per_block;
  for (int r = rsub; r < rows_per_block; r += rstep) {
    const int m = m0 + r;
    if (m >= g.M) break;
    const float* arow = (const float*)g.A + (int64_t)m * g.sam;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < g.K) {
        const float a = arow[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += a * b[k][j];
      }
    }
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = acc[j] + bias4[j];
      if (g.act == AFM_ACT_RELU) x = fmaxf(x, 0.f);
      v[j] = x;
    }
    *(f32x4*)((float*)g.C + (int64_t)m * g.ldc + cq * 4) = v;
  }
}

// wgrad:  C[m][n] += sum_k A[k][m] B[k][n]  (N <= 8),  optional a_colsum[m] += sum_k A[k][m];  fp32 atomics
template <int NMAX>
__global__ __launch_bounds__(256) void k_gemm_skinny_n(GemmArgs g, float* a_colsum, int rows_per_block) {
  __shared__ float bs[64][NMAX];
  const int k0 = blockIdx.x * rows_per_block;
  const int kend = min(g.K, k0 + rows_per_block);
  for (int mbase = blockIdx.y * 256; mbase < g.M; mbase += gridDim.y * 256) {
    const int m = mbase + threadIdx.x;
    float acc[NMAX], cs = 0.f;
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
    for (int kc = k0; kc < kend; kc += 64) {
      __syncthreads();
      for (int i = threadIdx.x; i < 64 * NMAX; i += 256) {
        const int kk = i / NMAX, n = i % NMAX;
        bs[kk][n] = (kc + kk < kend && n < g.N) ? ((const float*)g.B)[(int64_t)(kc + kk) * g.sbk + n * g.sbn] : 0.f;
      }
      __syncthreads();
      if (m < g.M) {
        const int lim = min(64, kend - kc);
        const float* ap = (const float*)g.A + (int64_t)kc * g.sak + m;
        int kk = 0;
        for (; kk + 8 <= lim; kk += 8) {      // eight independent loads in flight per thread
          float a[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] = ap[(int64_t)(kk + u) * g.sak];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            cs += a[u];
#pragma unroll
            for (int n = 0; n < NMAX; ++n) acc[n] += a[u] * bs[kk + u][n];
          }
        }
        for (; kk < lim; ++kk) {
          const float a = ap[(int64_t)kk * g.sak];
          cs += a;
#pragma unroll
          for (int n = 0; n < NMAX; ++n) acc[n] += a * bs[kk][n];
        }
      }
    }
    if (m < g.M) {
#pragma unroll
      for (int n = 0; n < NMAX; ++n)
        if (n < g.N) atomicAdd((float*)g.C + (int64_t)m * g.ldc + n, acc[n]);
      if (a_colsum) atomicAdd(a_colsum + m, cs);
    }
  }
}

// defined in afm_gemm_mfma.hip / afm_gemm_mfma_f16.hip (bf16 / fp16 operands); return AFM_ERR_UNSUPPORTED when the shape is not eligible
int afm_gemm_mfma_try(const afm_gemm_desc* d, hipStream_t st);
int afm_gemm_mfma_try_f16(const afm_gemm_desc* d, hipStream_t st);
bool afm_gemm_tn_group_eligible(const afm_gemm_desc* d);
bool afm_gemm_tn_group_eligible_f16(const afm_gemm_desc* d);
int afm_gemm_tn_group_launch(const afm_gemm_desc* const* ds, int count, hipStream_t st);
int afm_gemm_tn_group_launch_f16(const afm_gemm_desc* const* ds, int count, hipStream_t st);
// defined in afm_gemm_x3.hip: split-pair operands on three bf16 MFMAs per product
int afm_gemm_x3_try(const afm_gemm_desc* d, hipStream_t st);

static int gemm_generic(const afm_gemm_desc* d, hipStream_t st) {
  GemmArgs g;
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.sam = d->transA ? 1 : d->lda; g.sak = d->transA ? d->lda : 1;
  g.sbk = d->transB ? 1 : d->ldb; g.sbn = d->transB ? d->ldb : 1;
  g.ldc = d->ldc;
  g.a_dt = d->a_dtype; g.b_dt = d->b_dtype; g.c_dt = d->c_dtype;
  g.a_lo = d->lda / 2; g.b_lo = d->ldb / 2; g.c_lo = d->ldc / 2;
  g.A = d->A; g.B = d->B; g.C = d->C; g.bias = d->bias; g.residual = d->residual; g.pre_act = d->pre_act;
  g.act = d->act; g.accumulate = d->accumulate;
  g.dd = afm_make_drop(&d->drop);
  const bool all_f32 = d->a_dtype == AFM_F32 && d->b_dtype == AFM_F32 && d->c_dtype == AFM_F32;
  const bool plain = !d->residual && !d->pre_act && d->drop.p <= 0.f;
  if (all_f32 && plain && !d->transA && !d->accumulate && d->K <= 8 && d->M >= 4096 && (d->N & 3) == 0 && d->N <= 1024 && 256 % (d->N >> 2) == 0 &&
      (d->ldc & 3) == 0 && ((uintptr_t)d->C & 15) == 0 && (d->act == AFM_ACT_NONE || d->act == AFM_ACT_RELU)) {
    const int rows_per_block = 32;
    AFM_LAUNCH(k_gemm_skinny_k<8>, dim3((d->M + rows_per_block - 1) / rows_per_block), dim3(256), 0, st, g, rows_per_block);
    afm_set_last_algo("skinny_k");
    return AFM_OK;
  }
  if (all_f32 && plain && d->transA && !d->transB && d->N <= 8 && d->K >= 4096 && d->act == AFM_ACT_NONE && !d->bias) {
    if (!d->accumulate &&
        hipMemset2DAsync(d->C, sizeof(float) * d->ldc, 0, sizeof(float) * d->N, d->M, st) != hipSuccess)
      return AFM_ERR_LAUNCH;
    const int rows_per_block = 128;   // many small workgroups: the kernel streams A once and is latency-bound otherwise
    const int gy = d->M <= 2048 ? (d->M + 255) / 256 : 8;
    AFM_LAUNCH(k_gemm_skinny_n<8>, dim3((d->K + rows_per_block - 1) / rows_per_block, gy), dim3(256), 0, st, g, d->a_colsum,
               rows_per_block);
    afm_set_last_algo("skinny_n");
    return AFM_OK;
  }
  const int gx = (d->N + GT - 1) / GT, gy = (d->M + GT - 1) / GT;
  int splits = 1;
  const bool can_split = d->c_dtype == AFM_F32 && d->act == AFM_ACT_NONE && !d->pre_act && d->drop.p <= 0.f;
  if (can_split && (int64_t)gx * gy < 256 && d->K >= 2048) {
    splits = (int)((512 + (int64_t)gx * gy - 1) / ((int64_t)gx * gy));
    const int maxs = d->K / 512;
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
  }
  int kchunk = (d->K + splits - 1) / splits;
  kchunk = (kchunk + GK - 1) / GK * GK;
  splits = (d->K + kchunk - 1) / kchunk;
  g.splits = splits; g.kchunk = kchunk;
  if (splits > 1 && !d->accumulate) {
    // rows of C may be strided: clear row by row only if needed
    if (d->ldc == d->N) {
      if (hipMemsetAsync(d->C, 0, sizeof(float) * (size_t)d->M * d->N, st) != hipSuccess) return AFM_ERR_LAUNCH;
    } else {
      if (hipMemset2DAsync(d->C, sizeof(float) * d->ldc, 0, sizeof(float) * d->N, d->M, st) != hipSuccess)
        return AFM_ERR_LAUNCH;
    }
  }
  AFM_LAUNCH(k_gemm_generic, dim3(gx, gy, splits), dim3(256), 0, st, g);
  if (d->a_colsum) {  // bias gradient: column sums of A (K x M, row-major) through the colsum kernel
    const int r = afm_colsum(d->A, d->a_colsum, d->K, d->M, d->lda, d->a_dtype, 1, (void*)st);
    if (r != AFM_OK) return r;
  }
  afm_set_last_algo(splits > 1 ? "generic_splitk" : "generic");
  return AFM_OK;
}

extern "C" int afm_last_hint(void);
extern "C" int afm_gemm(const afm_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->C) return AFM_ERR_ARG;
  if (d->M < 0 || d->N < 0 || d->K < 0) return AFM_ERR_ARG;
  if (d->a_dtype < AFM_F32 || d->a_dtype > AFM_F16 || d->b_dtype < AFM_F32 || d->b_dtype > AFM_F16 ||
      d->c_dtype < AFM_F32 || d->c_dtype > AFM_F16) return AFM_ERR_ARG;
  if (d->act < AFM_ACT_NONE || d->act > AFM_ACT_GLU_BWD) return AFM_ERR_ARG;
  const bool glu = d->act >= AFM_ACT_GLU;
  if (d->act >= AFM_ACT_GELU_BWD && d->act != AFM_ACT_GLU && !d->pre_act) return AFM_ERR_ARG;
  if (glu && (d->glu_rows <= 0 || (d->glu_rows & 3) || d->transA || !d->transB ||
              d->N != (d->act == AFM_ACT_GLU_BWD ? d->glu_rows : 2 * d->glu_rows)))
    return AFM_ERR_ARG;
  if (d->act == AFM_ACT_MUL_SAVED && (d->bias || d->drop.p > 0.f)) return AFM_ERR_ARG;
  {  // split-pair rows hold two planes: ld >= 2 * width, even
    const int ca = d->a_dtype == AFM_BF16X2 ? 2 : 1, cb = d->b_dtype == AFM_BF16X2 ? 2 : 1, cc = d->c_dtype == AFM_BF16X2 ? 2 : 1;
    const int ncol_c = d->act == AFM_ACT_GLU_BWD ? 2 * d->N : (d->act == AFM_ACT_GLU || d->act == AFM_ACT_GLU_SAVE) ? d->N / 2 : d->N;
    if (d->ldc < cc * ncol_c || (cc == 2 && (d->ldc & 1))) return AFM_ERR_ARG;
    if (d->lda < ca * (d->transA ? d->M : d->K) || d->ldb < cb * (d->transB ? d->K : d->N)) return AFM_ERR_ARG;
    if ((ca == 2 && (d->lda & 1)) || (cb == 2 && (d->ldb & 1))) return AFM_ERR_ARG;
  }
  if (d->a_colsum && !d->transA) return AFM_ERR_ARG;
  afm_note_hint(d->k_live ? -1 : 0);      // (a kernel that takes the hint notes 1)
  if (d->M == 0 || d->N == 0) return AFM_OK;
  hipStream_t st = (hipStream_t)stream;
  if (d->algo != AFM_ALGO_GENERIC) {
    const int r = (d->a_dtype == AFM_BF16X2 && d->b_dtype == AFM_BF16X2) ? afm_gemm_x3_try(d, st)
                  : (d->a_dtype == AFM_F16 && d->b_dtype == AFM_F16) ? afm_gemm_mfma_try_f16(d, st) : afm_gemm_mfma_try(d, st);
    if (r != AFM_ERR_UNSUPPORTED) return r;
    if (d->algo == AFM_ALGO_MFMA) return AFM_ERR_UNSUPPORTED;
  }
  if (glu) return AFM_ERR_UNSUPPORTED;        // fused gated FFN: MFMA kernels only (the caller keeps afm_glu_fwd / afm_glu_bwd)
  if (d->glu_rows && d->transA) return AFM_ERR_UNSUPPORTED;
  return gemm_generic(d, st);
}

// count independent GEMMs with the result of calling afm_gemm on each.  Weight-gradient problems (TN, 16-bit operands of one
// type, fp32 accumulate into C, M, N >= 256, K a multiple of 64) are fused, up to 8 per launch, into one grid that shares ONE
// split-K budget; everything else goes through afm_gemm one by one, in order.  The problems must not alias each other's C.
extern "C" int afm_gemm_group(const afm_gemm_desc* descs, int32_t count, void* stream) {
  if (count < 0 || (count > 0 && !descs)) return AFM_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const afm_gemm_desc* grp[2][8];
  int ng[2] = {0, 0};
  int hint = 0;      // over all problems: -1 if any given hint was ignored, else 1 if any was honoured (afm_last_hint)
  auto fold = [&]() { const int s = afm_last_hint(); if (s < 0) hint = -1; else if (s > 0 && hint == 0) hint = 1; };
  auto flush = [&](int t) -> int {
    int r = AFM_OK;
    if (ng[t] == 1) { r = afm_gemm(grp[t][0], stream); fold(); }          // alone it keeps its own tile / split-K choice
    else if (ng[t] > 1) { r = t ? afm_gemm_tn_group_launch_f16(grp[t], ng[t], st) : afm_gemm_tn_group_launch(grp[t], ng[t], st); fold(); }
    ng[t] = 0;
    return r;
  };
  for (int i = 0; i < count; ++i) {
    const afm_gemm_desc* d = descs + i;
    if (!d->A || !d->B || !d->C || d->M <= 0 || d->N <= 0 || d->K < 0) { const int r = afm_gemm(d, stream); fold(); if (r != AFM_OK) return r; continue; }
    const int t = afm_gemm_tn_group_eligible_f16(d) ? 1 : afm_gemm_tn_group_eligible(d) ? 0 : -1;
    if (t < 0) { const int r = afm_gemm(d, stream); fold(); if (r != AFM_OK) return r; continue; }
    grp[t][ng[t]++] = d;
    if (ng[t] == 8) { const int r = flush(t); if (r != AFM_OK) return r; }
  }
  for (int t = 0; t < 2; ++t) { const int r = flush(t); if (r != AFM_OK) return r; }
  afm_note_hint(hint);
  return AFM_OK;
}
