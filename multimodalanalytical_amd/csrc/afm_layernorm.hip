// LayerNorm forward/backward for the fp32 residual stream (gfx950).  HBM-bound: one 64-lane
// wave owns a row, keeps it in registers (two-pass mean / centred variance like torch), and
// writes the normalised row once.  The forward fuses the embedding's positional add and the
// per-modality placement into the concatenated sequence; the backward fuses the residual
// gradient add.  dgamma/dbeta: per-block register partials -> LDS -> fp32 atomics (vectorised kernels) or a workspace and a
// reduce kernel (row-per-wave scalar kernels).
#include <algorithm>
#include "afm_common.h"

#define LN_MAXV 32  // up to d = 2048 held in registers (template NV = ceil(d/64) rounded up)

// `map` (afm_ln_shape.row_map, nullable): position p of sample b's concatenated sequence lives in row map[b * seg_stride + p]
__device__ __forceinline__ int64_t ln_out_row(int64_t r, int64_t seg_len, int64_t seg_stride,
                                              int64_t off, const int32_t* __restrict__ map = nullptr) {
  if (seg_len == 0) return r;
  const int64_t b = r / seg_len;
  const int64_t p = b * seg_stride + off + (r - b * seg_len);
  return map ? (int64_t)map[p] : p;
}

template <typename TY, int NV>
__global__ void k_ln_fwd(const float* x, const float* __restrict__ gamma,
                         const float* __restrict__ beta, const float* __restrict__ pos,
                         TY* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                         int64_t rows, int d, int64_t seg_len, int64_t seg_stride, int64_t off,
                         float eps, const void* __restrict__ add, int add_dtype, float* x_sum, DropDev adrop,
                         const int32_t* __restrict__ map) {
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
        const float a = add_dtype == AFM_BF16 ? ld_rc((const bf16*)add, r, j, d)
                        : add_dtype == AFM_F16 ? ld_rc((const f16*)add, r, j, d)
                        : add_dtype == AFM_BF16X2 ? ld_rc((const x2*)add, r, j, 2 * d) : ld_rc((const float*)add, r, j, d);
        v[i] += afm_drop(adrop, (uint64_t)r * (uint64_t)d + (uint64_t)j, a);   // branch dropout rides on the add
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
    const int64_t orow = ln_out_row(r, seg_len, seg_stride, off, map);
    const int64_t prow = seg_len == 0 ? r : off + (r % seg_len);
    const float* pr = pos ? pos + prow * (int64_t)d : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      if (j < d) {
        float o = (v[i] - mu) * rs * gamma[j] + beta[j];
        if (pr) o += pr[j];
        st_rc(y, orow, j, d * RowMul<TY>::v, o);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------- vectorised
// Identity-mapped rows (the transformer-block LayerNorms): every lane owns 8 consecutive features
// per 512-wide slab, so all traffic is 16 bytes per lane on whole lines (fp32 as 2 x float4, bf16 as
// one bf16x8): the scalar kernels above reached only ~2.3 TB/s on these 2-4-byte accesses.
struct F8 { f32x4 lo, hi; };
__device__ __forceinline__ F8 ld8(const float* p) { return {*(const f32x4*)p, *(const f32x4*)(p + 4)}; }
__device__ __forceinline__ F8 ld8(const bf16* p) {
  const bf16x8 v = *(const bf16x8*)p;
  return {(f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, (f32x4){(float)v[4], (float)v[5], (float)v[6], (float)v[7]}};
}
__device__ __forceinline__ F8 ld8(const f16* p) {
  const f16x8 v = *(const f16x8*)p;
  return {(f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, (f32x4){(float)v[4], (float)v[5], (float)v[6], (float)v[7]}};
}
__device__ __forceinline__ void st8(float* p, const F8& v) { *(f32x4*)p = v.lo; *(f32x4*)(p + 4) = v.hi; }
__device__ __forceinline__ void st8(f16* p, const F8& v) {
  f16x8 o = {(f16)v.lo[0], (f16)v.lo[1], (f16)v.lo[2], (f16)v.lo[3], (f16)v.hi[0], (f16)v.hi[1], (f16)v.hi[2], (f16)v.hi[3]};
  *(f16x8*)p = o;
}
__device__ __forceinline__ void st8(bf16* p, const F8& v) {
  bf16x8 o = {(bf16)v.lo[0], (bf16)v.lo[1], (bf16)v.lo[2], (bf16)v.lo[3], (bf16)v.hi[0], (bf16)v.hi[1], (bf16)v.hi[2], (bf16)v.hi[3]};
  *(bf16x8*)p = o;
}
// streaming (write-once) forms: nontemporal stores keep the written lines from displacing the read streams in L2
__device__ __forceinline__ void st8s(float* p, const F8& v) {
  __builtin_nontemporal_store(v.lo, (f32x4*)p); __builtin_nontemporal_store(v.hi, (f32x4*)(p + 4));
}
__device__ __forceinline__ void st8s(f16* p, const F8& v, int) {
  f16x8 o = {(f16)v.lo[0], (f16)v.lo[1], (f16)v.lo[2], (f16)v.lo[3], (f16)v.hi[0], (f16)v.hi[1], (f16)v.hi[2], (f16)v.hi[3]};
  __builtin_nontemporal_store(o, (f16x8*)p);
}
__device__ __forceinline__ void st8s(bf16* p, const F8& v, int) {
  bf16x8 o = {(bf16)v.lo[0], (bf16)v.lo[1], (bf16)v.lo[2], (bf16)v.lo[3], (bf16)v.hi[0], (bf16)v.hi[1], (bf16)v.hi[2], (bf16)v.hi[3]};
  __builtin_nontemporal_store(o, (bf16x8*)p);
}
__device__ __forceinline__ void st8s(float* p, const F8& v, int) { st8s(p, v); }
// the same with the lo-plane offset of the split-pair dtype as second argument (ignored by plain dtypes)
__device__ __forceinline__ F8 ld8(const float* p, int) { return ld8(p); }
__device__ __forceinline__ F8 ld8(const bf16* p, int) { return ld8(p); }
__device__ __forceinline__ F8 ld8(const f16* p, int) { return ld8(p); }
__device__ __forceinline__ F8 ld8(const x2* p, int lo) {
  const F8 a = ld8((const bf16*)p), b = ld8((const bf16*)p + lo);
  return {a.lo + b.lo, a.hi + b.hi};
}
__device__ __forceinline__ void st8(float* p, const F8& v, int) { st8(p, v); }
__device__ __forceinline__ void st8(bf16* p, const F8& v, int) { st8(p, v); }
__device__ __forceinline__ void st8(f16* p, const F8& v, int) { st8(p, v); }
__device__ __forceinline__ void st8(x2* p, const F8& v, int lo) {
  const float x[8] = {v.lo[0], v.lo[1], v.lo[2], v.lo[3], v.hi[0], v.hi[1], v.hi[2], v.hi[3]};
  bf16x8 h, l;
  afm_split8(x, h, l);
  *(bf16x8*)p = h;
  *(bf16x8*)((bf16*)p + lo) = l;
}
__device__ __forceinline__ void st8s(x2* p, const F8& v, int lo) { st8(p, v, lo); }
__device__ __forceinline__ float hsum8(const F8& v) { return (v.lo[0] + v.lo[1]) + (v.lo[2] + v.lo[3]) + (v.hi[0] + v.hi[1]) + (v.hi[2] + v.hi[3]); }

template <typename TY, typename TA, int NC>
__global__ __launch_bounds__(256) void k_ln_fwd_vec(const float* x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, TY* __restrict__ y,
                                                    float* __restrict__ mean, float* __restrict__ rstd,
                                                    int64_t rows, int d, float eps, const TA* __restrict__ add,
                                                    float* x_sum, DropDev adrop, const float* __restrict__ pos,
                                                    int64_t seg_len, int64_t seg_stride, int64_t off,
                                                    const int32_t* __restrict__ map, const uint8_t* __restrict__ row_live, int nofill) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const float inv_d = 1.0f / (float)d;
  F8 gm[NC], bt[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < d) { gm[i] = ld8(gamma + c); bt[i] = ld8(beta + c); }
  }
  // Two rows in flight per wave: row r + nwaves is requested before row r is reduced and stored (one row at a time left the kernel at
  // 4.5 TB/s of the ~6.3 a streaming kernel reaches: every wave spent a full memory latency per row with nothing outstanding).
  // row_live (forward sense, identity row mapping): rows of all-padding blocks are neither loaded nor normalised -- zeros go to y, x_sum,
  // mean and rstd; the flags run one iteration ahead of the loads, as in the backward kernel below
  F8 cx[NC], ca[NC];
  auto live_of = [&](int64_t r) -> bool { return !row_live || row_live[r >> 6] != 0; };
  auto load_row = [&](int64_t r, F8 (&vx)[NC], F8 (&va)[NC], bool live) {
    if (!live) return;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        vx[i] = ld8(x + r * (int64_t)d + c);
        if (add) va[i] = ld8(add + r * (int64_t)(d * RowMul<TA>::v) + c, d);
      }
    }
  };
  bool lv_cur = wave < rows ? live_of(wave) : true;
  bool lv_nxt = wave < rows ? live_of(wave + nwaves < rows ? wave + nwaves : wave) : true;
  if (wave < rows) load_row(wave, cx, ca, lv_cur);
  for (int64_t r = wave; r < rows; r += nwaves) {
    F8 nx[NC], na[NC];
    const int64_t rn = r + nwaves < rows ? r + nwaves : r;
    const bool lv_nn = live_of(rn + nwaves < rows ? rn + nwaves : rn);
    load_row(rn, nx, na, lv_nxt);      // (the last row of a wave is requested twice: no branch around the loads)
    const bool dead = !lv_cur;
    lv_cur = lv_nxt; lv_nxt = lv_nn;
    if (dead) {
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const F8 z = {z4, z4};
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < d && !nofill) {      // (nofill: the caller's buffers already hold finite values in these rows, nothing is written)
          if (add) st8s(x_sum + r * (int64_t)d + c, z);
          st8s(y + r * (int64_t)(d * RowMul<TY>::v) + c, z, d);
        }
        cx[i] = nx[i]; ca[i] = na[i];
      }
      if (lane == 0 && !nofill) {
        if (mean) mean[r] = 0.f;
        if (rstd) rstd[r] = 0.f;
      }
      continue;
    }
    F8 v[NC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        v[i] = cx[i];
        if (add) {
          F8 a = ca[i];
          if (adrop.thresh) {   // wave-uniform: dropout of the residual branch, index row-major in `add`
            const uint64_t base = (uint64_t)r * (uint64_t)d + (uint64_t)c;
            float kp[8];      // (one block hash and four pair mixes for the lane's eight elements)
            afm_keep_scale<8>(adrop, base, kp);
#pragma unroll
            for (int k = 0; k < 4; ++k) { a.lo[k] *= kp[k]; a.hi[k] *= kp[4 + k]; }
          }
          v[i].lo += a.lo; v[i].hi += a.hi;
          st8s(x_sum + r * (int64_t)d + c, v[i]);
        }
        s += hsum8(v[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) { cx[i] = nx[i]; ca[i] = na[i]; }
    const float mu = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        v[i].lo -= mu; v[i].hi -= mu;
        F8 sq = {v[i].lo * v[i].lo, v[i].hi * v[i].hi};
        q += hsum8(sq);
      }
    }
    const float rs = rsqrtf(wave_sum(q) * inv_d + eps);
    if (lane == 0) {
      if (mean) mean[r] = mu;
      if (rstd) rstd[r] = rs;
    }
    // the embedder's form: row r of modality rows lands in row orow of the concatenated sequence, plus its positional row
    const int64_t orow = ln_out_row(r, seg_len, seg_stride, off, map);
    const float* pr = pos ? pos + (seg_len == 0 ? r : off + (r % seg_len)) * (int64_t)d : nullptr;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        F8 o = {v[i].lo * rs * gm[i].lo + bt[i].lo, v[i].hi * rs * gm[i].hi + bt[i].hi};
        if (pr) { const F8 pv = ld8(pr + c); o.lo += pv.lo; o.hi += pv.hi; }
        st8s(y + orow * (int64_t)(d * RowMul<TY>::v) + c, o, d);
      }
    }
  }
}

template <typename TY, int NC>
__global__ __launch_bounds__(256) void k_ln_bwd_vec(const TY* __restrict__ dy, const float* __restrict__ x,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ mean,
                                                    const float* __restrict__ rstd,
                                                    const float* __restrict__ dres, float* __restrict__ dx,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int d,
                                                    TY* __restrict__ dx_drop, DropDev dd, int64_t seg_len, int64_t seg_stride,
                                                    int64_t off, const uint8_t* __restrict__ row_live, const int32_t* __restrict__ map, int nofill) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [4 waves][2][d]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const float inv_d = 1.0f / (float)d;
  F8 gm[NC], ag[NC], ab[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = (lane + 64 * i) * 8;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    ag[i] = {z, z}; ab[i] = {z, z}; gm[i] = {z, z};
    if (c < d) gm[i] = ld8(gamma + c);
  }
  // Two rows in flight per wave, as in the forward kernel: dy, x, dres, mean and rstd of row r + nwaves are requested before row r is
  // reduced (the residual-gradient load used to sit behind the two wave reductions).  Rows in all-padding blocks are neither loaded nor
  // computed, as before.
  // The padded-row flags run ONE MORE iteration ahead: whether row r + nwaves is loaded at all depends on its block's byte, and read
  // where it was needed that byte was a dependent round trip to L2 in front of every row's loads -- with hints the kernel moved 4.2 TB/s
  // in the c3 step against 5.2 without them (profiles/r05_c3_fp16_step_pmc.json before / after).
  struct RowIn { F8 dy[NC], x[NC], dr[NC]; float mu, rs; };
  auto live_of = [&](int64_t r) -> bool { return !row_live || row_live[r >> 6] != 0; };
  auto load_row = [&](int64_t r, RowIn& in, bool live) {
    if (!live) return;
    in.mu = mean[r]; in.rs = rstd[r];
    const int64_t dyrow = ln_out_row(r, seg_len, seg_stride, off, map);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        in.dy[i] = ld8(dy + dyrow * (int64_t)(d * RowMul<TY>::v) + c, d);
        in.x[i] = ld8(x + r * (int64_t)d + c);
        if (dres) in.dr[i] = ld8(dres + r * (int64_t)d + c);
      }
    }
  };
  RowIn cur, nxt;
  bool lv_cur = wave < rows ? live_of(wave) : true;
  bool lv_nxt = wave < rows ? live_of(wave + nwaves < rows ? wave + nwaves : wave) : true;
  if (wave < rows) load_row(wave, cur, lv_cur);
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t rn = r + nwaves < rows ? r + nwaves : r;
    const bool lv_nn = live_of(rn + nwaves < rows ? rn + nwaves : rn);      // the flag of the row after next: used one iteration from now
    load_row(rn, nxt, lv_nxt);
    const bool dead = !lv_cur;
    lv_cur = lv_nxt; lv_nxt = lv_nn;
    if (dead) {     // a block of padded positions: dy = dres = 0 there, so dx = 0 and nothing is added to dgamma / dbeta
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const F8 z = {z4, z4};
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < d && !nofill) {      // (nofill: the caller has checked that every consumer of dx / dx_drop takes the hint too)
          st8(dx + r * (int64_t)d + c, z);
          if (dx_drop) st8(dx_drop + r * (int64_t)(d * RowMul<TY>::v) + c, z, d);
        }
      }
      cur = nxt;
      continue;
    }
    const float mu = cur.mu, rs = cur.rs;
    F8 xh[NC], g[NC], drv[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        const F8 dyv = cur.dy[i];
        const F8 xv = cur.x[i];
        drv[i] = cur.dr[i];
        xh[i] = {(xv.lo - mu) * rs, (xv.hi - mu) * rs};
        g[i] = {dyv.lo * gm[i].lo, dyv.hi * gm[i].hi};
        ag[i].lo += dyv.lo * xh[i].lo; ag[i].hi += dyv.hi * xh[i].hi;
        ab[i].lo += dyv.lo; ab[i].hi += dyv.hi;
        s1 += hsum8(g[i]);
        F8 gx = {g[i].lo * xh[i].lo, g[i].hi * xh[i].hi};
        s2 += hsum8(gx);
      }
    }
    cur = nxt;
    const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = (lane + 64 * i) * 8;
      if (c < d) {
        F8 o = {rs * (g[i].lo - m1 - xh[i].lo * m2), rs * (g[i].hi - m1 - xh[i].hi * m2)};
        if (dres) { o.lo += drv[i].lo; o.hi += drv[i].hi; }
        st8(dx + r * (int64_t)d + c, o);
        if (dx_drop) {
          const uint64_t base = (uint64_t)r * (uint64_t)d + (uint64_t)c;
          F8 od;
          float kpd[8];
          if (dd.thresh) afm_keep_scale<8>(dd, base, kpd);
          else {
#pragma unroll
            for (int k = 0; k < 8; ++k) kpd[k] = 1.0f;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) { od.lo[k] = o.lo[k] * kpd[k]; od.hi[k] = o.hi[k] * kpd[4 + k]; }
          st8(dx_drop + r * (int64_t)(d * RowMul<TY>::v) + c, od, d);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < d) { st8(sm + (w * 2 + 0) * d + c, ag[i]); st8(sm + (w * 2 + 1) * d + c, ab[i]); }
  }
  __syncthreads();
  // block sums straight into dgamma / dbeta: 2d fp32 atomics per block (4 MB over the launch, a few microseconds at the memory
  // side's rate) instead of a workspace pass and a reduce kernel behind every LayerNorm backward
  for (int j = threadIdx.x; j < 2 * d; j += blockDim.x) {
    const float t = sm[0 * 2 * d + j] + sm[1 * 2 * d + j] + sm[2 * 2 * d + j] + sm[3 * 2 * d + j];
    if (j < d) { if (dgamma) atomicAdd(dgamma + j, t); }
    else if (dbeta) atomicAdd(dbeta + j - d, t);
  }
}

extern "C" int afm_layernorm_fwd(const afm_ln_shape* s, const float* x, const float* gamma,
                                 const float* beta, const float* pos, void* y, float* mean,
                                 float* rstd, const void* add, float* x_sum, void* stream) {
  if (!s || !x || !gamma || !beta || !y || s->rows < 0 || s->d <= 0) return AFM_ERR_ARG;
  if (add && (!x_sum || s->add_dtype < AFM_F32 || s->add_dtype > AFM_F16 || s->seg_len != 0)) return AFM_ERR_ARG;
  const int add_dtype = s->add_dtype;
  const DropDev adrop = afm_make_drop(add ? &s->add_drop : nullptr);
  if (s->d > 64 * LN_MAXV) return AFM_ERR_UNSUPPORTED;
  if (s->rows == 0) return AFM_OK;
  int64_t g = (s->rows + 3) / 4;
  static const int fwd_cap = getenv("AFM_LN_FWD_BLOCKS") ? std::max(1, atoi(getenv("AFM_LN_FWD_BLOCKS"))) : 1280;   // five workgroups per CU: what the vectorised kernel's registers allow (two rows in flight per wave)
  if (g > fwd_cap) g = fwd_cap;
  hipStream_t st = (hipStream_t)stream;
  if (s->y_dtype < AFM_F32 || s->y_dtype > AFM_F16) return AFM_ERR_ARG;
  const int32_t* map = s->seg_len != 0 ? s->row_map : nullptr;                                         // (a row map belongs to the embedder's placement form)
  const uint8_t* live = (s->seg_len == 0 && (s->rows & 63) == 0) ? s->row_live : nullptr;            // (a hint: only the vectorised kernel takes it)
  if ((s->d % 8) == 0 && s->d <= 2048 && s->rows >= 64 && (!pos || ((uintptr_t)pos & 15) == 0)) {   // vectorised path
    // The grid is what is RESIDENT at once (every wave strides over the rows): five workgroups per CU at d <= 512 (92 registers), three
    // at d <= 1024 (148), one beyond (268).  Until round 5 the 1 280 blocks of the d <= 512 kernel were launched for every d: at d = 768
    // (c4) that is 1.67 rounds of equal-length blocks on 768 slots -- the kernels ran at 4.6 TB/s there against 5.3 at d = 512.
    static const bool fwd_forced = getenv("AFM_LN_FWD_BLOCKS") != nullptr;
    if (!fwd_forced) g = std::min<int64_t>((s->rows + 3) / 4, 256 * (s->d <= 512 ? 5 : s->d <= 1024 ? 3 : 1));
#define LN_FV(TY, TA, NC) AFM_LAUNCH((k_ln_fwd_vec<TY, TA, NC>), dim3((int)g), dim3(256), 0, st, x, gamma, beta, (TY*)y, mean, \
                                     rstd, s->rows, s->d, s->eps, (const TA*)add, x_sum, adrop, pos, s->seg_len,   \
                                     s->out_seg_stride, s->out_off, map, live, (s->flags & 1) != 0)
#define LN_FV2(TY, TA) do { if (s->d <= 512) LN_FV(TY, TA, 1); else if (s->d <= 1024) LN_FV(TY, TA, 2); else LN_FV(TY, TA, 4); } while (0)
    // the branch added in front of the norm has the dtype of the mode's activations or fp32
    if (s->y_dtype == AFM_BF16) { if (add_dtype == AFM_BF16) LN_FV2(bf16, bf16); else if (add_dtype == AFM_F32) LN_FV2(bf16, float); else return AFM_ERR_UNSUPPORTED; }
    else if (s->y_dtype == AFM_BF16X2) { if (add_dtype == AFM_BF16X2) LN_FV2(x2, x2); else if (add_dtype == AFM_F32) LN_FV2(x2, float); else return AFM_ERR_UNSUPPORTED; }
    else if (s->y_dtype == AFM_F16) { if (add_dtype == AFM_F16) LN_FV2(f16, f16); else if (add_dtype == AFM_F32) LN_FV2(f16, float); else return AFM_ERR_UNSUPPORTED; }
    else { if (add_dtype == AFM_BF16) LN_FV2(float, bf16); else if (add_dtype == AFM_F32) LN_FV2(float, float); else if (add_dtype == AFM_F16) LN_FV2(float, f16); else LN_FV2(float, x2); }
#undef LN_FV2
#undef LN_FV
    return AFM_OK;
  }
#define LN_FWD(NV)                                                                                  \
  do {                                                                                              \
    AFM_DT_SWITCH(s->y_dtype, TY, AFM_LAUNCH((k_ln_fwd<TY, NV>), dim3((int)g), dim3(256), 0, st, x, gamma, beta, pos, \
                         (TY*)y, mean, rstd, s->rows, s->d, s->seg_len, s->out_seg_stride,       \
                         s->out_off, s->eps, add, add_dtype, x_sum, adrop, map));                            \
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
  static const int bwd_cap = getenv("AFM_LN_BWD_BLOCKS") ? std::max(1, atoi(getenv("AFM_LN_BWD_BLOCKS"))) : 768;   // three workgroups per CU (the vectorised kernel: 138 registers, two rows in flight per wave)
  if (g > bwd_cap) g = bwd_cap;
  if (g < 1) g = 1;
  return (int)g;
}
// the vectorised backward adds its block sums to dgamma / dbeta with atomics and touches no workspace
static inline bool ln_bwd_vectorised(const afm_ln_shape* s) { return (s->d % 8) == 0 && s->d <= 2048 && s->rows >= 64; }
extern "C" int64_t afm_layernorm_bwd_ws_floats(const afm_ln_shape* s) {
  if (!s) return 0;
  if (ln_bwd_vectorised(s)) return 0;
  return (int64_t)ln_bwd_blocks(s->rows) * 2 * s->d;
}

template <typename TY, int NV>
__global__ void k_ln_bwd(const TY* __restrict__ dy, const float* __restrict__ x,
                         const float* __restrict__ gamma, const float* __restrict__ mean,
                         const float* __restrict__ rstd, const float* __restrict__ dres,
                         float* __restrict__ dx, float* __restrict__ partial, int64_t rows, int d,
                         int64_t seg_len, int64_t seg_stride, int64_t off, TY* __restrict__ dx_drop,
                         DropDev dd, const int32_t* __restrict__ map) {
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
    const int64_t dyrow = ln_out_row(r, seg_len, seg_stride, off, map);
    float xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      if (j < d) {
        const float dyv = ld_rc(dy, dyrow, j, d * RowMul<TY>::v);
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
        if (dx_drop) st_rc(dx_drop, r, j, d * RowMul<TY>::v, afm_drop(dd, (uint64_t)r * (uint64_t)d + (uint64_t)j, o));
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
  if (!s || !dy || !x || !gamma || !mean || !rstd || !dx || s->rows < 0 || s->d <= 0)
    return AFM_ERR_ARG;
  if (!partial && !ln_bwd_vectorised(s)) return AFM_ERR_ARG;      // (afm_layernorm_bwd_ws_floats says which shapes need it)
  if (s->d > 64 * LN_MAXV) return AFM_ERR_UNSUPPORTED;
  if (s->rows == 0) return AFM_OK;
  const int g = ln_bwd_blocks(s->rows);
  hipStream_t st = (hipStream_t)stream;
  const size_t shm = sizeof(float) * 8 * s->d;
  if (s->y_dtype < AFM_F32 || s->y_dtype > AFM_F16) return AFM_ERR_ARG;
  afm_note_hint(s->row_live ? ((ln_bwd_vectorised(s) && s->seg_len == 0 && (s->rows & 63) == 0) ? 1 : -1) : 0);
  if (ln_bwd_vectorised(s)) {   // vectorised path (dy rows follow the embedder's placement, if any)
    // resident workgroups only, as in the forward: three per CU at d <= 512 (136 registers), two at d <= 1024 (229), one beyond (438)
    static const bool bwd_forced = getenv("AFM_LN_BWD_BLOCKS") != nullptr;
    const int gv = bwd_forced ? g : (int)std::min<int64_t>((s->rows + 3) / 4, 256 * (s->d <= 512 ? 3 : s->d <= 1024 ? 2 : 1));
#define LN_BV(TY, NC) AFM_LAUNCH((k_ln_bwd_vec<TY, NC>), dim3(gv), dim3(256), shm, st, (const TY*)dy, x, gamma, mean, rstd, dres, dx, \
                                 dgamma, dbeta, s->rows, s->d, (TY*)dx_drop, dd, s->seg_len, s->out_seg_stride, s->out_off,   \
                                 (s->seg_len == 0 && (s->rows & 63) == 0) ? s->row_live : nullptr, s->seg_len != 0 ? s->row_map : nullptr, (s->flags & 1) != 0)
#define LN_BV2(TY) do { if (s->d <= 512) LN_BV(TY, 1); else if (s->d <= 1024) LN_BV(TY, 2); else LN_BV(TY, 4); } while (0)
    if (s->y_dtype == AFM_BF16) LN_BV2(bf16); else if (s->y_dtype == AFM_BF16X2) LN_BV2(x2); else if (s->y_dtype == AFM_F16) LN_BV2(f16); else LN_BV2(float);
#undef LN_BV2
#undef LN_BV
    return AFM_OK;
  }
#define LN_BWD(NV)                                                                                   \
  do {                                                                                               \
    AFM_DT_SWITCH(s->y_dtype, TY, AFM_LAUNCH((k_ln_bwd<TY, NV>), dim3(g), dim3(256), shm, st, (const TY*)dy, x,     \
                         gamma, mean, rstd, dres, dx, partial, s->rows, s->d, s->seg_len,            \
                         s->out_seg_stride, s->out_off, (TY*)dx_drop, dd, s->seg_len != 0 ? s->row_map : nullptr));                  \
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
