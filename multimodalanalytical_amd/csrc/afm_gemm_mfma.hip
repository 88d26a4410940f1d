// bf16 MFMA GEMMs for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate) with the afm_gemm epilogue.
//
//   NT  C[m][n] = sum_k A[m][k] * B[n][k]      forward (x W^T) and dgrad (dy (W^T)^T, W^T kept as a
//                                              second bf16 copy): both operands K-contiguous.
//   TN  C[m][n] += sum_k A[k][m] * B[k][n]     wgrad (dy^T x): the reduction index is the ROW of both
//                                              operands; tiles are staged row-major and read back
//                                              transposed with ds_read_b64_tr_b16.
//
// Block = 256 threads = 4 waves (2 x 2), block tile 128 x 128, wave tile 64 x 64 = 4 x 4 MFMA
// fragments, k-step 64 (two MFMA k-slices), LDS double-buffered (64 KiB) with an XOR swizzle so
// the fragment reads are bank-conflict free, next tile prefetched into registers while the
// current one is multiplied (one barrier per k-step).  The MFMA is issued with the WEIGHT tile as
// the first operand so a lane ends up with 4 consecutive output columns of one row: 8-/16-byte
// epilogue accesses.  Blocks are renumbered so the 8 XCDs each walk a contiguous range of tiles
// (all column tiles of a row panel share one L2).
#include "afm_common.h"

struct MfmaArgs {
  int M, N, K;
  int lda, ldb, ldc;
  const bf16* A;
  const bf16* B;
  void* C;
  const float* bias;
  const void* residual;
  void* pre_act;
  int act, accumulate;
  int tiles_m, tiles_n;
  int ksplit, kchunk;  // TN only
  DropDev dd;
};

#define BM 128
#define BN 128
#define BK 64

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective "each XCD gets a contiguous chunk" renumbering (8 XCDs, round-robin dispatch)
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------------------------------ epilogue
// Lane holds, for fragment (jn, im): C[m][n0 .. n0+3], m = m_base + im*16 + (lane&15),
// n0 = n_base + jn*16 + (lane>>4)*4.
template <bool C_BF16>
__device__ __forceinline__ void epilogue4(const MfmaArgs& g, int m, int n0, f32x4 v, bool vec_ok) {
  if (m >= g.M || n0 >= g.N) return;
  const int64_t ci = (int64_t)m * g.ldc + n0;
  const int nv = min(4, g.N - n0);
  if (vec_ok && nv == 4) {
    if (g.bias) v += *(const f32x4*)(g.bias + n0);
    if (g.pre_act) {
      if (C_BF16) { bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}; *(bf16x4*)((bf16*)g.pre_act + ci) = o; }
      else *(f32x4*)((float*)g.pre_act + ci) = v;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = v[r];
      if (g.act == AFM_ACT_RELU) x = fmaxf(x, 0.f);
      else if (g.act == AFM_ACT_GELU) x = afm_gelu(x);
      v[r] = afm_drop(g.dd, (uint64_t)m * (uint64_t)g.N + (uint64_t)(n0 + r), x);
    }
    if (g.residual) {
      if (C_BF16) { const bf16x4 rr = *(const bf16x4*)((const bf16*)g.residual + ci); v[0] += (float)rr[0]; v[1] += (float)rr[1]; v[2] += (float)rr[2]; v[3] += (float)rr[3]; }
      else v += *(const f32x4*)((const float*)g.residual + ci);
    }
    if (g.accumulate) {
      if (C_BF16) { const bf16x4 rr = *(const bf16x4*)((const bf16*)g.C + ci); v[0] += (float)rr[0]; v[1] += (float)rr[1]; v[2] += (float)rr[2]; v[3] += (float)rr[3]; }
      else v += *(const f32x4*)((const float*)g.C + ci);
    }
    if (C_BF16) { bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}; *(bf16x4*)((bf16*)g.C + ci) = o; }
    else *(f32x4*)((float*)g.C + ci) = v;
    return;
  }
  for (int r = 0; r < nv; ++r) {
    float x = v[r];
    const int n = n0 + r;
    if (g.bias) x += g.bias[n];
    if (g.pre_act) { if (C_BF16) ((bf16*)g.pre_act)[ci + r] = (bf16)x; else ((float*)g.pre_act)[ci + r] = x; }
    if (g.act == AFM_ACT_RELU) x = fmaxf(x, 0.f);
    else if (g.act == AFM_ACT_GELU) x = afm_gelu(x);
    x = afm_drop(g.dd, (uint64_t)m * (uint64_t)g.N + (uint64_t)n, x);
    if (g.residual) x += C_BF16 ? (float)((const bf16*)g.residual)[ci + r] : ((const float*)g.residual)[ci + r];
    if (g.accumulate) x += C_BF16 ? (float)((const bf16*)g.C)[ci + r] : ((const float*)g.C)[ci + r];
    if (C_BF16) ((bf16*)g.C)[ci + r] = (bf16)x; else ((float*)g.C)[ci + r] = x;
  }
}

// ------------------------------------------------------------------------------------------ NT
// LDS tile [128 rows][64 k] bf16 = 128-byte rows of 8 16-byte chunks; chunk c of row r is stored at
// chunk (c ^ (r & 7)): the 16 lanes of a ds_read_b128 group then hit 16 distinct 16-byte slots.
__device__ __forceinline__ int nt_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <bool C_BF16>
__global__ __launch_bounds__(256) void k_gemm_nt(MfmaArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * BM * BK * 2];  // [buf][A|B]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  const int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;

  // staging: thread -> 4 rows x one 16-byte chunk per operand
  const int srow = t >> 3, sch = t & 7;
  const bf16* ap[4];
  const bf16* bp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = min(m0 + srow + 32 * i, g.M - 1);  // clamp: rows past M are computed, never stored
    const int rb = min(n0 + srow + 32 * i, g.N - 1);
    ap[i] = g.A + (int64_t)ra * g.lda + sch * 8;
    bp[i] = g.B + (int64_t)rb * g.ldb + sch * 8;
  }
  uint4 ra_[4], rb_[4];
  auto gload = [&](int k0) {
    const bool in = k0 + sch * 8 < g.K;  // K % 8 == 0: a chunk is entirely inside or outside
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra_[i] = in ? *(const uint4*)(ap[i] + k0) : make_uint4(0, 0, 0, 0);
      rb_[i] = in ? *(const uint4*)(bp[i] + k0) : make_uint4(0, 0, 0, 0);
    }
  };
  auto sstore = [&](int buf) {
    unsigned char* a = lds + buf * (2 * BM * BK * 2);
    unsigned char* b = a + BM * BK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = srow + 32 * i;
      *(uint4*)(a + nt_off(r, sch)) = ra_[i];
      *(uint4*)(b + nt_off(r, sch)) = rb_[i];
    }
  };

  f32x4 acc[4][4];  // [jn][im]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (g.K + BK - 1) / BK;
  gload(0);
  sstore(0);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const unsigned char* a = lds + buf * (2 * BM * BK * 2);
    const unsigned char* b = a + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *(const bf16x8*)(a + nt_off(wm * 64 + i * 16 + fr, ks * 4 + fq));
        bfr[i] = *(const bf16x8*)(b + nt_off(wn * 64 + i * 16 + fr, ks * 4 + fq));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[j][i], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  const bool vec_ok = (g.ldc & 3) == 0 && (g.N & 3) == 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      epilogue4<C_BF16>(g, m0 + wm * 64 + i * 16 + fr, n0 + wn * 64 + j * 16 + fq * 4, acc[j][i], vec_ok);
}

// ------------------------------------------------------------------------------------------ TN (wgrad)
// C[m][n] += sum_k A[k][m] B[k][n]: A is dy (rows = tokens, cols = output features m), B is x
// (rows = tokens, cols = input features n).  LDS tile [64 k-rows][128 cols] bf16 = 256-byte rows of
// 16 chunks; chunk c of row r is stored at chunk c ^ s(r), s(r) = 2*(r&3) + 8*((r>>3)&1): a
// ds_read_b64_tr_b16 half-wave (2 groups x 4 rows x 4 column quads) then covers all 64 banks once.
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }
__device__ __forceinline__ int tn_off(int row, int chunk) { return row * 256 + ((chunk ^ tn_swz(row)) << 4); }

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x4 ds_read_tr(const unsigned char* p) {
  // ds_read_b64_tr_b16 through the compiler builtin, so hipcc schedules and counts it (lgkmcnt)
  const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(bf16x4, r);
}

__global__ __launch_bounds__(256) void k_gemm_tn(MfmaArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * BK * BM * 2];  // [buf][A|B][64][128]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int ntile = g.tiles_m * g.tiles_n;
  const int bid = xcd_remap(blockIdx.x, ntile * g.ksplit);
  const int tile = bid / g.ksplit, ks_id = bid % g.ksplit;
  const int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;
  const int kbeg = ks_id * g.kchunk, kend = min(g.K, kbeg + g.kchunk);

  // staging: thread -> 4 k-rows x one 16-byte chunk (8 columns) per operand
  const int srow = t >> 4, sch = t & 15;
  const bool a_in = m0 + sch * 8 < g.M, b_in = n0 + sch * 8 < g.N;  // M, N % 8 == 0
  uint4 ra_[4], rb_[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + srow + 16 * i;
      const bool kin = k < kend;
      ra_[i] = (kin && a_in) ? *(const uint4*)(g.A + (int64_t)k * g.lda + m0 + sch * 8) : make_uint4(0, 0, 0, 0);
      rb_[i] = (kin && b_in) ? *(const uint4*)(g.B + (int64_t)k * g.ldb + n0 + sch * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto sstore = [&](int buf) {
    unsigned char* a = lds + buf * (2 * BK * BM * 2);
    unsigned char* b = a + BK * BM * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = srow + 16 * i;
      *(uint4*)(a + tn_off(r, sch)) = ra_[i];
      *(uint4*)(b + tn_off(r, sch)) = rb_[i];
    }
  };

  f32x4 acc[4][4];  // [im][jn]: D rows = m (A operand), cols = n (B operand)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    gload(kbeg);
    sstore(0);
  }
  __syncthreads();
  // transposed fragment read: lane = 16*grp + 4*q + p supplies row (kb + 8*grp + 4*half + q), the 4
  // columns (cbase + 4*p ..); it receives, for column cbase + (lane&15), the 4 rows of the block.
  const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);
    const unsigned char* a = lds + buf * (2 * BK * BM * 2);
    const unsigned char* b = a + BK * BM * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // two 32-deep k-slices
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ca = wm * 64 + i * 16, cb = wn * 64 + i * 16;  // first column of the 16-wide fragment
        const int r0 = ks * 32 + grp * 8 + q, r1 = r0 + 4;
        const int cha = (ca >> 3) + (p >> 1), chb = (cb >> 3) + (p >> 1);
        const bf16x4 a0 = ds_read_tr(a + tn_off(r0, cha) + ((p & 1) << 3));
        const bf16x4 a1 = ds_read_tr(a + tn_off(r1, cha) + ((p & 1) << 3));
        const bf16x4 b0 = ds_read_tr(b + tn_off(r0, chb) + ((p & 1) << 3));
        const bf16x4 b1 = ds_read_tr(b + tn_off(r1, chb) + ((p & 1) << 3));
        af[i] = (bf16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        bfr[i] = (bf16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // D[row = fq*4 + r][col = fr] -> C[m = .. + fq*4 + r][n = .. + fr]; fp32 atomics when the
  // reduction is split over blocks (the gradient buffer accumulates anyway), plain += otherwise.
  const int fr = lane & 15, fq = lane >> 4;
  float* C = (float*)g.C;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + i * 16 + fq * 4 + r;
        if (m < g.M && n < g.N) {
          float* c = C + (int64_t)m * g.ldc + n;
          if (g.ksplit > 1) atomicAdd(c, acc[i][j][r]);
          else *c = acc[i][j][r] + (g.accumulate ? *c : 0.f);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------ dispatch
static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int afm_gemm_mfma_try(const afm_gemm_desc* d, hipStream_t st) {
  if (d->a_dtype != AFM_BF16 || d->b_dtype != AFM_BF16) return AFM_ERR_UNSUPPORTED;
  MfmaArgs g;
  g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
  g.A = (const bf16*)d->A; g.B = (const bf16*)d->B; g.C = d->C;
  g.bias = d->bias; g.residual = d->residual; g.pre_act = d->pre_act;
  g.act = d->act; g.accumulate = d->accumulate;
  g.dd = afm_make_drop(&d->drop);
  g.tiles_m = (d->M + BM - 1) / BM; g.tiles_n = (d->N + BN - 1) / BN;
  g.ksplit = 1; g.kchunk = d->K;
  if (!aligned16(d->A) || !aligned16(d->B) || (d->lda & 7) || (d->ldb & 7)) return AFM_ERR_UNSUPPORTED;
  if (!d->transA && d->transB) {  // NT
    if ((d->K & 7) || d->K < 32 || d->N < 16) return AFM_ERR_UNSUPPORTED;
    if (d->bias && !aligned16(d->bias)) return AFM_ERR_UNSUPPORTED;
    if (!aligned16(d->C) || (d->residual && !aligned16(d->residual)) || (d->pre_act && !aligned16(d->pre_act)))
      return AFM_ERR_UNSUPPORTED;
    const dim3 grid(g.tiles_m * g.tiles_n);
    if (d->c_dtype == AFM_BF16) AFM_LAUNCH(k_gemm_nt<true>, grid, dim3(256), 0, st, g);
    else AFM_LAUNCH(k_gemm_nt<false>, grid, dim3(256), 0, st, g);
    afm_set_last_algo("mfma_nt");
    return AFM_OK;
  }
  if (d->transA && !d->transB) {  // TN: the wgrad form only
    if (d->c_dtype != AFM_F32 || d->bias || d->residual || d->pre_act || d->act != AFM_ACT_NONE || d->drop.p > 0.f)
      return AFM_ERR_UNSUPPORTED;
    if ((d->M & 7) || (d->N & 7) || d->K < 64 || d->M < 16 || d->N < 16) return AFM_ERR_UNSUPPORTED;
    const int tiles = g.tiles_m * g.tiles_n;
    int ksplit = 1;
    if (tiles < 512) {
      ksplit = (768 + tiles - 1) / tiles;
      const int maxs = (d->K + 511) / 512;  // at least 8 k-steps per block
      if (ksplit > maxs) ksplit = maxs;
      if (ksplit < 1) ksplit = 1;
    }
    int kchunk = (d->K + ksplit - 1) / ksplit;
    kchunk = (kchunk + BK - 1) / BK * BK;
    ksplit = (d->K + kchunk - 1) / kchunk;
    g.ksplit = ksplit; g.kchunk = kchunk;
    if (ksplit > 1 && !d->accumulate) {
      if (hipMemset2DAsync(d->C, sizeof(float) * d->ldc, 0, sizeof(float) * d->N, d->M, st) != hipSuccess)
        return AFM_ERR_LAUNCH;
    }
    AFM_LAUNCH(k_gemm_tn, dim3(tiles * ksplit), dim3(256), 0, st, g);
    afm_set_last_algo(ksplit > 1 ? "mfma_tn_splitk" : "mfma_tn");
    return AFM_OK;
  }
  return AFM_ERR_UNSUPPORTED;
}
