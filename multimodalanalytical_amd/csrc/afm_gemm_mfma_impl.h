// Single-pass 16-bit MFMA GEMMs for gfx950 (v_mfma_f32_16x16x32_bf16 / _f16, fp32 accumulate) with the afm_gemm epilogue.
// Written against `e16` (afm_common.h) and compiled twice: afm_gemm_mfma.hip (bf16) and afm_gemm_mfma_f16.hip (fp16, the
// reference's own GPU precision: trainer/trainer.py:69 "16-mixed").
//
//   NT  C[m][n] = sum_k A[m][k] * B[n][k]      forward (x W^T) and dgrad (dy (W^T)^T, W^T kept as a
//                                              second e16 copy): both operands K-contiguous.
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
#include <algorithm>
#include <functional>
#include <vector>
#include "afm_common.h"

namespace AFM_E16_NS {

struct MfmaArgs {
  int M, N, K;
  int lda, ldb, ldc;
  const e16* A;
  const e16* B;
  void* C;
  const float* bias;
  const void* residual;
  void* pre_act;
  float* a_colsum;
  const uint8_t* k_live;   // afm_gemm_desc.k_live: 64-row blocks of A's stored rows (TN: k-steps; NT: row blocks of A and C)
  int live_off;            // NT, persistent kernels: LDS byte offset of the live / dead tile lists (0: no hint in use)
  int mperm;               // NT with tile lists: row panels dealt round-robin to the XCDs (tile_mn), = tiles_m / 8; 0 = contiguous ranges
  int deal;                // afm_gemm_desc.reserved2 bit 2: the hint's live rows are packed to the front of the matrix (deal panels / k-steps)
  int dead_pre;            // NT with k_live in the FORWARD sense (afm_gemm_desc.reserved2 bit 1): pre_act is an output, zero-filled with C in dead tiles
  int dead_nofill;         // ... reserved2 bit 3: dead tiles are not written at all (the caller's buffers hold finite values there already)
  int act, accumulate;
  int tiles_m, tiles_n;
  int xgc;             // persistent NT kernels: column groups of the XCD-aware tile walk (tile_mn below); 0 / 1 = row-major
  int ksplit, kchunk;  // TN only
  int bias_in_lds;     // persistent NT: staged epilogue enabled (bias vector cached in LDS)
  int glu_f;           // gated-FFN interleave (include/afm_hip.h): bias / wgrad rows are translated to the [W1 ; Wg] order
  unsigned long long* stamps;   // ablation builds: per-workgroup phase time stamps (wall_clock64), else null
  DropDev dd;
};

// Tile index -> (row tile, column tile) of the persistent NT kernels.  XCD x owns the indices [x * tpx, (x + 1) * tpx).  Row-major
// (xgc <= 1) that is a band of row panels with ALL column tiles: every A panel is fetched once, but the whole weight matrix streams
// through the XCD's 4-MB L2 once per row panel -- fine while it fits (c2: <= 2 MB), ruinous when it does not: c4's gated FFN
// up-projection (N 6144, K 768: 9.4 MB) read 3.9 GB per launch for 0.21 GB of operands (profiles/r05_c4_gemm_fp16_pmc.json, L2 hit 0.56).
// With xgc column groups an XCD owns a RECTANGLE: (tiles_m / (8 / xgc)) row panels x (tiles_n / xgc) column tiles, walked row-major
// inside it, so its slice of the weights stays in L2 and an A panel is fetched by xgc XCDs instead of one.  The launchers pick xgc
// (nt_pick_xgc) where the weights exceed what an L2 keeps and the tile grid divides evenly.
// With a padded-row hint in use (mperm = tiles_m / 8 > 0) the row panels are DEALT to the XCDs instead of cut into eight contiguous bands:
// panel p of the banded order is row tile (p % mperm) * 8 + p / mperm, so XCD x works on the row tiles 8 j + x.  Round 6: with the batch's
// live rows packed to the front of the matrix (afm_compact_plan mode 2) the bands of the last XCDs held nothing but dead tiles -- every
// hinted GEMM ran on half the chip (c3 4 400 -> 3 890 samples/s) -- and the tile lists only balance the workgroups INSIDE an XCD.
__device__ __forceinline__ void tile_mn(const MfmaArgs& g, int tile, int& mt, int& nt) {
  if (g.xgc <= 1) { mt = tile / g.tiles_n; nt = tile % g.tiles_n; }
  else {
    const int tpx = (g.tiles_m * g.tiles_n) >> 3;
    const int x = tile / tpx, i = tile - x * tpx;
    const int cw = g.tiles_n / g.xgc, rh = g.tiles_m / (8 / g.xgc);
    mt = (x / g.xgc) * rh + i / cw;
    nt = (x % g.xgc) * cw + i % cw;
  }
  if (g.mperm > 0) mt = (mt % g.mperm) * 8 + mt / g.mperm;
}
static inline int nt_pick_xgc(int tiles_m, int tiles_n, int64_t weight_bytes) {
  static const int force = getenv("AFM_NT_XGC") ? atoi(getenv("AFM_NT_XGC")) : -1;     // (A / B runs: 1 = row-major everywhere)
  const int64_t keep = 5 * 512 * 1024;                           // 2.5 MB of a 4-MB L2 for the weights
  if (((int64_t)tiles_m * tiles_n) & 7) return 1;
  int pick = 1;
  if (force < 0 && weight_bytes <= keep) return 1;
  for (int gc = 2; gc <= 8; gc *= 2) {
    if (tiles_n % gc || tiles_m % (8 / gc)) continue;
    pick = gc;
    if (force < 0 && weight_bytes / gc <= keep) break;
    if (force == gc) break;
  }
  return force == 1 ? 1 : pick;
}

#define BM 128
#define BN 128
#define BK 64

// interleaved row / column n of the gated up-projection -> its index in the reference's [W1 ; Wg] order
__device__ __forceinline__ int glu_deint(int n, int f) { return ((n >> 3) << 2) + (n & 3) + ((n >> 2) & 1) * f; }

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
    if (g.act == AFM_ACT_GELU_BWD) {
      f32x4 u;
      if (C_BF16) { const e16x4 uu = *(const e16x4*)((const e16*)g.pre_act + ci); u = (f32x4){(float)uu[0], (float)uu[1], (float)uu[2], (float)uu[3]}; }
      else u = *(const f32x4*)((const float*)g.pre_act + ci);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        v[r] = afm_drop(g.dd, (uint64_t)m * (uint64_t)g.N + (uint64_t)(n0 + r), v[r]) * afm_gelu_grad_o<C_BF16>(u[r]);
    } else {
      if (g.pre_act) {
        if (C_BF16) { e16x4 o = {(e16)v[0], (e16)v[1], (e16)v[2], (e16)v[3]}; *(e16x4*)((e16*)g.pre_act + ci) = o; }
        else *(f32x4*)((float*)g.pre_act + ci) = v;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = v[r];
        if (g.act == AFM_ACT_RELU) x = fmaxf(x, 0.f);
        else if (g.act == AFM_ACT_GELU) x = afm_gelu_o<C_BF16>(x);
        v[r] = afm_drop(g.dd, (uint64_t)m * (uint64_t)g.N + (uint64_t)(n0 + r), x);
      }
    }
    if (g.residual) {
      if (C_BF16) { const e16x4 rr = *(const e16x4*)((const e16*)g.residual + ci); v[0] += (float)rr[0]; v[1] += (float)rr[1]; v[2] += (float)rr[2]; v[3] += (float)rr[3]; }
      else v += *(const f32x4*)((const float*)g.residual + ci);
    }
    if (g.accumulate) {
      if (C_BF16) { const e16x4 rr = *(const e16x4*)((const e16*)g.C + ci); v[0] += (float)rr[0]; v[1] += (float)rr[1]; v[2] += (float)rr[2]; v[3] += (float)rr[3]; }
      else v += *(const f32x4*)((const float*)g.C + ci);
    }
    if (C_BF16) { e16x4 o = {(e16)v[0], (e16)v[1], (e16)v[2], (e16)v[3]}; *(e16x4*)((e16*)g.C + ci) = o; }
    else *(f32x4*)((float*)g.C + ci) = v;
    return;
  }
  for (int r = 0; r < nv; ++r) {
    float x = v[r];
    const int n = n0 + r;
    if (g.bias) x += g.bias[n];
    if (g.act == AFM_ACT_GELU_BWD) {
      const float u = C_BF16 ? (float)((const e16*)g.pre_act)[ci + r] : ((const float*)g.pre_act)[ci + r];
      x = afm_drop(g.dd, (uint64_t)m * (uint64_t)g.N + (uint64_t)n, x) * afm_gelu_grad_o<C_BF16>(u);
    } else {
      if (g.pre_act) { if (C_BF16) ((e16*)g.pre_act)[ci + r] = (e16)x; else ((float*)g.pre_act)[ci + r] = x; }
      if (g.act == AFM_ACT_RELU) x = fmaxf(x, 0.f);
      else if (g.act == AFM_ACT_GELU) x = afm_gelu_o<C_BF16>(x);
      x = afm_drop(g.dd, (uint64_t)m * (uint64_t)g.N + (uint64_t)n, x);
    }
    if (g.residual) x += C_BF16 ? (float)((const e16*)g.residual)[ci + r] : ((const float*)g.residual)[ci + r];
    if (g.accumulate) x += C_BF16 ? (float)((const e16*)g.C)[ci + r] : ((const float*)g.C)[ci + r];
    if (C_BF16) ((e16*)g.C)[ci + r] = (e16)x; else ((float*)g.C)[ci + r] = x;
  }
}

// Same, for a tile known to be entirely inside C with 16-byte aligned rows: no per-lane exits, so
// every wave issues exactly one store instruction per fragment (the persistent kernel counts on it).
template <bool C_BF16>
__device__ __forceinline__ void epilogue4_full(const MfmaArgs& g, int m, int n0, f32x4 v) {
  const int64_t ci = (int64_t)m * g.ldc + n0;
  if (g.bias) v += *(const f32x4*)(g.bias + n0);
  if (g.act == AFM_ACT_GELU_BWD) {
    f32x4 u;
    if (C_BF16) { const e16x4 uu = *(const e16x4*)((const e16*)g.pre_act + ci); u = (f32x4){(float)uu[0], (float)uu[1], (float)uu[2], (float)uu[3]}; }
    else u = *(const f32x4*)((const float*)g.pre_act + ci);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      v[r] = afm_drop(g.dd, (uint64_t)m * (uint64_t)g.N + (uint64_t)(n0 + r), v[r]) * afm_gelu_grad_o<C_BF16>(u[r]);
  } else if (g.pre_act) {
    if (C_BF16) { e16x4 o = {(e16)v[0], (e16)v[1], (e16)v[2], (e16)v[3]}; *(e16x4*)((e16*)g.pre_act + ci) = o; }
    else *(f32x4*)((float*)g.pre_act + ci) = v;
  }
  if (g.act != AFM_ACT_GELU_BWD && (g.act != AFM_ACT_NONE || g.dd.thresh)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = v[r];
      if (g.act == AFM_ACT_RELU) x = fmaxf(x, 0.f);
      else if (g.act == AFM_ACT_GELU) x = afm_gelu_o<C_BF16>(x);
      v[r] = afm_drop(g.dd, (uint64_t)m * (uint64_t)g.N + (uint64_t)(n0 + r), x);
    }
  }
  if (g.residual) {
    if (C_BF16) { const e16x4 rr = *(const e16x4*)((const e16*)g.residual + ci); v[0] += (float)rr[0]; v[1] += (float)rr[1]; v[2] += (float)rr[2]; v[3] += (float)rr[3]; }
    else v += *(const f32x4*)((const float*)g.residual + ci);
  }
  if (g.accumulate) {
    if (C_BF16) { const e16x4 rr = *(const e16x4*)((const e16*)g.C + ci); v[0] += (float)rr[0]; v[1] += (float)rr[1]; v[2] += (float)rr[2]; v[3] += (float)rr[3]; }
    else v += *(const f32x4*)((const float*)g.C + ci);
  }
  if (C_BF16) { e16x4 o = {(e16)v[0], (e16)v[1], (e16)v[2], (e16)v[3]}; *(e16x4*)((e16*)g.C + ci) = o; }
  else *(f32x4*)((float*)g.C + ci) = v;
}

// ------------------------------------------------------------------------------------------ NT
// LDS tile [128 rows][64 k] e16 = 128-byte rows of 8 16-byte chunks; chunk c of row r is stored at
// chunk (c ^ (r & 7)): the 16 lanes of a ds_read_b128 group then hit 16 distinct 16-byte slots.
__device__ __forceinline__ int nt_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// WM x WN MFMA fragments (16x16) per wave, NWM x NWN waves per block, BKT-deep k-steps.
template <bool C_BF16, int WM, int WN, int NWM, int NWN, int BKT>
__global__ __launch_bounds__(64 * NWM * NWN) void k_gemm_nt(MfmaArgs g) {
  constexpr int NT = 64 * NWM * NWN;             // threads
  constexpr int TBM = 16 * WM * NWM, TBN = 16 * WN * NWN;
  constexpr int CH = BKT / 8;                    // 16-byte chunks per tile row
  constexpr int ROWB = BKT * 2;                  // bytes per tile row
  constexpr int RPP = NT / CH;                   // rows staged per pass
  constexpr int PA = TBM / RPP, PB = TBN / RPP;  // passes
  static_assert(TBM % RPP == 0 && TBN % RPP == 0, "tile/thread mismatch");
  constexpr int ABYTES = TBM * ROWB, BBYTES = TBN * ROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [2][A | B]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w / NWN, wn = w % NWN;
  const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  const int m0 = (tile / g.tiles_n) * TBM, n0 = (tile % g.tiles_n) * TBN;
  auto off = [](int row, int chunk) { return row * ROWB + ((chunk ^ (row & (CH - 1))) << 4); };

  const int srow = t / CH, sch = t % CH;
  const e16* ap[PA];
  const e16* bp[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) ap[i] = g.A + (int64_t)min(m0 + srow + RPP * i, g.M - 1) * g.lda + sch * 8;
#pragma unroll
  for (int i = 0; i < PB; ++i) bp[i] = g.B + (int64_t)min(n0 + srow + RPP * i, g.N - 1) * g.ldb + sch * 8;
  uint4 ra_[PA], rb_[PB];
  auto gload = [&](int k0) {
    const bool in = k0 + sch * 8 < g.K;  // K % 8 == 0: a chunk is entirely inside or outside
#pragma unroll
    for (int i = 0; i < PA; ++i) ra_[i] = in ? *(const uint4*)(ap[i] + k0) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i) rb_[i] = in ? *(const uint4*)(bp[i] + k0) : make_uint4(0, 0, 0, 0);
  };
  auto sstore = [&](int buf) {
    unsigned char* a = lds + buf * (ABYTES + BBYTES);
    unsigned char* b = a + ABYTES;
#pragma unroll
    for (int i = 0; i < PA; ++i) *(uint4*)(a + off(srow + RPP * i, sch)) = ra_[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *(uint4*)(b + off(srow + RPP * i, sch)) = rb_[i];
  };

  f32x4 acc[WN][WM];
#pragma unroll
  for (int j = 0; j < WN; ++j)
#pragma unroll
    for (int i = 0; i < WM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (g.K + BKT - 1) / BKT;
  gload(0);
  sstore(0);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BKT);
    const unsigned char* a = lds + buf * (ABYTES + BBYTES);
    const unsigned char* b = a + ABYTES;
#pragma unroll
    for (int ks = 0; ks < BKT / 32; ++ks) {
      e16x8 af[WM], bfr[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) af[i] = *(const e16x8*)(a + off(wm * 16 * WM + i * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < WN; ++j) bfr[j] = *(const e16x8*)(b + off(wn * 16 * WN + j * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i)
          acc[j][i] = mfma16(bfr[j], af[i], acc[j][i]);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  const bool vec_ok = (g.ldc & 3) == 0 && (g.N & 3) == 0;
#pragma unroll
  for (int j = 0; j < WN; ++j)
#pragma unroll
    for (int i = 0; i < WM; ++i)
      epilogue4<C_BF16>(g, m0 + wm * 16 * WM + i * 16 + fr, n0 + wn * 16 * WN + j * 16 + fq * 4, acc[j][i], vec_ok);
}

template <bool C_BF16, int WM, int WN, int NWM, int NWN, int BKT>
static int launch_nt(MfmaArgs& g, hipStream_t st) {
  constexpr int TBM = 16 * WM * NWM, TBN = 16 * WN * NWN;
  constexpr int shm = 2 * (TBM + TBN) * BKT * 2;
  g.tiles_m = (g.M + TBM - 1) / TBM; g.tiles_n = (g.N + TBN - 1) / TBN;
  auto kern = k_gemm_nt<C_BF16, WM, WN, NWM, NWN, BKT>;
  if (shm > 64 * 1024) {
    static AfmOncePerDevice done;  // per instantiation
    if (done.need()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
  }
  AFM_LAUNCH(kern, dim3(g.tiles_m * g.tiles_n), dim3(64 * NWM * NWN), shm, st, g);
  return AFM_OK;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ------------------------------------------------------------------------------------------ staged epilogue
// Full-tile epilogue of the persistent kernel.  A wave's 64 x 64 accumulator block is transposed
// through a wave-private LDS patch (16 rows x 64 fp32 at a time, 272-byte rows) so that every global
// access of the epilogue is 16 bytes per lane on whole 128-byte lines: a store instruction covers
// 8 complete rows of the tile instead of 16 scattered 32-byte segments (the fragment layout gives
// a lane 4 columns of one row), which is what bounded the K = 512 GEMMs of this model.
#define STG_LD 68   // floats per staged row (64 + 4: conflict-free 16-byte writes, 16-byte aligned reads)
template <bool C_BF16, int WM>
__device__ __forceinline__ void epilogue_staged(const MfmaArgs& g, float* stg, const float* bias_lds,
                                                f32x4 (&acc)[4][WM], int mw, int nw, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  if (C_BF16) {
    const int c8 = (lane & 7) * 8, r8 = lane >> 3;
    const int n = nw + c8;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) { b0 = *(const f32x4*)(bias_lds + n); b1 = *(const f32x4*)(bias_lds + n + 4); }
    e16x8 uu[2 * WM];
    if (g.act == AFM_ACT_GELU_BWD) {   // all pre-activation loads first: one wait, before any store
#pragma unroll
      for (int q = 0; q < 2 * WM; ++q)
        uu[q] = *(const e16x8*)((const e16*)g.pre_act + (int64_t)(mw + (q >> 1) * 16 + (q & 1) * 8 + r8) * g.ldc + n);
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)(stg + fr * STG_LD + j * 16 + fq * 4) = acc[j][i];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int row = hf * 8 + r8;
        const int mrow = mw + i * 16 + row;
        const f32x4 v0 = *(const f32x4*)(stg + row * STG_LD + c8) + b0;
        const f32x4 v1 = *(const f32x4*)(stg + row * STG_LD + c8 + 4) + b1;
        float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const int64_t ci = (int64_t)mrow * g.ldc + n;
        const uint64_t di = (uint64_t)mrow * (uint64_t)g.N + (uint64_t)n;
        float kp8[8];      // keep * scale of the lane's eight elements (one block hash, four pair mixes), ones without dropout
        if (g.dd.thresh) afm_keep_scale<8>(g.dd, di, kp8);
        else {
#pragma unroll
          for (int k = 0; k < 8; ++k) kp8[k] = 1.0f;
        }
        if (g.act == AFM_ACT_GELU_BWD) {
          const e16x8 u = uu[i * 2 + hf];
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = x[k] * kp8[k] * afm_gelu_grad16((float)u[k]);
        } else {
          if (g.pre_act) {
            e16x8 o = {(e16)x[0], (e16)x[1], (e16)x[2], (e16)x[3], (e16)x[4], (e16)x[5], (e16)x[6], (e16)x[7]};
            *(e16x8*)((e16*)g.pre_act + ci) = o;
          }
          if (g.act != AFM_ACT_NONE || g.dd.thresh) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              float y = x[k];
              if (g.act == AFM_ACT_RELU) y = fmaxf(y, 0.f);
              else if (g.act == AFM_ACT_GELU) y = afm_gelu16(y);
              x[k] = y * kp8[k];
            }
          }
        }
        e16x8 o = {(e16)x[0], (e16)x[1], (e16)x[2], (e16)x[3], (e16)x[4], (e16)x[5], (e16)x[6], (e16)x[7]};
        *(e16x8*)((e16*)g.C + ci) = o;
      }
    }
  } else {
    const int c4 = (lane & 15) * 4, r4 = lane >> 4;
    const int n = nw + c4;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) b0 = *(const f32x4*)(bias_lds + n);
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)(stg + fr * STG_LD + j * 16 + fq * 4) = acc[j][i];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int row = qq * 4 + r4;
        const int mrow = mw + i * 16 + row;
        f32x4 v = *(const f32x4*)(stg + row * STG_LD + c4) + b0;
        const int64_t ci = (int64_t)mrow * g.ldc + n;
        const uint64_t di = (uint64_t)mrow * (uint64_t)g.N + (uint64_t)n;
        float kp4[4];
        if (g.dd.thresh) afm_keep_scale<4>(g.dd, di, kp4);
        else {
#pragma unroll
          for (int k = 0; k < 4; ++k) kp4[k] = 1.0f;
        }
        if (g.act == AFM_ACT_GELU_BWD) {
          const f32x4 u = *(const f32x4*)((const float*)g.pre_act + ci);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v[k] * kp4[k] * afm_gelu_grad(u[k]);
        } else {
          if (g.pre_act) *(f32x4*)((float*)g.pre_act + ci) = v;
          if (g.act != AFM_ACT_NONE || g.dd.thresh) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float y = v[k];
              if (g.act == AFM_ACT_RELU) y = fmaxf(y, 0.f);
              else if (g.act == AFM_ACT_GELU) y = afm_gelu(y);
              v[k] = y * kp4[k];
            }
          }
        }
        if (g.residual) v += *(const f32x4*)((const float*)g.residual + ci);
        if (g.accumulate) v += *(const f32x4*)((const float*)g.C + ci);
        *(f32x4*)((float*)g.C + ci) = v;
      }
    }
  }
}

// Compile-time epilogue kinds of the e16 staged epilogue (the training step's four fused forms): the
// generic one above tests act / dropout / pre_act per element group at run time, which costs scalar
// branches between every eight elements and keeps the compiler from scheduling across them.
//   EPI_PLAIN     C = acc + bias
//   EPI_DROP      C = dropout(acc + bias)
//   EPI_GELU      pre_act = acc + bias (if kept);  C = dropout(gelu(acc + bias))
//   EPI_GELU_BWD  C = dropout(acc) * gelu'(pre_act)
//   EPI_GELU_SG   C = dropout(gelu(acc + bias));  pre_act = keep * scale * gelu'(acc + bias)
//   EPI_MUL       C = acc * pre_act                (the dgrad partner of EPI_GELU_SG: no erf, no hash)
// Dropout indices are 32-bit here (the dispatcher requires M*N <= 2^32, where the stream's high-word
// term is zero), which also removes a 64-bit multiply-add chain per element group.
//   EPI_GLU / EPI_GLU_SG  gated FFN forward on the interleaved (2f-wide) accumulators: C (f wide) = dropout(gelu(u) * v)
//                         [SG: pre_act (2f wide) = keep*scale*[gelu'(u) v | gelu(u)]]
//   EPI_GLU_BWD           accumulators = dg (f wide): C (2f wide, interleaved) = [dg * saved_a | dg * saved_b]
enum { EPI_GENERIC = 0, EPI_PLAIN = 1, EPI_DROP = 2, EPI_GELU = 3, EPI_GELU_BWD = 4, EPI_GELU_SG = 5, EPI_MUL = 6,
       EPI_GLU = 7, EPI_GLU_SG = 8, EPI_GLU_BWD = 9 };
// keep * scale of N consecutive elements from idx0 (a multiple of N; index below 2^32), or all ones where the site does not drop
template <int N, bool DROP_ON>
__device__ __forceinline__ void afm_keep_scale32(const DropDev& d, uint32_t idx0, float (&out)[N]) {
  if constexpr (DROP_ON) afm_keep_scale<N>(d, (uint64_t)idx0, out);
  else {
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = 1.0f;
  }
}
// DROP_ON is the wave-uniform "this site drops" bit as a template argument: tested per element (`drop_on ? hash : 1`) it compiled
// to a scalar branch around every element's hash chain, 128 basic blocks per wave tile that nothing could be scheduled across.
// Output stores of the persistent NT kernels are nontemporal.  Measured (tools/experiments/nt_epi_burst.py, round 4, QKV 131072 x
// 1536 x 512 on the loader-wave kernel): main loop alone 183 us; + epilogue with its stores aimed at one L2-resident tile 208 us;
// + the real stores, default policy 252 us, sc1 243 us, nt 226 us (FFN-up shape: 354 / 333 / 299 us).  Holding the stores back and
// issuing them one per eighth of the next tile's k-steps changed nothing (260 us plain, 228 us nt): the cost is not the burst but the
// output lines themselves, which the write-back L2 keeps at the expense of the A panel and the weight tile.
#ifndef AFM_C_STORE_AUX
#define AFM_C_STORE_AUX 2
#endif

// 16-byte store of an output piece with a cache policy (aux: 0 plain, 2 nt, 16 sc1 = written through and dropped from the XCD's L2,
// MI355X_MICROARCH.md "stores of each flavour"): the output stream of a GEMM is never read again by this kernel, and left in the
// write-back L2 it evicts the weight tile and the A panels the other column tiles of the row are about to re-read.
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
template <int AUX>
__device__ __forceinline__ void store16_policy(void* base, uint64_t byte_off, uint4 v) {
  typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
  if constexpr (AUX == 0) *(uint4*)((char*)base + byte_off) = v;
  else if constexpr (AUX == 2) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, v), (u32x4_nt*)((char*)base + byte_off));
  else {   // (timing experiments only: 32-bit buffer offsets)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000);
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, (int)(uint32_t)byte_off, 0, AUX);
  }
}

// 16-byte load of a READ-ONCE epilogue operand (the stored gradient factors of EPI_MUL / EPI_GLU_BWD, the pre-activations of
// EPI_GELU_BWD): 537 MB per c2 launch that no workgroup reads twice.  Loaded with the default policy they allocate in the XCD's L2
// and evict the dY panel the row's other column tiles are about to re-read (round 4: 944-950 MB read per launch against 671 MB
// algorithmic, L2 hit 0.61); nontemporal marks the lines evict-first, like the output stores above.  AFM_PRE_LOAD_NT=0: A/B builds.
#ifndef AFM_PRE_LOAD_NT
#define AFM_PRE_LOAD_NT 1
#endif
__device__ __forceinline__ e16x8 load16_once(const e16* p) {
  typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
#if AFM_PRE_LOAD_NT
  return __builtin_bit_cast(e16x8, __builtin_nontemporal_load((const u32x4_nt*)p));
#else
  return *(const e16x8*)p;
#endif
}

template <int WM, int EPI, bool DROP_ON, int CAUX = 0>
__device__ __forceinline__ void epilogue_staged_e16(const MfmaArgs& g, float* stg, const float* bias_lds,
                                                    f32x4 (&acc)[4][WM], int mw, int nw, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  const int c8 = (lane & 7) * 8, r8 = lane >> 3;
  const int n = nw + c8;
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == EPI_GLU || EPI == EPI_GLU_SG) {   // u / v biases of hidden units (n >> 1) .. +3, reference order [b1 ; bg]
    if (g.bias) { b0 = *(const f32x4*)(g.bias + (n >> 1)); b1 = *(const f32x4*)(g.bias + g.glu_f + (n >> 1)); }
  } else if (EPI != EPI_GELU_BWD && EPI != EPI_MUL && EPI != EPI_GLU_BWD && g.bias) { b0 = *(const f32x4*)(bias_lds + n); b1 = *(const f32x4*)(bias_lds + n + 4); }
  constexpr bool drop_on = DROP_ON;
  e16* const cbase = (e16*)g.C + (int64_t)(mw + r8) * g.ldc + n;
  e16* const pbase = (e16*)g.pre_act + (int64_t)(mw + r8) * g.ldc + n;
  const uint32_t dbase = (uint32_t)(mw + r8) * (uint32_t)g.N + (uint32_t)n;
  // pre-activations of the dropout * GELU' form: loaded PF row-groups ahead of their use (all 2*WM at once
  // cost 8 VGPRs each: 64 at 128-row wave tiles, which spilled)
  constexpr int PF = 4;
  e16x8 uu[2 * WM];
  if (EPI == EPI_GELU_BWD || EPI == EPI_MUL) {
#pragma unroll
    for (int q = 0; q < (PF < 2 * WM ? PF : 2 * WM); ++q) uu[q] = load16_once(pbase + (int64_t)(q * 8) * g.ldc);
  }
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(stg + fr * STG_LD + j * 16 + fq * 4) = acc[j][i];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int q = i * 2 + hf;                       // 8-row group of the wave's 16*WM rows
      const int row = hf * 8 + r8;
      // (EPI_GLU_BWD: the lane takes hidden units 4 L .. + 3 and 32 + 4 L .. + 3 of the wave's 64, L = lane & 7 -- see below)
      constexpr bool GB = EPI == EPI_GLU_BWD;
      const f32x4 v0 = *(const f32x4*)(stg + row * STG_LD + (GB ? (c8 >> 1) : c8)) + b0;
      const f32x4 v1 = *(const f32x4*)(stg + row * STG_LD + (GB ? 32 + (c8 >> 1) : c8 + 4)) + b1;
      float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      const int64_t ro = (int64_t)(q * 8) * g.ldc;
      const uint32_t di = dbase + (uint32_t)(q * 8) * (uint32_t)g.N;
      if constexpr (EPI == EPI_GLU || EPI == EPI_GLU_SG) {
        // x[0..3] = u, x[4..7] = v of hidden units (n >> 1) .. +3; C and the dropout stream are f = N/2 wide
        const int64_t rowi = mw + r8 + q * 8;
        const int hcol = n >> 1;
        const uint32_t dg0 = (uint32_t)rowi * (uint32_t)(g.N >> 1) + (uint32_t)hcol;
        float gv[4], sa[4], sb[4], kp[4];
        afm_keep_scale32<4, drop_on>(g.dd, dg0, kp);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float y, yp;
          afm_gelu_both16(x[k], y, yp);
          const float keep = kp[k];
          gv[k] = y * x[4 + k] * keep; sa[k] = yp * x[4 + k] * keep; sb[k] = y * keep;
        }
        e16x4 o = {(e16)gv[0], (e16)gv[1], (e16)gv[2], (e16)gv[3]};
        if constexpr (CAUX != 0) __builtin_nontemporal_store(__builtin_bit_cast(uint2_t, o), (uint2_t*)((e16*)g.C + rowi * g.ldc + hcol));
        else *(e16x4*)((e16*)g.C + rowi * g.ldc + hcol) = o;
        if (EPI == EPI_GLU_SG) {
          e16x8 sv = {(e16)sa[0], (e16)sa[1], (e16)sa[2], (e16)sa[3], (e16)sb[0], (e16)sb[1], (e16)sb[2], (e16)sb[3]};
          store16_policy<CAUX>(g.pre_act, (uint64_t)((rowi * g.N + n) * 2), __builtin_bit_cast(uint4, sv));
        }
        continue;
      }
      if constexpr (EPI == EPI_GLU_BWD) {
        // x[0..3] = dg of hidden units nw + 4 L .. + 3, x[4..7] of nw + 32 + 4 L .. + 3 (L = lane & 7): one interleave group each, i.e. the
        // 16-byte pieces 8 L and 64 + 8 L of the wave's 128 saved / output columns 2 nw .. 2 nw + 127 (2f = ldc wide).  Round 5: with a
        // lane on hidden units n .. n + 7 its two pieces were NEIGHBOURS (32 bytes per lane), so each of the two read-once loads touched
        // every 64-byte granule of the row and HBM delivered the 1.6-GB factor tensor twice (3.6 GB read for 1.8, profiles/r05_*); this way
        // a load instruction's eight lanes cover 128 contiguous bytes of a row, and so does a store.
        const int64_t rowi = mw + r8 + q * 8;
        const e16* sp = (const e16*)g.pre_act + rowi * g.ldc + 2 * nw + c8;
        e16* cp = (e16*)g.C + rowi * g.ldc + 2 * nw + c8;
        const e16x8 s0 = load16_once(sp), s1 = load16_once(sp + 64);
        e16x8 o0, o1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o0[k] = (e16)(x[k] * (float)s0[k]); o0[4 + k] = (e16)(x[k] * (float)s0[4 + k]);
          o1[k] = (e16)(x[4 + k] * (float)s1[k]); o1[4 + k] = (e16)(x[4 + k] * (float)s1[4 + k]);
        }
        store16_policy<CAUX>(g.C, (uint64_t)((cp - (e16*)g.C) * 2), __builtin_bit_cast(uint4, o0));
        store16_policy<CAUX>(g.C, (uint64_t)((cp + 64 - (e16*)g.C) * 2), __builtin_bit_cast(uint4, o1));
        continue;
      }
      if (EPI == EPI_GELU) {
        if (g.pre_act) {
          e16x8 o = {(e16)x[0], (e16)x[1], (e16)x[2], (e16)x[3], (e16)x[4], (e16)x[5], (e16)x[6], (e16)x[7]};
          store16_policy<CAUX>(g.pre_act, (uint64_t)((pbase + ro - (e16*)g.pre_act) * 2), __builtin_bit_cast(uint4, o));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = afm_gelu16(x[k]);
      }
      if (EPI == EPI_GELU_SG) {
        float gp[8], kp[8];
        afm_keep_scale32<8, drop_on>(g.dd, di, kp);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float y, yp;
          afm_gelu_both16(x[k], y, yp);
          x[k] = y * kp[k]; gp[k] = yp * kp[k];
        }
        e16x8 o = {(e16)gp[0], (e16)gp[1], (e16)gp[2], (e16)gp[3], (e16)gp[4], (e16)gp[5], (e16)gp[6], (e16)gp[7]};
        store16_policy<CAUX>(g.pre_act, (uint64_t)((pbase + ro - (e16*)g.pre_act) * 2), __builtin_bit_cast(uint4, o));
      }
      if (EPI == EPI_MUL) {
        const e16x8 u = uu[q];
        if (q + PF < 2 * WM) uu[q + PF < 2 * WM ? q + PF : 0] = load16_once(pbase + (int64_t)((q + PF) * 8) * g.ldc);
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] *= (float)u[k];
      }
      if (EPI == EPI_GELU_BWD) {
        const e16x8 u = uu[q];
        if (q + PF < 2 * WM) uu[q + PF < 2 * WM ? q + PF : 0] = load16_once(pbase + (int64_t)((q + PF) * 8) * g.ldc);
        if (drop_on) {
          float kp[8];
          afm_keep_scale32<8, true>(g.dd, di, kp);
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = x[k] * kp[k] * afm_gelu_grad16((float)u[k]);
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] *= afm_gelu_grad16((float)u[k]);
        }
      } else if ((EPI == EPI_DROP || EPI == EPI_GELU) && drop_on) {
        float kp[8];
        afm_keep_scale32<8, true>(g.dd, di, kp);
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] *= kp[k];
      }
      e16x8 o = {(e16)x[0], (e16)x[1], (e16)x[2], (e16)x[3], (e16)x[4], (e16)x[5], (e16)x[6], (e16)x[7]};
      store16_policy<CAUX>(g.C, (uint64_t)((cbase + ro - (e16*)g.C) * 2), __builtin_bit_cast(uint4, o));
    }
  }
}

template <int WM, int EPI, int CAUX = 0>
__device__ __forceinline__ void epilogue_staged_bf16(const MfmaArgs& g, float* stg, const float* bias_lds,
                                                     f32x4 (&acc)[4][WM], int mw, int nw, int lane) {
  constexpr bool drops = EPI == EPI_DROP || EPI == EPI_GELU || EPI == EPI_GELU_BWD || EPI == EPI_GELU_SG || EPI == EPI_GLU || EPI == EPI_GLU_SG;
  if (drops && g.dd.thresh != 0) epilogue_staged_e16<WM, EPI, true, CAUX>(g, stg, bias_lds, acc, mw, nw, lane);
  else epilogue_staged_e16<WM, EPI, false, CAUX>(g, stg, bias_lds, acc, mw, nw, lane);
}

// ------------------------------------------------------------------------------------------ NT: padded row tiles
// afm_gemm_desc.k_live for the NT form (A's rows are token positions; no bias / residual / accumulate, act NONE or x saved): a tile
// whose TBM rows of A are all padding has a zero product.  Each workgroup splits the tiles of its strided walk into a live list (the
// ring and the main loop only ever see those) and a dead list, whose C tiles it writes as zeros before the ring starts.
#define NT_LIVE_MAX 120
#define NT_LIVE_BYTES (4 * (2 * NT_LIVE_MAX + 2))
template <int TBM, int TBN, bool C16, int NTHREADS, int EPI = EPI_PLAIN>      // EPI: the kernel's epilogue kind (which columns of C / pre_act a tile covers)
__device__ __forceinline__ void nt_tile_lists(const MfmaArgs& g, int* list, int tlo, int thi, int nbx, int bx) {
  const int t = threadIdx.x, lane = t & 63;
  int* dead = list + 1 + NT_LIVE_MAX;        // list[0] / dead[0] = counts
  if (t < 64) {
    // every workgroup of the XCD scans the XCD's whole tile range and keeps every nbx-th LIVE tile (and every nbx-th dead one):
    // with the plain strided walk the workgroups that own the tails of the sequences would get nothing but dead tiles
    int nl = 0, nd = 0, il = 0, id = 0;      // kept so far; live / dead tiles seen so far
    for (int t0 = tlo; t0 < thi; t0 += 64) {
      const int tt = t0 + lane;
      const bool valid = tt < thi;
      bool live = false;
      if (valid) {
        int mt_, nt_;
        tile_mn(g, tt, mt_, nt_);
        const int mb = mt_ * (TBM / 64);
#pragma unroll
        for (int i = 0; i < TBM / 64; ++i) live |= g.k_live[mb + i] != 0;
      }
      const unsigned long long bl = __ballot(valid && live), bd = __ballot(valid && !live), lt = (1ull << lane) - 1ull;
      const int li = il + __popcll(bl & lt), di = id + __popcll(bd & lt);        // this tile's index among the live / dead ones
      const bool mine_l = valid && live && (li % nbx) == bx, mine_d = valid && !live && (di % nbx) == bx;
      const unsigned long long ml = __ballot(mine_l), md = __ballot(mine_d);
      if (mine_l) list[1 + nl + __popcll(ml & lt)] = tt;
      if (mine_d) dead[1 + nd + __popcll(md & lt)] = tt;
      nl += __popcll(ml); nd += __popcll(md);
      il += __popcll(bl); id += __popcll(bd);
    }
    if (lane == 0) { list[0] = nl; dead[0] = nd; }
  }
  __syncthreads();
  const int nd = g.dead_nofill ? 0 : dead[0];
  // columns of C one tile covers: the gated data-gradient form writes 2 x TBN interleaved columns, the gated forward forms TBN / 2
  constexpr bool GLU_F = EPI == EPI_GLU || EPI == EPI_GLU_SG, GLU_B = EPI == EPI_GLU_BWD;
  constexpr int EB = C16 ? 2 : 4, PER = 16 / EB;
  constexpr int CW = GLU_B ? 2 * TBN : GLU_F ? TBN / 2 : TBN, CPR = CW / PER;   // 16-byte chunks per tile row
  const int cn = GLU_B ? 2 * g.N : GLU_F ? g.N / 2 : g.N;
  for (int i = 0; i < nd; ++i) {
    const int tt = dead[1 + i];
    int mt_, nt_;
    tile_mn(g, tt, mt_, nt_);
    const int m0 = mt_ * TBM, n0 = nt_ * CW;
    for (int c = t; c < TBM * CPR; c += NTHREADS) {
      const int r = c / CPR, cc = (c % CPR) * PER;
      if (n0 + cc < cn) *(uint4*)((char*)g.C + ((int64_t)(m0 + r) * g.ldc + n0 + cc) * EB) = make_uint4(0u, 0u, 0u, 0u);
    }
    if (g.dead_pre) {      // the forward forms that store a second tensor (pre-activations / backward factors: TBN columns per tile, N wide)
      constexpr int PPR = TBN / PER;
      const int64_t ldp = EPI == EPI_GLU_SG ? (int64_t)g.N : (int64_t)g.ldc;      // EPI_GLU_SG writes its factors densely (M x N)
      for (int c = t; c < TBM * PPR; c += NTHREADS) {
        const int r = c / PPR, cc = (c % PPR) * PER;
        if (nt_ * TBN + cc < g.N) *(uint4*)((char*)g.pre_act + ((int64_t)(m0 + r) * ldp + nt_ * TBN + cc) * EB) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ NT, persistent ring
// One workgroup per CU walks a contiguous range of output tiles of its XCD; the LDS-DMA ring keeps
// running ACROSS tiles (the first k-steps of the next tile are in flight while the current tile
// finishes and its epilogue stores drain), so the per-tile prologue bubble and the workgroup
// launch/teardown disappear from the critical path.
// ABL (timing experiments only, bit mask): 1 = skip the LDS reads + MFMAs, 2 = skip the LDS-DMA loads,
// 4 = skip the epilogue.
// EPI: compile-time epilogue kind of full tiles (e16 output); EDGE = false drops the fragment epilogue of
// partial tiles (the dispatcher then only sends shapes made of whole tiles).
template <bool C_BF16, int NWM, int NWN, int S, int ABL = 0, int WM = 4, int EPI = EPI_GENERIC, bool EDGE = true>
__global__ __launch_bounds__(64 * NWM * NWN) void k_gemm_nt_pring(MfmaArgs g) {
  constexpr int NW = NWM * NWN;
  constexpr int TBM = 16 * WM * NWM, TBN = 64 * NWN;
  constexpr int NI = (TBM + TBN) / 8, NIW = NI / NW;
  static_assert(NI % NW == 0, "pieces must divide over the waves");
  constexpr int STAGE = (TBM + TBN) * 128;
  static_assert(NW * 16 * STG_LD * 4 <= STAGE, "wave-private staging patches must fit one ring slot");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* bias_lds = (float*)(lds + S * STAGE);   // whole bias vector, loaded once (g.bias_in_lds)
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w / NWN, wn = w % NWN;
  const int ntiles = g.tiles_m * g.tiles_n;
  constexpr bool GLU_EPI = EPI == EPI_GLU || EPI == EPI_GLU_SG || EPI == EPI_GLU_BWD;   // these read the bias from global memory
  if (g.bias_in_lds && !GLU_EPI) {   // plain loads, retired (barrier) before the first LDS-DMA piece is issued
    for (int n = t; n < g.N; n += 64 * NW) bias_lds[n] = g.bias ? g.bias[n] : 0.f;
    __syncthreads();
  }
  // XCD x owns tiles [x*tpx, (x+1)*tpx); its blocks (blockIdx % 8 == x) stride through them together
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const int tpx = (ntiles + 7) >> 3;
  const int tlo = xcd * tpx, thi = min(ntiles, tlo + tpx);
  int* const tlist = (int*)(lds + g.live_off);
  if (g.live_off) nt_tile_lists<TBM, TBN, C_BF16, 64 * NW, EPI>(g, tlist, tlo, thi, nbx, bx);
  auto tile_of = [&](int it) {
    if (g.live_off) return it < tlist[0] ? tlist[1 + it] : -1;
    const int tt = tlo + it * nbx + bx;
    return tt < thi ? tt : -1;
  };
  const int nk = g.K / 64;

  const e16* src[NIW];
  auto set_src = [&](int tile) {
    int mt_, nt_;
    tile_mn(g, tile, mt_, nt_);
    const int m0 = mt_ * TBM, n0 = nt_ * TBN;
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
      const int ii = w + NW * j;
      const int r8 = lane >> 3, ch = (lane & 7) ^ r8;
      if (ii < TBM / 8) src[j] = g.A + (int64_t)min(m0 + ii * 8 + r8, g.M - 1) * g.lda + ch * 8;
      else src[j] = g.B + (int64_t)min(n0 + (ii - TBM / 8) * 8 + r8, g.N - 1) * g.ldb + ch * 8;
    }
  };
  int is_it = 0, is_kt = 0, is_slot = 0;       // next step to issue: tile iteration, k-step, ring slot
  int is_tile = tile_of(0);
  if (is_tile >= 0) set_src(is_tile);
  int ahead = 0;                                // steps issued but not yet consumed
  auto issue_one = [&]() {
    if (is_tile < 0) return;
    unsigned char* st = lds + is_slot * STAGE;
    if (!(ABL & 2)) {
#pragma unroll
      for (int j = 0; j < NIW; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + is_kt * 64),
                                         (__attribute__((address_space(3))) void*)(st + (w + NW * j) * 1024), 16, 0, 0);
    }
    ++ahead;
    is_slot = is_slot + 1 == S ? 0 : is_slot + 1;
    if (++is_kt == nk) {
      is_kt = 0;
      is_tile = tile_of(++is_it);
      if (is_tile >= 0) set_src(is_tile);
    }
  };
  auto off = [](int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); };
#pragma unroll
  for (int s = 0; s < S - 1; ++s) issue_one();

  const int fr = lane & 15, fq = lane >> 4;
  const bool vec_ok = (g.ldc & 3) == 0 && (g.N & 3) == 0;
  int slot = 0;
  bool prev_full = false;
  for (int it = 0;; ++it) {
    const int tile = tile_of(it);
    if (tile < 0) break;
    int mt_, nt_;
    tile_mn(g, tile, mt_, nt_);
    const int m0 = mt_ * TBM, n0 = nt_ * TBN;
#ifdef AFM_GEMM_ABLATIONS
    if (g.stamps && t == 0 && it < 16) g.stamps[(blockIdx.x * 16 + it) * 4 + 0] = wall_clock64();
#endif
    f32x4 acc[4][WM];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < WM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
      // `ahead` counts the current step too; (ahead-1)*NIW younger pieces may stay in flight.  The
      // previous tile's epilogue sits in the same in-order counter BEHIND the pieces of this tile's
      // first S-1 steps: after a full-tile epilogue (exactly 8 / 16 stores per wave for e16 / fp32
      // output, plus loads) those younger operations may stay outstanding too, so the stores drain under the next tile's
      // MFMAs; after an edge-tile epilogue (store count unknown) drain everything.
#ifdef AFM_GEMM_ABLATIONS
      const bool stampk = g.stamps && lane == 0 && it == 5 && blockIdx.x < 64;
      unsigned long long* sk = g.stamps + 512 * 64 + ((blockIdx.x * 12 + w) * 8 + kt) * 5;
      if (stampk) sk[0] = clock64();
#endif
      if (it > 0 && kt < S - 1 && !prev_full) wait_vmcnt<0>();
      else if (it > 0 && kt < S - 1 && ahead - 1 >= S - 2) wait_vmcnt<NIW * (S - 2) + (C_BF16 ? 2 * WM : 4 * WM)>();
      else if (ahead - 1 >= S - 2) wait_vmcnt<NIW * (S - 2)>();
      else if (S > 3 && ahead - 1 == S - 3) wait_vmcnt<NIW * (S > 3 ? S - 3 : 0)>();
      else wait_vmcnt<0>();
#ifdef AFM_GEMM_ABLATIONS
      if (stampk) sk[1] = clock64();
#endif
      __builtin_amdgcn_s_barrier();
#ifdef AFM_GEMM_ABLATIONS
      if (stampk) sk[2] = clock64();
#endif
      --ahead;
      issue_one();
#ifdef AFM_GEMM_ABLATIONS
      if (stampk) sk[3] = clock64();
#endif
      const unsigned char* a = lds + slot * STAGE;
      const unsigned char* b = a + TBM * 128;
      slot = slot + 1 == S ? 0 : slot + 1;
      if (ABL & 1) continue;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        e16x8 af[WM], bfr[4];
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = *(const e16x8*)(a + off(wm * 16 * WM + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(const e16x8*)(b + off(wn * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < WM; ++i)
            acc[j][i] = mfma16(bfr[j], af[i], acc[j][i]);
      }
#ifdef AFM_GEMM_ABLATIONS
      if (stampk) sk[4] = clock64();
#endif
    }
#ifdef AFM_GEMM_ABLATIONS
    if (g.stamps && t == 0 && it < 16) g.stamps[(blockIdx.x * 16 + it) * 4 + 1] = wall_clock64();
#endif
    prev_full = g.bias_in_lds && m0 + TBM <= g.M && n0 + TBN <= g.N;
    if (ABL & 4) {
      if (acc[0][0][0] == 123.456f) ((float*)g.C)[0] = 1.f;   // keep the accumulators alive
      prev_full = false;
    } else if (prev_full) {
      // the slot read by the last k-step is free until the next issue: stage through it
      __builtin_amdgcn_s_barrier();
      float* stg = (float*)(lds + (slot == 0 ? S - 1 : slot - 1) * STAGE) + w * (16 * STG_LD);
      if constexpr (C_BF16 && EPI != EPI_GENERIC) epilogue_staged_bf16<WM, EPI, AFM_C_STORE_AUX>(g, stg, bias_lds, acc, m0 + wm * 16 * WM, n0 + wn * 64, lane);
      else epilogue_staged<C_BF16, WM>(g, stg, bias_lds, acc, m0 + wm * 16 * WM, n0 + wn * 64, lane);
    } else if constexpr (EDGE) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i)
          epilogue4<C_BF16>(g, m0 + wm * 16 * WM + i * 16 + fr, n0 + wn * 64 + j * 16 + fq * 4, acc[j][i], vec_ok);
    }
#ifdef AFM_GEMM_ABLATIONS
    if (g.stamps && t == 0 && it < 16) g.stamps[(blockIdx.x * 16 + it) * 4 + 2] = wall_clock64();
#endif
  }
}

template <bool C_BF16, int NWM, int NWN, int S, int ABL = 0, int WM = 4, int EPI = EPI_GENERIC, bool EDGE = true>
static int launch_nt_pring(MfmaArgs& g, hipStream_t st, int blocks_per_cu) {
  constexpr int TBM = 16 * WM * NWM, TBN = 64 * NWN;
  constexpr int ring = S * (TBM + TBN) * 128;
  static_assert(ring <= 160 * 1024, "ring does not fit the 160 KiB LDS");
  g.tiles_m = (g.M + TBM - 1) / TBM; g.tiles_n = (g.N + TBN - 1) / TBN;
  g.xgc = (g.M % TBM == 0 && g.N % TBN == 0 && blocks_per_cu == 1) ? nt_pick_xgc(g.tiles_m, g.tiles_n, (int64_t)g.N * g.K * 2) : 1;
  // staged epilogue: needs 16-byte rows everywhere and room for the bias vector behind the ring
  const int bias_bytes = ((g.N * 4 + 15) / 16) * 16;
  const bool rows16 = (g.N % 8) == 0 && (g.ldc % 8) == 0;
  const bool modes_ok = C_BF16 ? (!g.residual && !g.accumulate) : true;
  constexpr bool GLU_EPI = EPI == EPI_GLU || EPI == EPI_GLU_SG || EPI == EPI_GLU_BWD;
  g.bias_in_lds = rows16 && modes_ok && (GLU_EPI || ring * blocks_per_cu + bias_bytes * blocks_per_cu <= 160 * 1024) ? 1 : 0;
  int shm = ring + (g.bias_in_lds && !GLU_EPI ? bias_bytes : 0);
  g.live_off = 0;
  if (g.k_live && TBM % 64 == 0 && (g.M % TBM) == 0 && blocks_per_cu == 1 && shm + NT_LIVE_BYTES <= 160 * 1024) {
    int grid0 = 256;
    const int nt0 = g.tiles_m * g.tiles_n;
    if (grid0 > ((nt0 + 7) / 8) * 8) grid0 = ((nt0 + 7) / 8) * 8;
    const int tpx0 = (nt0 + 7) / 8, nbx0 = grid0 / 8;
    if ((tpx0 + nbx0 - 1) / nbx0 <= NT_LIVE_MAX) { g.live_off = shm; shm += NT_LIVE_BYTES; }
  }
  g.mperm = (g.live_off && g.deal && !(g.tiles_m & 7)) ? g.tiles_m >> 3 : 0;
  if (g.k_live && g.live_off) afm_note_hint(1);
  auto kern = k_gemm_nt_pring<C_BF16, NWM, NWN, S, ABL, WM, EPI, EDGE>;
  static AfmOncePerDevice attr_shm;   // per instantiation and per device (function attributes are per device)
  if (attr_shm.need()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int grid = 256 * blocks_per_cu;                      // 256 CUs; multiple of 8 (XCD ranges)
  const int ntiles = g.tiles_m * g.tiles_n;
  if (grid > ((ntiles + 7) / 8) * 8) grid = ((ntiles + 7) / 8) * 8;
  AFM_LAUNCH(kern, dim3(grid), dim3(64 * NWM * NWN), shm, st, g);
  return AFM_OK;
}

// ------------------------------------------------------------------------------------------ NT, loader waves
// Same tile walk and LDS ring as k_gemm_nt_pring, but the LDS-DMA pieces are issued by NL extra
// "loader" waves.  Measured on the all-waves-issue kernel (in-kernel stamps, tools/stamp_gemm.py): a
// global_load_lds wave-instruction holds its wave for ~20 ns (the CU takes in ~50 GB/s, one 1-KiB
// piece at a time), so the 48 pieces of a k-step cost every compute wave 0.5-1 us of issue time in
// front of 0.95 us of LDS reads + MFMAs, and the per-step barrier makes all of them wait for the
// last issuer: fill and compute ran strictly one after the other (2.5 us per k-step).  Loader waves
// take the issue time (and the counted vmcnt waits) off the MFMA waves; the workgroup barrier of each
// k-step publishes a landed slot and frees the one read a step earlier.  The compute waves issue no
// LDS-DMA at all, so their epilogue stores need no counted waits.
template <bool C_BF16, int NL, int ABL = 0, int EPI = EPI_GENERIC, int CAUX = AFM_C_STORE_AUX>
__global__ __launch_bounds__(64 * (8 + NL)) void k_gemm_nt_ws(MfmaArgs g) {
  constexpr int NWN = 2, NW = 8, WM = 4, S = 3;
  constexpr int TBM = 256, TBN = 128;
  constexpr int NI = (TBM + TBN) / 8, NIL = NI / NL;   // 1-KiB pieces per k-step, per loader wave
  static_assert(NI % NL == 0, "pieces must divide over the loader waves");
  constexpr int STAGE = (TBM + TBN) * 128;
  static_assert(NW * 16 * STG_LD * 4 <= STAGE, "wave-private staging patches must fit one ring slot");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* bias_lds = (float*)(lds + S * STAGE);
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntiles = g.tiles_m * g.tiles_n;
  constexpr bool GLU_EPI = EPI == EPI_GLU || EPI == EPI_GLU_SG || EPI == EPI_GLU_BWD;   // these read the bias from global memory
  if (g.bias_in_lds && !GLU_EPI) {   // plain loads, retired (barrier) before the first LDS-DMA piece is issued
    for (int n = t; n < g.N; n += 64 * (NW + NL)) bias_lds[n] = g.bias ? g.bias[n] : 0.f;
    __syncthreads();
  }
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const int tpx = (ntiles + 7) >> 3;
  const int tlo = xcd * tpx, thi = min(ntiles, tlo + tpx);
  int* const tlist = (int*)(lds + g.live_off);
  if (g.live_off) nt_tile_lists<TBM, TBN, C_BF16, 64 * (NW + NL), EPI>(g, tlist, tlo, thi, nbx, bx);
  auto tile_of = [&](int it) {
    if (g.live_off) return it < tlist[0] ? tlist[1 + it] : -1;
    const int tt = tlo + it * nbx + bx;
    return tt < thi ? tt : -1;
  };
  auto tile_full = [&](int tile) {
    int mt_, nt_;
    tile_mn(g, tile, mt_, nt_);
    const int m0 = mt_ * TBM, n0 = nt_ * TBN;
    return g.bias_in_lds && m0 + TBM <= g.M && n0 + TBN <= g.N;
  };
  const int nk = g.K / 64;

  if (w >= NW) {
    // ---------------------------------------------------------------- loader wave
    const int lw = w - NW;
    const e16* src[NIL];
    auto set_src = [&](int tile) {
      int mt_, nt_;
    tile_mn(g, tile, mt_, nt_);
    const int m0 = mt_ * TBM, n0 = nt_ * TBN;
#pragma unroll
      for (int j = 0; j < NIL; ++j) {
        const int ii = lw + NL * j;
        const int r8 = lane >> 3, ch = (lane & 7) ^ r8;
        if (ii < TBM / 8) src[j] = g.A + (int64_t)min(m0 + ii * 8 + r8, g.M - 1) * g.lda + ch * 8;
        else src[j] = g.B + (int64_t)min(n0 + (ii - TBM / 8) * 8 + r8, g.N - 1) * g.ldb + ch * 8;
      }
    };
    int is_it = 0, is_kt = 0, is_slot = 0;
    int is_tile = tile_of(0);
    if (is_tile >= 0) set_src(is_tile);
    int ahead = 0;   // steps issued and not yet published
    auto issue_one = [&]() {
      if (is_tile < 0) return;
      unsigned char* st = lds + is_slot * STAGE;
      if (!(ABL & 2)) {
#pragma unroll
        for (int j = 0; j < NIL; ++j)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + is_kt * 64),
                                           (__attribute__((address_space(3))) void*)(st + (lw + NL * j) * 1024), 16, 0, 0);
      }
      ++ahead;
      is_slot = is_slot + 1 == S ? 0 : is_slot + 1;
      if (++is_kt == nk) {
        is_kt = 0;
        is_tile = tile_of(++is_it);
        if (is_tile >= 0) set_src(is_tile);
      }
    };
#pragma unroll
    for (int s = 0; s < S - 1; ++s) issue_one();
    for (int it = 0;; ++it) {
      const int tile = tile_of(it);
      if (tile < 0) break;
      for (int kt = 0; kt < nk; ++kt) {
        // publish this step: its pieces must have landed; the step issued after it may stay in flight
#ifdef AFM_GEMM_ABLATIONS
        const bool stampk = g.stamps && lane == 0 && it == 5 && blockIdx.x < 64 && w < 12;
        unsigned long long* sk = g.stamps + 512 * 64 + ((blockIdx.x * 12 + w) * 8 + kt) * 5;
        if (stampk) sk[0] = clock64();
#endif
        if (ahead - 1 >= S - 2) wait_vmcnt<NIL*(S - 2)>(); else wait_vmcnt<0>();
#ifdef AFM_GEMM_ABLATIONS
        if (stampk) sk[1] = clock64();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef AFM_GEMM_ABLATIONS
        if (stampk) sk[2] = clock64();
#endif
        --ahead;
        issue_one();   // into the slot every compute wave finished reading before this barrier
#ifdef AFM_GEMM_ABLATIONS
        if (stampk) { sk[3] = clock64(); sk[4] = sk[3]; }
#endif
      }
      if (tile_full(tile)) __builtin_amdgcn_s_barrier();   // the compute waves' pre-epilogue barrier
    }
    return;
  }

  // ------------------------------------------------------------------ compute wave
  const int wm = w / NWN, wn = w % NWN;
  auto off = [](int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); };
  const int fr = lane & 15, fq = lane >> 4;
  const bool vec_ok = (g.ldc & 3) == 0 && (g.N & 3) == 0;
  int slot = 0;
  for (int it = 0;; ++it) {
    const int tile = tile_of(it);
    if (tile < 0) break;
    int mt_, nt_;
    tile_mn(g, tile, mt_, nt_);
    const int m0 = mt_ * TBM, n0 = nt_ * TBN;
    f32x4 acc[4][WM];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < WM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef AFM_GEMM_ABLATIONS
    if (g.stamps && t == 0 && it < 16) g.stamps[(blockIdx.x * 16 + it) * 4 + 0] = wall_clock64();
#endif
    for (int kt = 0; kt < nk; ++kt) {
#ifdef AFM_GEMM_ABLATIONS
      const bool stampk = g.stamps && lane == 0 && it == 5 && blockIdx.x < 64;
      unsigned long long* sk = g.stamps + 512 * 64 + ((blockIdx.x * 12 + w) * 8 + kt) * 5;
      if (stampk) { sk[0] = clock64(); sk[1] = sk[0]; }
#endif
      __builtin_amdgcn_s_barrier();
#ifdef AFM_GEMM_ABLATIONS
      if (stampk) { sk[2] = clock64(); sk[3] = sk[2]; }
#endif
      const unsigned char* a = lds + slot * STAGE;
      const unsigned char* b = a + TBM * 128;
      slot = slot + 1 == S ? 0 : slot + 1;
      if (ABL & 1) continue;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        e16x8 af[WM], bfr[4];
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = *(const e16x8*)(a + off(wm * 16 * WM + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(const e16x8*)(b + off(wn * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < WM; ++i)
            acc[j][i] = mfma16(bfr[j], af[i], acc[j][i]);
      }
#ifdef AFM_GEMM_ABLATIONS
      if (stampk) sk[4] = clock64();
#endif
    }
#ifdef AFM_GEMM_ABLATIONS
    if (g.stamps && t == 0 && it < 16) g.stamps[(blockIdx.x * 16 + it) * 4 + 1] = wall_clock64();
#endif
    if (ABL & 4) {   // timing only: keep every accumulator alive, store nothing
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i) sacc += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
      if (sacc == 123.456f) ((float*)g.C)[0] = sacc;
      if (tile_full(tile)) __builtin_amdgcn_s_barrier();
    } else if (tile_full(tile)) {
      // the slot read by the last k-step stays free until the next step's barrier: stage through it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      float* stg = (float*)(lds + (slot == 0 ? S - 1 : slot - 1) * STAGE) + w * (16 * STG_LD);
      // ABL & 8 (timing only): every tile's epilogue writes tile 0 (L2-resident): the epilogue's arithmetic without its HBM stores
      const int em0 = (ABL & 8) ? 0 : m0, en0 = (ABL & 8) ? 0 : n0;
      if constexpr (C_BF16 && EPI != EPI_GENERIC) epilogue_staged_bf16<WM, EPI, CAUX>(g, stg, bias_lds, acc, em0 + wm * 16 * WM, en0 + wn * 64, lane);
      else epilogue_staged<C_BF16, WM>(g, stg, bias_lds, acc, em0 + wm * 16 * WM, en0 + wn * 64, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging reads done before the slot is handed back
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i)
          epilogue4<C_BF16>(g, m0 + wm * 16 * WM + i * 16 + fr, n0 + wn * 64 + j * 16 + fq * 4, acc[j][i], vec_ok);
    }
#ifdef AFM_GEMM_ABLATIONS
    if (g.stamps && t == 0 && it < 16) g.stamps[(blockIdx.x * 16 + it) * 4 + 2] = wall_clock64();
#endif
  }
}

template <bool C_BF16, int NL, int ABL = 0, int EPI = EPI_GENERIC, int CAUX = AFM_C_STORE_AUX>
static int launch_nt_ws(MfmaArgs& g, hipStream_t st) {
  constexpr int TBM = 256, TBN = 128, S = 3;
  constexpr int ring = S * (TBM + TBN) * 128;
  g.tiles_m = (g.M + TBM - 1) / TBM; g.tiles_n = (g.N + TBN - 1) / TBN;
  g.xgc = (g.M % TBM == 0 && g.N % TBN == 0) ? nt_pick_xgc(g.tiles_m, g.tiles_n, (int64_t)g.N * g.K * 2) : 1;
  const int bias_bytes = ((g.N * 4 + 15) / 16) * 16;
  const bool rows16 = (g.N % 8) == 0 && (g.ldc % 8) == 0;
  const bool modes_ok = C_BF16 ? (!g.residual && !g.accumulate) : true;
  constexpr bool GLU_EPI = EPI == EPI_GLU || EPI == EPI_GLU_SG || EPI == EPI_GLU_BWD;
  g.bias_in_lds = rows16 && modes_ok && (GLU_EPI || ring + bias_bytes <= 160 * 1024) ? 1 : 0;
  int shm = ring + (g.bias_in_lds && !GLU_EPI ? bias_bytes : 0);
  g.live_off = 0;
  if (g.k_live && (g.M % TBM) == 0 && shm + NT_LIVE_BYTES <= 160 * 1024) {
    int grid0 = 256;
    const int nt0 = g.tiles_m * g.tiles_n;
    if (grid0 > ((nt0 + 7) / 8) * 8) grid0 = ((nt0 + 7) / 8) * 8;
    const int tpx0 = (nt0 + 7) / 8, nbx0 = grid0 / 8;
    if ((tpx0 + nbx0 - 1) / nbx0 <= NT_LIVE_MAX) { g.live_off = shm; shm += NT_LIVE_BYTES; }
  }
  g.mperm = (g.live_off && g.deal && !(g.tiles_m & 7)) ? g.tiles_m >> 3 : 0;
  if (g.k_live && g.live_off) afm_note_hint(1);
  auto kern = k_gemm_nt_ws<C_BF16, NL, ABL, EPI, CAUX>;
  static AfmOncePerDevice attr_done;   // per instantiation
  if (attr_done.need()) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  int grid = 256;
  const int ntiles = g.tiles_m * g.tiles_n;
  if (grid > ((ntiles + 7) / 8) * 8) grid = ((ntiles + 7) / 8) * 8;
#ifdef AFM_GEMM_ABLATIONS
  { const char* eg = getenv("AFM_GRID"); if (eg) grid = atoi(eg); }
#endif
  AFM_LAUNCH(kern, dim3(grid), dim3(64 * (8 + NL)), shm, st, g);
  return AFM_OK;
}

#include "afm_gemm_pp_impl.h"
#include "afm_gemm_w4_impl.h"

// ------------------------------------------------------------------------------------------ TN (wgrad)
// C[m][n] += sum_k A[k][m] B[k][n]: A is dy (rows = tokens, cols = output features m), B is x
// (rows = tokens, cols = input features n).  LDS tile [64 k-rows][128 cols] e16 = 256-byte rows of
// 16 chunks; chunk c of row r is stored at chunk c ^ s(r), s(r) = 2*(r&3) + 8*((r>>3)&1): a
// ds_read_b64_tr_b16 half-wave (2 groups x 4 rows x 4 column quads) then covers all 64 banks once.
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }
__device__ __forceinline__ int tn_off(int row, int chunk) { return row * 256 + ((chunk ^ tn_swz(row)) << 4); }

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ e16x4 ds_read_tr(const unsigned char* p) {
  // ds_read_b64_tr_b16 through the compiler builtin, so hipcc schedules and counts it (lgkmcnt)
  const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(e16x4, r);
}

__global__ __launch_bounds__(256) void k_gemm_tn(MfmaArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * BK * BM * 2];  // [buf][A|B][64][128]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int ntile = g.tiles_m * g.tiles_n;
  const int bid = xcd_remap(blockIdx.x, ntile * g.ksplit);
  // tile index fastest: the workgroups of one XCD share a k-chunk, so the dy / x rows they stream are
  // fetched from HBM once and served to the other tiles from that XCD's L2
  const int tile = bid % ntile, ks_id = bid / ntile;
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
  // bias gradient: the blocks of the first column tile also sum the dy rows they stage
  const bool do_cs = g.a_colsum != nullptr && (tile % g.tiles_n) == 0;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto sstore = [&](int buf) {
    unsigned char* a = lds + buf * (2 * BK * BM * 2);
    unsigned char* b = a + BK * BM * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = srow + 16 * i;
      *(uint4*)(a + tn_off(r, sch)) = ra_[i];
      *(uint4*)(b + tn_off(r, sch)) = rb_[i];
      if (do_cs) {
        const e16x8 v = __builtin_bit_cast(e16x8, ra_[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += (float)v[j];
      }
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
      e16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ca = wm * 64 + i * 16, cb = wn * 64 + i * 16;  // first column of the 16-wide fragment
        const int r0 = ks * 32 + grp * 8 + q, r1 = r0 + 4;
        const int cha = (ca >> 3) + (p >> 1), chb = (cb >> 3) + (p >> 1);
        const e16x4 a0 = ds_read_tr(a + tn_off(r0, cha) + ((p & 1) << 3));
        const e16x4 a1 = ds_read_tr(a + tn_off(r1, cha) + ((p & 1) << 3));
        const e16x4 b0 = ds_read_tr(b + tn_off(r0, chb) + ((p & 1) << 3));
        const e16x4 b1 = ds_read_tr(b + tn_off(r1, chb) + ((p & 1) << 3));
        af[i] = (e16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        bfr[i] = (e16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  if (do_cs) {  // 16 row phases x 128 columns of partial sums -> LDS -> one atomic per column
    float* red = (float*)lds;               // the k-loop ended with a barrier: tiles are dead
#pragma unroll
    for (int j = 0; j < 8; ++j) red[srow * 128 + sch * 8 + j] = cs[j];
    __syncthreads();
    if (t < 128 && m0 + t < g.M) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += red[r * 128 + t];
      atomicAdd(g.a_colsum + (g.glu_f ? glu_deint(m0 + t, g.glu_f) : m0 + t), s);
    }
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
          float* c = C + (int64_t)(g.glu_f ? glu_deint(m, g.glu_f) : m) * g.ldc + n;
          if (g.ksplit > 1) atomicAdd(c, acc[i][j][r]);
          else *c = acc[i][j][r] + (g.accumulate ? *c : 0.f);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------ TN, LDS-DMA ring
// wgrad with the operands streamed HBM -> LDS by LDS-DMA into a 3-stage ring (two 64-row k-steps in
// flight), 8 waves, 256 x 128 output tile (dy columns x input features), split-K over the token rows
// with fp32 atomics into the gradient buffer.  Stage image: A rows of 256 columns (512 B) and B rows of
// 128 columns (256 B), 16-byte chunks XOR-swizzled with tn_swz(row) on the source side; fragments by
// ds_read_b64_tr_b16 exactly as in k_gemm_tn.  Needs K % 64 == 0 (no zero fill with LDS-DMA).
template <int ABL>
__global__ __launch_bounds__(512, 2) void k_gemm_tn_ring(MfmaArgs g) {
  constexpr int S = 3, TBM = 256, TBN = 128, NW = 8, NIW = 6;
  constexpr int ABYTES = 64 * TBM * 2, STAGE = 64 * (TBM + TBN) * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int ntile = g.tiles_m * g.tiles_n;
  const int bid = xcd_remap(blockIdx.x, ntile * g.ksplit);
  // tile index fastest: the workgroups of one XCD share a k-chunk, so the dy / x rows they stream are
  // fetched from HBM once and served to the other tiles from that XCD's L2
  const int tile = bid % ntile, ks_id = bid / ntile;
  const int m0 = (tile / g.tiles_n) * TBM, n0 = (tile % g.tiles_n) * TBN;
  const int kbeg = ks_id * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  const int nk = (kend - kbeg) / 64;

  // this wave's pieces: ii = w + 8 j; ii < 32 -> A rows {2 ii, 2 ii + 1} (512 B each), else B rows 4 (ii-32) ..
  const e16* src[NIW];
  int64_t pitch[NIW];
#pragma unroll
  for (int j = 0; j < NIW; ++j) {
    const int ii = w + NW * j;
    if (ii < 32) {
      const int r = ii * 2 + (lane >> 5);
      const int c = (lane & 31) ^ tn_swz(r);
      src[j] = g.A + (int64_t)(kbeg + r) * g.lda + min(m0 + c * 8, g.M - 8);
      pitch[j] = (int64_t)64 * g.lda;
    } else {
      const int r = (ii - 32) * 4 + (lane >> 4);
      const int c = (lane & 15) ^ tn_swz(r);
      src[j] = g.B + (int64_t)(kbeg + r) * g.ldb + min(n0 + c * 8, g.N - 8);
      pitch[j] = (int64_t)64 * g.ldb;
    }
  }
  auto issue = [&](int kt) {
    unsigned char* st = lds + (kt % S) * STAGE;
    if (ABL == 2) return;
#pragma unroll
    for (int j = 0; j < NIW; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * pitch[j]),
                                       (__attribute__((address_space(3))) void*)(st + (w + NW * j) * 1024), 16, 0, 0);
  };
  auto offA = [](int row, int chunk) { return row * 512 + ((chunk ^ tn_swz(row)) << 4); };
  auto offB = [](int row, int chunk) { return row * 256 + ((chunk ^ tn_swz(row)) << 4); };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dy, from the A fragments.  Every wave of the tile row holds them (the fragment depends on wm
  // only), so the k-slices are dealt round-robin over the 2 x tiles_n waves that share them: a few instructions per wave instead of
  // 128 per k-step on one wave of every tile row, which made those workgroups the tail of the launch (-11 .. 19 %).
  const bool do_cs = g.a_colsum != nullptr;
  const int cs_slots = 2 * g.tiles_n;
  int cs_next = (tile % g.tiles_n) * 2 + wn;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) issue(s);
  // Transposed-fragment addresses, hoisted: a read is  stage + lane_off[i] + (ks*32 + 4*half) * pitch.
  // The swizzle of row r = ks*32 + grp*8 + q (+4) is tn_swz(r) = 2q | 8(grp&1): lane-constant, so only
  // the fragment index i needs its own per-lane offset (8 VGPRs) and the loop issues reads with
  // immediate offsets instead of recomputing the XOR address (2 VALU ops per read before).
  const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  int laneA[4], laneB[4];
  {
    const int lrow = grp * 8 + q, swz = tn_swz(lrow), sub = (p & 1) << 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cha = ((wm * 64 + i * 16) >> 3) + (p >> 1), chb = ((wn * 64 + i * 16) >> 3) + (p >> 1);
      laneA[i] = lrow * 512 + ((cha ^ swz) << 4) + sub;
      laneB[i] = ABYTES + lrow * 256 + ((chb ^ swz) << 4) + sub;
    }
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int later = min(S - 2, nk - 1 - kt);
    if (later >= S - 2) wait_vmcnt<NIW * (S - 2)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + S - 1 < nk) issue(kt + S - 1);
    const unsigned char* a = lds + (kt % S) * STAGE;
    if (ABL == 1) continue;
    // fragment reads in inline asm: the ds_read_tr builtin makes hipcc drain the whole LDS-DMA ring
    // (s_waitcnt vmcnt(0)) in front of every k-step; asm reads carry immediate offsets and are waited
    // for by hand (lgkmcnt(0) + sched_barrier, cdna_hip_programming.md 5.7 form iii)
    unsigned va[4], vb[4];
    const unsigned sbase = (unsigned)(uintptr_t)a;
#pragma unroll
    for (int i = 0; i < 4; ++i) { va[i] = sbase + (unsigned)laneA[i]; vb[i] = sbase + (unsigned)laneB[i]; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // two 32-deep k-slices
      s16x4 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (ks == 0) {
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a0[i]) : "v"(va[i]));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(a1[i]) : "v"(va[i]));
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0[i]) : "v"(vb[i]));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(b1[i]) : "v"(vb[i]));
        } else {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:16384" : "=v"(a0[i]) : "v"(va[i]));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:18432" : "=v"(a1[i]) : "v"(va[i]));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(b0[i]) : "v"(vb[i]));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9216" : "=v"(b1[i]) : "v"(vb[i]));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      e16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const e16x4 x0 = __builtin_bit_cast(e16x4, a0[i]), x1 = __builtin_bit_cast(e16x4, a1[i]);
        const e16x4 y0 = __builtin_bit_cast(e16x4, b0[i]), y1 = __builtin_bit_cast(e16x4, b1[i]);
        af[i] = (e16x8){x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        bfr[i] = (e16x8){y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
      }
      if (do_cs && kt * 2 + ks == cs_next) {   // this wave's turn (lane: column fr, 8 rows)
        cs_next += cs_slots;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) cs[i] += (float)af[i][j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
    }
  }
  const int fr = lane & 15, fq = lane >> 4;
  if (do_cs) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = cs[i];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const int mm = m0 + wm * 64 + i * 16 + fr;
      if (fq == 0 && mm < g.M) atomicAdd(g.a_colsum + (g.glu_f ? glu_deint(mm, g.glu_f) : mm), s);
    }
  }
  float* C = (float*)g.C;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = m0 + wm * 64 + i * 16 + fq * 4 + r;
        if (mm < g.M && n < g.N) {
          float* c = C + (int64_t)(g.glu_f ? glu_deint(mm, g.glu_f) : mm) * g.ldc + n;
          if (g.ksplit > 1) atomicAdd(c, acc[i][j][r]);
          else *c = acc[i][j][r] + (g.accumulate ? *c : 0.f);
        }
      }
    }
}

// Same kernel on 256 x 256 tiles: 8 waves of 128 x 64 (2 x 4), two 64-KiB ring slots.  Per FLOP a quarter
// less L2->LDS fill and a quarter fewer transposed LDS reads than the 256 x 128 form, and 64 MFMAs per
// wave between barriers; needs more split-K (fewer tiles), i.e. more fp32 atomics on the small dW.
#define TN_LIST_MAX 4096     // live-k-step list entries behind the 128-KiB ring (16 KiB)
// One (tile, k-chunk) unit of the wgrad form; shared by the single-problem kernel and the grouped one.
struct TnProb {
  const e16* A; const e16* B; float* C; float* a_colsum;
  const uint8_t* k_live;   // one byte per 64-token k-step, 0 = all of A's rows there are zero (padding): left out; null = all live
  int M, N, K, lda, ldb, ldc, tiles_n, ntile, ksplit, kchunk, glu_f, accumulate;
  int unit0;       // grouped launch: index of this problem's first (tile, k-chunk) unit
  int deal;        // afm_gemm_desc.reserved2 bit 2 (with k_live): k-steps dealt round-robin to the problem's units instead of contiguous chunks
};
__device__ __forceinline__ void tn256_unit(const TnProb& g, int tile, int ks_id, unsigned char* lds) {
  constexpr int S = 2, TBM = 256, TBN = 256, NW = 8, NIW = 8;
  constexpr int ABYTES = 64 * TBM * 2, STAGE = 64 * (TBM + TBN) * 2;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int m0 = (tile / g.tiles_n) * TBM, n0 = (tile % g.tiles_n) * TBN;
  // With the padded-row hint the k-steps are DEALT to the problem's ksplit units (unit c takes the steps c, c + ksplit, ...) instead of cut
  // into contiguous chunks: with the batch's live rows packed to the front of the token axis (afm_compact_plan mode 2) the chunks of the
  // late units held nothing but dead steps and half of a launch's units finished at once.  The list then holds ABSOLUTE steps.
  const bool deal = g.deal && g.k_live != nullptr && g.ksplit > 1 && (g.K / 64 + g.ksplit - 1) / g.ksplit <= TN_LIST_MAX;
  const int kbeg = deal ? 0 : ks_id * g.kchunk, kend = deal ? g.K : min(g.K, kbeg + g.kchunk);
  int nk = deal ? (g.K / 64 - ks_id + g.ksplit - 1) / g.ksplit : (kend - kbeg) / 64;
  // k-steps whose 64 token rows are all padding (A rows exact zeros: afm_gemm_desc.k_live) are left out: the list of the live ones
  // sits behind the ring (wave 0 compacts it with one ballot per 64 steps); without the hint it is the identity
  int* const kl = (int*)(lds + S * STAGE);
  const bool listed = g.k_live != nullptr && nk <= TN_LIST_MAX;
  if (listed) {
    if (w == 0) {
      int n = 0;
      for (int t0 = 0; t0 < nk; t0 += 64) {
        const int tt = t0 + lane;
        const int gs = deal ? ks_id + tt * g.ksplit : kbeg / 64 + tt;      // the step's index on the whole token axis
        const bool live = tt < nk && g.k_live[gs] != 0;
        const unsigned long long bal = __ballot(live);
        if (live) kl[1 + n + __popcll(bal & ((1ull << lane) - 1ull))] = deal ? gs : tt;
        n += __popcll(bal);
      }
      if (lane == 0) kl[0] = n;
    }
    __syncthreads();
    nk = __builtin_amdgcn_readfirstlane(kl[0]);
  }
  auto step_of = [&](int j) { return listed ? kl[1 + j] : j; };

  // this wave's pieces: ii = w + 8 j; ii < 32 -> A rows {2 ii, 2 ii + 1}, else B rows {2 (ii-32), +1} (512 B rows)
  const e16* src[NIW];
  int64_t pitch[NIW];
#pragma unroll
  for (int j = 0; j < NIW; ++j) {
    const int ii = w + NW * j;
    const int r = (ii & 31) * 2 + (lane >> 5);
    const int c = (lane & 31) ^ tn_swz(r);
    if (ii < 32) {
      src[j] = g.A + (int64_t)(kbeg + r) * g.lda + min(m0 + c * 8, g.M - 8);
      pitch[j] = (int64_t)64 * g.lda;
    } else {
      src[j] = g.B + (int64_t)(kbeg + r) * g.ldb + min(n0 + c * 8, g.N - 8);
      pitch[j] = (int64_t)64 * g.ldb;
    }
  }
  auto issue = [&](int kt) {
    unsigned char* st = lds + (kt % S) * STAGE;
    const int64_t step = step_of(kt);
#pragma unroll
    for (int j = 0; j < NIW; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + step * pitch[j]),
                                       (__attribute__((address_space(3))) void*)(st + (w + NW * j) * 1024), 16, 0, 0);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_cs = g.a_colsum != nullptr;     // k-slices dealt over the 4 x tiles_n waves holding the same A fragments (k_gemm_tn_ring)
  const int cs_slots = 4 * g.tiles_n;
  int cs_next = (tile % g.tiles_n) * 4 + wn;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  if (nk > 0) issue(0);
  const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  int laneA[8], laneB[4];
  {
    const int lrow = grp * 8 + q, swz = tn_swz(lrow), sub = (p & 1) << 3;
#pragma unroll
    for (int i = 0; i < 8; ++i) laneA[i] = lrow * 512 + (((((wm * 128 + i * 16) >> 3) + (p >> 1)) ^ swz) << 4) + sub;
#pragma unroll
    for (int j = 0; j < 4; ++j) laneB[j] = ABYTES + lrow * 512 + (((((wn * 64 + j * 16) >> 3) + (p >> 1)) ^ swz) << 4) + sub;
  }
  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt<0>();                       // two slots: only this step's pieces are in flight
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) issue(kt + 1);
    const unsigned sbase = (unsigned)(uintptr_t)(lds + (kt % S) * STAGE);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {       // two 32-deep k-slices; asm reads as in k_gemm_tn_ring
      s16x4 a0[8], a1[8], b0[4], b1[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned va = sbase + (unsigned)laneA[i];
        if (ks == 0) {
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a0[i]) : "v"(va));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(a1[i]) : "v"(va));
        } else {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:16384" : "=v"(a0[i]) : "v"(va));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:18432" : "=v"(a1[i]) : "v"(va));
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned vb = sbase + (unsigned)laneB[j];
        if (ks == 0) {
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0[j]) : "v"(vb));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(b1[j]) : "v"(vb));
        } else {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:16384" : "=v"(b0[j]) : "v"(vb));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:18432" : "=v"(b1[j]) : "v"(vb));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      e16x8 af[8], bfr[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const e16x4 x0 = __builtin_bit_cast(e16x4, a0[i]), x1 = __builtin_bit_cast(e16x4, a1[i]);
        af[i] = (e16x8){x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const e16x4 y0 = __builtin_bit_cast(e16x4, b0[j]), y1 = __builtin_bit_cast(e16x4, b1[j]);
        bfr[j] = (e16x8){y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
      }
      if (do_cs && kt * 2 + ks == cs_next) {
        cs_next += cs_slots;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) cs[i] += (float)af[i][j];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
    }
  }
  const int fr = lane & 15, fq = lane >> 4;
  if (do_cs) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float s = cs[i];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const int mm = m0 + wm * 128 + i * 16 + fr;
      if (fq == 0 && mm < g.M) atomicAdd(g.a_colsum + (g.glu_f ? glu_deint(mm, g.glu_f) : mm), s);
    }
  }
  float* C = g.C;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = m0 + wm * 128 + i * 16 + fq * 4 + r;
        if (mm < g.M && n < g.N) {
          float* c = C + (int64_t)(g.glu_f ? glu_deint(mm, g.glu_f) : mm) * g.ldc + n;
          if (g.ksplit > 1) atomicAdd(c, acc[i][j][r]);
          else *c = acc[i][j][r] + (g.accumulate ? *c : 0.f);
        }
      }
    }
}


__global__ __launch_bounds__(512) void k_gemm_tn_ring256(MfmaArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  TnProb pr;
  pr.A = g.A; pr.B = g.B; pr.C = (float*)g.C; pr.a_colsum = g.a_colsum; pr.k_live = g.k_live; pr.deal = g.deal;
  pr.M = g.M; pr.N = g.N; pr.K = g.K; pr.lda = g.lda; pr.ldb = g.ldb; pr.ldc = g.ldc;
  pr.tiles_n = g.tiles_n; pr.ntile = g.tiles_m * g.tiles_n; pr.ksplit = g.ksplit; pr.kchunk = g.kchunk;
  pr.glu_f = g.glu_f; pr.accumulate = g.accumulate; pr.unit0 = 0;
  const int bid = xcd_remap(blockIdx.x, pr.ntile * pr.ksplit);
  tn256_unit(pr, bid % pr.ntile, bid / pr.ntile, lds);     // tile index fastest: an XCD's workgroups share a k-chunk
}

// Grouped wgrad: the (tile, k-chunk) units of up to AFM_TN_GROUP_MAX weight gradients in ONE launch.  A layer's weight
// gradients are small matrices (4 .. 16 tiles of 256 x 256) over a long token axis: launched one by one each needs split-K 16 .. 64
// to fill 256 CUs, and every split adds the whole dW once more through memory-side fp32 atomics (1.3 TB/s chip-wide: 51 us of a
// 300-us launch at 131 072 tokens, 25 of 33 us at 16 384).  Together they fill the chip at split-K 4 .. 5.
#define AFM_TN_GROUP_MAX 8
#define TN_LIST_MAX_BYTES (4 * (TN_LIST_MAX + 1))
struct TnGroup { int n, units; TnProb p[AFM_TN_GROUP_MAX]; };
__global__ __launch_bounds__(512) void k_gemm_tn_group256(TnGroup gr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int bid = xcd_remap(blockIdx.x, gr.units);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < AFM_TN_GROUP_MAX; ++i)
    if (i < gr.n && bid >= gr.p[i].unit0) pi = i;
  const TnProb pr = gr.p[pi];
  const int local = bid - pr.unit0;
  tn256_unit(pr, local % pr.ntile, local / pr.ntile, lds);
}

#include "afm_gemm_tnw4_impl.h"

}  // namespace AFM_E16_NS
using namespace AFM_E16_NS;

// ------------------------------------------------------------------------------------------ dispatch
static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int AFM_E16_FN(afm_gemm_mfma_try)(const afm_gemm_desc* d, hipStream_t st) {
  if (d->a_dtype != AFM_E16 || d->b_dtype != AFM_E16) return AFM_ERR_UNSUPPORTED;
  MfmaArgs g;
  g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
  g.A = (const e16*)d->A; g.B = (const e16*)d->B; g.C = d->C;
  g.bias = d->bias; g.residual = d->residual; g.pre_act = d->pre_act; g.a_colsum = d->a_colsum; g.k_live = d->k_live;
  g.act = d->act; g.accumulate = d->accumulate;
  g.dd = afm_make_drop(&d->drop);
  g.tiles_m = (d->M + BM - 1) / BM; g.tiles_n = (d->N + BN - 1) / BN;
  g.ksplit = 1; g.kchunk = d->K; g.bias_in_lds = 0; g.live_off = 0; g.xgc = 0; g.dead_pre = 0; g.dead_nofill = 0; g.mperm = 0; g.deal = (d->reserved2 & 4) != 0;
  g.glu_f = d->glu_rows;
  g.stamps = nullptr;
#ifdef AFM_GEMM_ABLATIONS
  { const char* e = getenv("AFM_STAMPS"); if (e) g.stamps = (unsigned long long*)strtoull(e, nullptr, 0); }
#endif
  if (!aligned16(d->A) || !aligned16(d->B) || (d->lda & 7) || (d->ldb & 7)) return AFM_ERR_UNSUPPORTED;
  if (!d->transA && d->transB) {  // NT
    if ((d->K & 7) || d->K < 32 || d->N < 16) return AFM_ERR_UNSUPPORTED;
    // row hint (k_live): only where a zero row of A means a zero row of C, and C can be zero-filled in 16-byte pieces ...
    const bool fill16 = !((d->N & 7) || (d->ldc & 7) || !aligned16(d->C) || (d->M & 63));
    if (d->reserved2 & 2) {
      // ... or, in the FORWARD sense (reserved2 bit 1), wherever the caller does not care what the marked rows of C (and of a stored
      // pre_act) hold: they are written as zeros, whatever bias / activation / dropout the epilogue applies to the others
      if (d->residual || d->accumulate || !fill16 || (d->pre_act && (!aligned16(d->pre_act) || d->act == AFM_ACT_MUL_SAVED || d->act == AFM_ACT_GELU_BWD || d->act == AFM_ACT_GLU_BWD)))
        g.k_live = nullptr;
      else { g.dead_pre = d->pre_act != nullptr; g.dead_nofill = (d->reserved2 & 8) != 0; }
    }
    else if (d->bias || d->residual || d->accumulate || (d->act != AFM_ACT_NONE && d->act != AFM_ACT_MUL_SAVED && d->act != AFM_ACT_GLU_BWD) ||
        (d->pre_act && d->act == AFM_ACT_NONE) || d->drop.p > 0.f || !fill16)
      g.k_live = nullptr;
    else g.dead_nofill = (d->reserved2 & 8) != 0;      // (backward sense: the caller has checked that every consumer takes the hint too)
    if (d->bias && !aligned16(d->bias)) return AFM_ERR_UNSUPPORTED;
    if (!aligned16(d->C) || (d->residual && !aligned16(d->residual)) || (d->pre_act && !aligned16(d->pre_act)))
      return AFM_ERR_UNSUPPORTED;
    int variant = d->reserved;  // tile-shape experiments (tools/bench_gemm.py); 0 = pick by shape
    if (d->act >= AFM_ACT_GLU) {
      // fused gated FFN: whole 256 x 128 tiles through the loader-wave kernel, e16 in / out, contiguous C and pre_act (the
      // backward form reads pre_act with C's row stride, which may exceed the row: hi planes of pair tensors in mixed mode)
      const int ncol_c = d->act == AFM_ACT_GLU_BWD ? 2 * d->N : d->N / 2;
      if ((d->K & 63) || (d->M & 255) || (d->N & 127) || d->c_dtype != AFM_E16 || d->residual || d->accumulate ||
          (d->act == AFM_ACT_GLU_BWD ? (d->ldc < ncol_c || (d->ldc & 7)) : d->ldc != ncol_c) ||   // GLU_BWD: C and pre_act share ldc
          (d->drop.p > 0.f && (uint64_t)d->M * (uint64_t)d->N > 0x100000000ull) ||
          (d->act == AFM_ACT_GLU_BWD && (d->bias || d->drop.p > 0.f)))
        return AFM_ERR_UNSUPPORTED;
      int r;
      // 256 x 256 tiles (the round-3 persistent kernel) for the FORWARD forms where there are >= 1 024 of them, as for the other
      // fused epilogues: bit-identical, 1.545 -> 1.468 ms at the c4 shape (f 3072, d 768), 0.797 -> 0.745 at c5's; the backward
      // form is 2 .. 3 % slower there (1.033 vs 1.010) and stays on 256 x 128.  reserved = 24 / 28 force one form (A / B tests).
      const bool big = !(d->N & 255) && (int64_t)(d->M >> 8) * (d->N >> 8) >= 1024;
      const bool use28 = variant == 28 || (variant == 0 && big && d->act != AFM_ACT_GLU_BWD);
      if (use28 && big) {
        if (d->act == AFM_ACT_GLU) r = launch_nt_pring<true, 2, 4, 2, 0, 8, EPI_GLU, false>(g, st, 1);
        else if (d->act == AFM_ACT_GLU_SAVE) r = launch_nt_pring<true, 2, 4, 2, 0, 8, EPI_GLU_SG, false>(g, st, 1);
        else r = launch_nt_pring<true, 2, 4, 2, 0, 8, EPI_GLU_BWD, false>(g, st, 1);
      }
      else if (d->act == AFM_ACT_GLU) r = launch_nt_ws<true, 4, 0, EPI_GLU>(g, st);
      else if (d->act == AFM_ACT_GLU_SAVE) r = launch_nt_ws<true, 4, 0, EPI_GLU_SG>(g, st);
      else r = launch_nt_ws<true, 4, 0, EPI_GLU_BWD>(g, st);
      if (r != AFM_OK) return r;
      afm_set_last_algo("mfma_nt_glu");
      return AFM_OK;
    }
    if (variant == 0 && (d->K & 63) == 0) {
      // persistent LDS-DMA ring: 256x128 tiles when there are enough of them to fill the chip,
      // otherwise 128x128 at two workgroups per CU; anything with K % 64 != 0 keeps the
      // register-staged kernel (case 100).
      const int64_t big_tiles = (int64_t)((d->M + 255) / 256) * ((d->N + 127) / 128);
      variant = (d->N > 128 && big_tiles >= 256) ? 24 : 13;
      if (d->act >= AFM_ACT_GELU_SAVE_GRAD) variant = 24;   // only the loader-wave / 256x256 kernels know these epilogues
      // wide outputs of the long encoder sequence: 256x256 tiles (a quarter less L2->LDS fill and a quarter
      // fewer LDS fragment reads per FLOP, 64 MFMAs per wave between barriers): +7..12 % at N >= 1024
      const bool small_idx28 = (uint64_t)d->M * (uint64_t)d->N <= 0x100000000ull;
      if (variant == 24 && d->N >= 1024 && !(d->M & 255) && !(d->N & 255) && (int64_t)(d->M >> 8) * (d->N >> 8) >= 1024 &&
          d->c_dtype == AFM_E16 && !d->residual && !d->accumulate && !(d->ldc % 8) && d->act != AFM_ACT_RELU &&
          !(d->pre_act && d->act == AFM_ACT_NONE) && (d->drop.p <= 0.f || small_idx28))
        variant = 28;
    }
    // plain (bias-only) products over many whole 256 x 256 tiles: the ping-pong kernel (afm_gemm_pp_impl.h).  Measured against the
    // 256 x 128 loader-wave form at the c2 step's shapes, same box: N = 512 / K = 2048  +5 %, N = 512 / K = 1536  +9 %,
    // N = 1024 / K = 512  +6 %, N = 2048 / K = 512  +4 %, N = 1536 / K = 512 and N = 512 / K = 512  equal (left where they were);
    // 64-row-tile problems (the decoder's 16 384 rows) lose: too few tiles for a 256-CU chip.  Round 6 (tools/experiments/dec_rows_gemm.py, all tile
    // shapes at M = 16 384): the 256 x 128 loader-wave form wins every decoder-row shape (N 2304 / K 768: 61 us against the four-wave kernel's 69,
    // N 2048 / K 512: 37 against the ping-pong kernel's 41), so these kernels also want M >= 32 768 rows (the tile count alone let c4's decoder QKV in).
    if (d->reserved == 0 && (variant == 24 || variant == 28) && d->act == AFM_ACT_NONE && !d->pre_act && d->drop.p <= 0.f &&
        !d->residual && !d->accumulate && d->c_dtype == AFM_E16 && !(d->M & 255) && !(d->N & 255) && !(d->K & 63) && d->K >= 128 &&
        !(d->ldc % 8) && d->N <= PP_BIAS_MAX && (int64_t)(d->M >> 8) * (d->N >> 8) >= 512 && d->M >= 32768 && (d->K >= 768 || d->N >= 1024))
    {
      variant = (d->K & 127) ? 30 : 32;      // balanced phases where the K-tile count is even (+1 .. 5 % on most shapes, two runs)
      // K >= 768: four waves of 128 x 128 with the overlap inside the wave (afm_gemm_w4_impl.h; bit-identical to the ping-pong kernel).
      // One box, one process, order swapped between two rounds (tools/experiments/w4_gemm.py, fp16, M = 131 072): N 512 / K 2048 233-235 us
      // vs 240-244, N 512 / K 1536 183 vs 189, c4's N 768 / K 3072 486-488 vs 502-510, N 768 / K 2304 375 vs 387-391, N 768 / K 6144 937 vs
      // 969-988 (-3 .. 4 %); its K = 768 products 0 .. 2 % ahead (400 vs 402-411, 266 vs 268-275, 535 vs 541-548); K = 512 stays where it was
      // (196-221 vs 195-201 at N 1536: the epilogue, which all four waves reach together, weighs too much there).
      if (!(d->K & 127) && d->K >= 768 && d->N <= W4_BIAS_MAX) variant = 40;
    }
    if (d->act >= AFM_ACT_GELU_SAVE_GRAD && variant != 24 && variant != 28 && variant != 213 && variant != 214) return AFM_ERR_UNSUPPORTED;
    int r;
#define NT_CASE(WM, WN, NWM, NWN, BKT) \
    (d->c_dtype == AFM_E16 ? launch_nt<true, WM, WN, NWM, NWN, BKT>(g, st) : launch_nt<false, WM, WN, NWM, NWN, BKT>(g, st))
#define PRING_CASE(NWM, NWN, S, BPC) \
    (d->c_dtype == AFM_E16 ? launch_nt_pring<true, NWM, NWN, S>(g, st, BPC) : launch_nt_pring<false, NWM, NWN, S>(g, st, BPC))
    switch (variant) {
      case 213:   // (round-6 probe) the save-grad epilogues on 128 x 128 tiles, four waves, TWO workgroups per CU: one's epilogue under the other's main loop
      case 214: { // (214: 256 x 128 tiles, eight waves, 2 stages = 96 KB: one per CU, for comparison)
        if ((d->K & 63) || (d->M & 255) || (d->N & 127) || d->c_dtype != AFM_E16 || d->residual || d->accumulate || (d->ldc % 8) ||
            (d->act != AFM_ACT_GELU_SAVE_GRAD && d->act != AFM_ACT_MUL_SAVED) || (uint64_t)d->M * (uint64_t)d->N > 0x100000000ull) { r = AFM_ERR_UNSUPPORTED; break; }
        if (variant == 213) r = d->act == AFM_ACT_GELU_SAVE_GRAD ? launch_nt_pring<true, 2, 2, 2, 0, 4, EPI_GELU_SG, false>(g, st, 2)
                                                                 : launch_nt_pring<true, 2, 2, 2, 0, 4, EPI_MUL, false>(g, st, 2);
        else r = d->act == AFM_ACT_GELU_SAVE_GRAD ? launch_nt_pring<true, 4, 2, 2, 0, 4, EPI_GELU_SG, false>(g, st, 1)
                                                  : launch_nt_pring<true, 4, 2, 2, 0, 4, EPI_MUL, false>(g, st, 1);
        break;
      }
      case 12: r = (d->K & 63) ? AFM_ERR_UNSUPPORTED : PRING_CASE(4, 2, 3, 1); break;   // persistent 256x128, 8 waves, 3 stages
      case 13: r = (d->K & 63) ? AFM_ERR_UNSUPPORTED : PRING_CASE(2, 2, 2, 2); break;   // persistent 128x128, 4 waves, 2 per CU
      case 28: {   // persistent 256x256 (8 waves of 128x64, 2 stages), whole tiles only, e16 output
        const bool small_idx = (uint64_t)d->M * (uint64_t)d->N <= 0x100000000ull;
        if ((d->K & 63) || (d->M & 255) || (d->N & 255) || d->c_dtype != AFM_E16 || d->residual || d->accumulate ||
            (d->N % 8) || (d->ldc % 8) || d->act == AFM_ACT_RELU || (d->pre_act && d->act == AFM_ACT_NONE) ||
            (d->drop.p > 0.f && !small_idx)) { r = AFM_ERR_UNSUPPORTED; break; }
        if (d->act == AFM_ACT_GELU_SAVE_GRAD) r = launch_nt_pring<true, 2, 4, 2, 0, 8, EPI_GELU_SG, false>(g, st, 1);
        else if (d->act == AFM_ACT_MUL_SAVED) r = launch_nt_pring<true, 2, 4, 2, 0, 8, EPI_MUL, false>(g, st, 1);
        else if (d->act == AFM_ACT_GELU_BWD) r = launch_nt_pring<true, 2, 4, 2, 0, 8, EPI_GELU_BWD, false>(g, st, 1);
        else if (d->act == AFM_ACT_GELU) r = launch_nt_pring<true, 2, 4, 2, 0, 8, EPI_GELU, false>(g, st, 1);
        else if (d->drop.p > 0.f) r = launch_nt_pring<true, 2, 4, 2, 0, 8, EPI_DROP, false>(g, st, 1);
        else r = launch_nt_pring<true, 2, 4, 2, 0, 8, EPI_PLAIN, false>(g, st, 1);
        break;
      }
      case 30:   // ping-pong 256 x 256 (afm_gemm_pp_impl.h): whole tiles, e16 output, plain (+ bias) epilogue
        if ((d->K & 63) || d->K < 128 || (d->M & 255) || (d->N & 255) || d->N > PP_BIAS_MAX || d->c_dtype != AFM_E16 || d->residual ||
            d->accumulate || d->act != AFM_ACT_NONE || d->pre_act || d->drop.p > 0.f || (d->ldc % 8)) { r = AFM_ERR_UNSUPPORTED; break; }
        r = launch_nt_pp<EPI_PLAIN>(g, st);
        break;
      case 40:   // four waves of 128 x 128 with the overlap inside the wave (afm_gemm_w4_impl.h): whole tiles, K % 128 == 0, plain (+ bias)
        if ((d->K & 127) || d->K < 256 || (d->M & 255) || (d->N & 255) || d->N > W4_BIAS_MAX || d->c_dtype != AFM_E16 || d->residual ||
            d->accumulate || d->act != AFM_ACT_NONE || d->pre_act || d->drop.p > 0.f || (d->ldc % 8)) { r = AFM_ERR_UNSUPPORTED; break; }
        r = launch_nt_w4<0>(g, st);
        break;
#ifdef AFM_GEMM_ABLATIONS
      case 401: r = launch_nt_w4<1>(g, st); break;
      case 402: r = launch_nt_w4<2>(g, st); break;
      case 404: r = launch_nt_w4<4>(g, st); break;
      case 406: r = launch_nt_w4<6>(g, st); break;
#endif
      case 33:   // (A/B partner of 30: one barrier per phase)
      case 34:   // (one barrier per phase + balanced phases; K % 128 == 0)
        if ((d->K & (variant == 34 ? 127 : 63)) || d->K < 128 || (d->M & 255) || (d->N & 255) || d->N > PP_BIAS_MAX || d->c_dtype != AFM_E16 ||
            d->residual || d->accumulate || d->act != AFM_ACT_NONE || d->pre_act || d->drop.p > 0.f || (d->ldc % 8)) { r = AFM_ERR_UNSUPPORTED; break; }
        r = variant == 34 ? launch_nt_pp<EPI_PLAIN, 0, false, true, true>(g, st) : launch_nt_pp<EPI_PLAIN, 0, false, false, true>(g, st);
        break;
      case 32:   // (A/B partner of 30: balanced phases, b0 of the next K-tile read one phase early; K % 128 == 0)
        if ((d->K & 127) || d->K < 128 || (d->M & 255) || (d->N & 255) || d->N > PP_BIAS_MAX || d->c_dtype != AFM_E16 || d->residual ||
            d->accumulate || d->act != AFM_ACT_NONE || d->pre_act || d->drop.p > 0.f || (d->ldc % 8)) { r = AFM_ERR_UNSUPPORTED; break; }
        r = launch_nt_pp<EPI_PLAIN, 0, false, true>(g, st);
        break;
      case 31:   // (A/B partner of 30: the second DMA piece of a phase issued in the wave's MFMA section)
        if ((d->K & 63) || d->K < 128 || (d->M & 255) || (d->N & 255) || d->N > PP_BIAS_MAX || d->c_dtype != AFM_E16 || d->residual ||
            d->accumulate || d->act != AFM_ACT_NONE || d->pre_act || d->drop.p > 0.f || (d->ldc % 8)) { r = AFM_ERR_UNSUPPORTED; break; }
        r = launch_nt_pp<EPI_PLAIN, 0, true>(g, st);
        break;
#ifdef AFM_GEMM_ABLATIONS
      case 301: r = launch_nt_pp<EPI_PLAIN, 1>(g, st); break;
      case 302: r = launch_nt_pp<EPI_PLAIN, 2>(g, st); break;
      case 304: r = launch_nt_pp<EPI_PLAIN, 4>(g, st); break;
      case 306: r = launch_nt_pp<EPI_PLAIN, 6>(g, st); break;
      case 310: g.accumulate = 7; r = launch_nt_pp<EPI_PLAIN, 0>(g, st); break;   // tile stamps only (in-kernel clock)
      case 316: g.accumulate = 7; r = launch_nt_pp<EPI_PLAIN, 6>(g, st); break;
#endif
      case 22: r = (d->K & 63) ? AFM_ERR_UNSUPPORTED : (d->c_dtype == AFM_E16 ? launch_nt_ws<true, 2>(g, st) : launch_nt_ws<false, 2>(g, st)); break;
      case 24:   // persistent 256x128, 8 MFMA waves + 4 loader waves, epilogue picked at compile time
      case 25: { // (25: same tile walk with the generic run-time epilogue, for A/B timing)
        if (d->K & 63) { r = AFM_ERR_UNSUPPORTED; break; }
        if (d->c_dtype != AFM_E16) { r = d->act >= AFM_ACT_GELU_SAVE_GRAD ? AFM_ERR_UNSUPPORTED : launch_nt_ws<false, 4>(g, st); break; }
        int epi = EPI_GENERIC;
        const bool small_idx = (uint64_t)d->M * (uint64_t)d->N <= 0x100000000ull;
        const bool dropping = d->drop.p > 0.f;
        if (variant == 24 && !d->residual && !d->accumulate && (small_idx || !dropping)) {
          if (d->act == AFM_ACT_GELU_SAVE_GRAD) epi = EPI_GELU_SG;
          else if (d->act == AFM_ACT_MUL_SAVED) epi = EPI_MUL;
          else if (d->act == AFM_ACT_GELU_BWD) epi = EPI_GELU_BWD;
          else if (d->act == AFM_ACT_GELU) epi = EPI_GELU;
          else if (d->act == AFM_ACT_NONE && !d->pre_act) epi = dropping ? EPI_DROP : EPI_PLAIN;
        }
        // the save-grad pair exists only as whole-tile staged epilogues (the fragment epilogue of partial tiles
        // is kept small: growing it demotes the accumulators of every kernel that inlines it to scratch)
        if (d->act >= AFM_ACT_GELU_SAVE_GRAD && (epi == EPI_GENERIC || (d->M & 255) || (d->N & 127) || (d->ldc % 8))) {
          r = AFM_ERR_UNSUPPORTED; break;   // FMA kernel
        }
        switch (epi) {
          case EPI_PLAIN: r = launch_nt_ws<true, 4, 0, EPI_PLAIN>(g, st); break;
          case EPI_DROP: r = launch_nt_ws<true, 4, 0, EPI_DROP>(g, st); break;
          case EPI_GELU: r = launch_nt_ws<true, 4, 0, EPI_GELU>(g, st); break;
          case EPI_GELU_BWD: r = launch_nt_ws<true, 4, 0, EPI_GELU_BWD>(g, st); break;
          case EPI_GELU_SG: r = launch_nt_ws<true, 4, 0, EPI_GELU_SG>(g, st); break;
          case EPI_MUL: r = launch_nt_ws<true, 4, 0, EPI_MUL>(g, st); break;
          default: r = launch_nt_ws<true, 4>(g, st); break;
        }
        break;
      }
#ifdef AFM_GEMM_ABLATIONS
      case 241: r = launch_nt_ws<true, 4, 1>(g, st); break;
      case 242: r = launch_nt_ws<true, 4, 2>(g, st); break;
      case 243: r = launch_nt_ws<true, 4, 3>(g, st); break;
      case 244: r = launch_nt_ws<true, 4, 4>(g, st); break;
      case 246: r = launch_nt_ws<true, 4, 6>(g, st); break;
      case 247: r = launch_nt_ws<true, 4, 7>(g, st); break;
      case 248: r = launch_nt_ws<true, 4, 8, EPI_PLAIN, 0>(g, st); break;    // epilogue without HBM stores (every tile writes tile 0)
      case 249: r = launch_nt_ws<true, 4, 0, EPI_PLAIN, 0>(g, st); break;    // plain compile-time epilogue, default store policy
      case 250: r = launch_nt_ws<true, 4, 4, EPI_PLAIN, 0>(g, st); break;    // no epilogue
      case 252: r = launch_nt_ws<true, 4, 0, EPI_PLAIN, 16>(g, st); break;   // sc1 stores (written through, dropped from L2)
      case 256: r = launch_nt_ws<true, 4, 0, EPI_PLAIN, 2>(g, st); break;    // nt stores (the shipped policy)
      case 257: r = launch_nt_ws<true, 4, 0, EPI_PLAIN, 18>(g, st); break;   // sc1 nt
      case 121: r = launch_nt_pring<true, 4, 2, 3, 1>(g, st, 1); break;
      case 122: r = launch_nt_pring<true, 4, 2, 3, 2>(g, st, 1); break;
      case 124: r = launch_nt_pring<true, 4, 2, 3, 4>(g, st, 1); break;
      case 125: r = launch_nt_pring<true, 4, 2, 3, 5>(g, st, 1); break;
      case 126: r = launch_nt_pring<true, 4, 2, 3, 6>(g, st, 1); break;
      case 127: r = launch_nt_pring<true, 4, 2, 3, 7>(g, st, 1); break;
#endif
      default: r = NT_CASE(4, 4, 2, 2, 64); break;                                       // register-staged 128x128
    }
#undef PRING_CASE
#undef NT_CASE
    if (r != AFM_OK) return r;
    afm_set_last_algo(variant == 28 ? "mfma_nt_256" : (variant >= 30 && variant <= 34) ? "mfma_nt_pp" : (variant == 40 || variant / 100 == 4) ? "mfma_nt_w4" : "mfma_nt");     // (_256: the 256 x 256-tile form)
    return AFM_OK;
  }
  if (d->transA && !d->transB) {  // TN: the wgrad form only
    if (d->c_dtype != AFM_F32 || d->bias || d->residual || d->pre_act || d->act != AFM_ACT_NONE || d->drop.p > 0.f ||
        (d->a_colsum && ((uintptr_t)d->a_colsum & 3)))
      return AFM_ERR_UNSUPPORTED;
    if ((d->M & 7) || (d->N & 7) || d->K < 64 || d->M < 16 || d->N < 16) return AFM_ERR_UNSUPPORTED;
    // 256 x 256 tiles when the gradient matrix has at least 8 of them and the token count is long enough for
    // every workgroup to run >= 64 k-steps (measured at 131072 tokens: +4..18 % at 1536x512, 2048x512, 512x2048;
    // -3 % at 512x512, which keeps the 256 x 128 form, as do the decoder's 16 k-token shapes); 105 / 106 force a form
    const bool want256 = d->reserved == 105 || d->reserved == 107 || d->reserved == 108 ||
                         (d->reserved == 0 && d->K >= 65536 && (int64_t)((d->M + 255) / 256) * ((d->N + 255) / 256) >= 8);
    if (want256 && (d->K & 63) == 0 && d->K >= 4096 && d->M >= 256 && d->N >= 256) {
      // 256 x 256 tiles
      g.tiles_m = (d->M + 255) / 256; g.tiles_n = (d->N + 255) / 256;
      const int tiles = g.tiles_m * g.tiles_n;
      int ksplit = tiles >= 256 ? 1 : 256 / tiles;
      const int maxs = d->K / 1024;
      if (ksplit > maxs) ksplit = maxs;
      if (ksplit < 1) ksplit = 1;
      int kchunk = ((d->K / 64 + ksplit - 1) / ksplit) * 64;
      ksplit = (d->K + kchunk - 1) / kchunk;
      g.ksplit = ksplit; g.kchunk = kchunk;
      if (ksplit > 1 && !d->accumulate) {
        if (hipMemset2DAsync(d->C, sizeof(float) * d->ldc, 0, sizeof(float) * d->N, d->M, st) != hipSuccess)
          return AFM_ERR_LAUNCH;
      }
      static AfmOncePerDevice attr256;
      if (attr256.need()) {
        (void)hipFuncSetAttribute((const void*)k_gemm_tn_ring256, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 512 * 2 + TN_LIST_MAX_BYTES);
        (void)hipFuncSetAttribute((const void*)k_gemm_tn_w4, hipFuncAttributeMaxDynamicSharedMemorySize, TW4_RING + TN_LIST_MAX_BYTES);
      }
      // whole 256 x 256 tiles: the four-wave unit (afm_gemm_tnw4_impl.h); reserved = 107 keeps the eight-wave one (A / B tests)
      if (g.k_live && (d->K / 64 + ksplit - 1) / ksplit <= TN_LIST_MAX && kchunk / 64 <= TN_LIST_MAX) afm_note_hint(1);
      if (!(d->M & 255) && !(d->N & 255) && d->reserved == 108) {
        AFM_LAUNCH(k_gemm_tn_w4, dim3(tiles * ksplit), dim3(256), TW4_RING + TN_LIST_MAX_BYTES, st, g);
        afm_set_last_algo(ksplit > 1 ? "mfma_tn_w4_splitk" : "mfma_tn_w4");
        return AFM_OK;
      }
      AFM_LAUNCH(k_gemm_tn_ring256, dim3(tiles * ksplit), dim3(512), 2 * 64 * 512 * 2 + TN_LIST_MAX_BYTES, st, g);
      afm_set_last_algo(ksplit > 1 ? "mfma_tn_ring256_splitk" : "mfma_tn_ring256");
      return AFM_OK;
    }
    if (d->reserved != 100 && (d->K & 63) == 0 && d->K >= 4096 && d->M >= 64 && d->N >= 64) {
      // LDS-DMA ring kernel: 256 x 128 tiles, one 8-wave workgroup per CU, split-K to fill the chip
      g.tiles_m = (d->M + 255) / 256; g.tiles_n = (d->N + 127) / 128;
      const int tiles = g.tiles_m * g.tiles_n;
      int ksplit = tiles >= 256 ? 1 : 256 / tiles;   // at most one workgroup per CU, no ragged second wave
      const int maxs = d->K / 1024;            // >= 16 k-steps per workgroup
      if (ksplit > maxs) ksplit = maxs;
      if (ksplit < 1) ksplit = 1;
      int kchunk = ((d->K / 64 + ksplit - 1) / ksplit) * 64;
      ksplit = (d->K + kchunk - 1) / kchunk;
      g.ksplit = ksplit; g.kchunk = kchunk;
      if (ksplit > 1 && !d->accumulate) {
        if (hipMemset2DAsync(d->C, sizeof(float) * d->ldc, 0, sizeof(float) * d->N, d->M, st) != hipSuccess)
          return AFM_ERR_LAUNCH;
      }
      static AfmOncePerDevice attr_done;
      if (attr_done.need()) {
        (void)hipFuncSetAttribute((const void*)k_gemm_tn_ring<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 384 * 2);
        (void)hipFuncSetAttribute((const void*)k_gemm_tn_ring<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 384 * 2);
        (void)hipFuncSetAttribute((const void*)k_gemm_tn_ring<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 384 * 2);
      }
      if (d->reserved == 101) AFM_LAUNCH(k_gemm_tn_ring<1>, dim3(tiles * ksplit), dim3(512), 3 * 64 * 384 * 2, st, g);
      else if (d->reserved == 102) AFM_LAUNCH(k_gemm_tn_ring<2>, dim3(tiles * ksplit), dim3(512), 3 * 64 * 384 * 2, st, g);
      else AFM_LAUNCH(k_gemm_tn_ring<0>, dim3(tiles * ksplit), dim3(512), 3 * 64 * 384 * 2, st, g);
      afm_set_last_algo(ksplit > 1 ? "mfma_tn_ring_splitk" : "mfma_tn_ring");
      return AFM_OK;
    }
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

// ------------------------------------------------------------------------------------------ grouped wgrad
// A problem the grouped wgrad launch takes: the TN form with 16-bit operands of this translation unit's type, fp32 accumulate
// INTO C (the gradient buffer), whole 64-token k-steps, at least one 256 x 256 tile's worth of rows and columns.
bool AFM_E16_FN(afm_gemm_tn_group_eligible)(const afm_gemm_desc* d) {
  if (!d->transA || d->transB || d->a_dtype != AFM_E16 || d->b_dtype != AFM_E16 || d->c_dtype != AFM_F32) return false;
  if (!d->accumulate || d->bias || d->residual || d->pre_act || d->act != AFM_ACT_NONE || d->drop.p > 0.f) return false;
  if (d->a_colsum && ((uintptr_t)d->a_colsum & 3)) return false;
  if ((d->M & 7) || (d->N & 7) || d->M < 256 || d->N < 256 || (d->K & 63) || d->K < 1024) return false;
  if (d->lda < d->M || d->ldb < d->N || d->ldc < d->N) return false;     // (afm_gemm reports it)
  if (!aligned16(d->A) || !aligned16(d->B) || (d->lda & 7) || (d->ldb & 7)) return false;
  if (d->algo == AFM_ALGO_GENERIC || (d->reserved != 0 && d->reserved != 105 && d->reserved != 107 && d->reserved != 108)) return false;
  return true;
}
// count <= AFM_TN_GROUP_MAX eligible problems.  One chunk length (in 64-token k-steps) for all of them: the smallest for which the
// units fit one wave of workgroups (256), so every problem is split only as far as filling the chip TOGETHER needs.
int AFM_E16_FN(afm_gemm_tn_group_launch)(const afm_gemm_desc* const* ds, int count, hipStream_t st) {
  if (count < 1 || count > AFM_TN_GROUP_MAX) return AFM_ERR_ARG;
  TnGroup gr;
  gr.n = count;
  int steps[AFM_TN_GROUP_MAX], tiles = 0;
  int64_t work = 0;
  int max_steps = 0;
  for (int i = 0; i < count; ++i) {
    const afm_gemm_desc* d = ds[i];
    TnProb& pr = gr.p[i];
    pr.A = (const e16*)d->A; pr.B = (const e16*)d->B; pr.C = (float*)d->C; pr.a_colsum = d->a_colsum; pr.k_live = d->k_live; pr.deal = (d->reserved2 & 4) != 0;
    pr.M = d->M; pr.N = d->N; pr.K = d->K; pr.lda = d->lda; pr.ldb = d->ldb; pr.ldc = d->ldc;
    pr.tiles_n = (d->N + 255) / 256; pr.ntile = ((d->M + 255) / 256) * pr.tiles_n;
    pr.glu_f = d->glu_rows; pr.accumulate = 1;
    steps[i] = d->K / 64;
    tiles += pr.ntile;
    work += (int64_t)pr.ntile * steps[i];
    if (steps[i] > max_steps) max_steps = steps[i];
  }
  // The chunk length (in 64-token steps) that minimises the estimated launch time.  Rounds 3-4 took the smallest chunk whose units fit
  // ONE wave of 256 workgroups; that leaves the chip half empty where the tile count sits between 128 and 256 -- c4's encoder layer:
  // 144 tiles, split-K 2 would be 288 units, so every tile ran unsplit on 144 of 256 CUs (784 TF/s where c2's 48-tile layer reaches
  // 1 000).  Units are dispatched in index order as workgroup slots free up, so several near-full rounds of short units beat one
  // half-empty round of long ones: cost(chunk) = max(sum of unit lengths / 256, longest unit), every unit charged OVH steps for its
  // ring fill and its 256 KB of fp32 atomics (~40 steps: 50 us per round of 256 units at the 1.3 TB/s atomic rate + the prologue).
  // c4 encoder layer: chunk 293 (7 windows, 1 008 units, ~4 rounds): estimated 1 312 steps against 2 083; c2's (48 tiles) keeps split 5.
  // The same estimate serves groups of 256 tiles and more (larger models; several layers in one launch): unsplit, 288 tiles are one full
  // round and one of 32 units -- the probe that found it lost 8 % of the c4 step (tools/experiments/r5_wgrad_layers.sh).
  int chunk = max_steps;                                   // (AFM_TN_ONE_WAVE with >= 256 tiles: no split-K at all)
  static const int ovh = getenv("AFM_TN_OVH") ? atoi(getenv("AFM_TN_OVH")) : 40;
  static const bool one_wave = getenv("AFM_TN_ONE_WAVE") != nullptr;      // (read once: this is the launch path of every layer's backward)
  if (!one_wave) {
    // (the plan depends on the shapes only: the last 64 are remembered, replaced round-robin -- c2 + c3 + c4 + c5 in one bench process
    // are ~20 distinct layer sets)
    struct Plan { uint64_t key; int chunk; };
    static thread_local Plan cache[64];
    static thread_local int cache_n = 0, cache_next = 0;
    uint64_t key = 1469598103934665603ull ^ (uint64_t)count;
    for (int i = 0; i < count; ++i) key = (key * 1099511628211ull) ^ ((uint64_t)gr.p[i].ntile << 32 | (uint32_t)steps[i]);
    bool hit = false;
    for (int i = 0; i < cache_n && !hit; ++i)
      if (cache[i].key == key) { chunk = cache[i].chunk; hit = true; }
    if (!hit) {
      double best = 1e30;
      for (int ks0 = 1; ks0 <= 64; ++ks0) {                // candidate: the longest problem split ks0 ways
        const int c = (max_steps + ks0 - 1) / ks0;
        if (c < 16) break;                                 // >= 1024 tokens per unit
        if ((int64_t)tiles * ks0 > 8192) break;            // (host time of the estimate: units x 256 slots per candidate)
        // greedy schedule of the units in launch order on 256 slots: a min-heap of the slots' finish times (O(units log 256) per candidate)
        std::vector<int> finish(256, 0);      // (all zeros is a heap)
        int makespan = 0;
        for (int i = 0; i < count; ++i) {
          int ks = (steps[i] + c - 1) / c;
          const int per = (steps[i] + ks - 1) / ks;
          ks = (steps[i] + per - 1) / per;
          const int n = gr.p[i].ntile * ks;
          for (int u = 0; u < n; ++u) {
            std::pop_heap(finish.begin(), finish.end(), std::greater<int>());      // the earliest slot to the back
            finish.back() += per + ovh;
            if (finish.back() > makespan) makespan = finish.back();
            std::push_heap(finish.begin(), finish.end(), std::greater<int>());
          }
        }
        if (makespan < best * 0.97) { best = makespan; chunk = c; }   // (a finer split has to pay by 3 %)
      }
      cache[cache_n < 64 ? cache_n++ : (cache_next++ & 63)] = Plan{key, chunk};
    }
  } else if (tiles < 256) {                                // AFM_TN_ONE_WAVE: the rule of rounds 3-4 (A / B runs)
    chunk = (int)((work + 255) / 256);
    if (chunk < 16) chunk = 16;
    for (;; ++chunk) {
      int units = 0;
      for (int i = 0; i < count; ++i) units += gr.p[i].ntile * ((steps[i] + chunk - 1) / chunk);
      if (units <= 256 || chunk >= max_steps) break;
    }
  }
  int units = 0;
  for (int i = 0; i < count; ++i) {
    TnProb& pr = gr.p[i];
    int ks = (steps[i] + chunk - 1) / chunk;
    const int per = (steps[i] + ks - 1) / ks;              // equal chunks inside a problem
    ks = (steps[i] + per - 1) / per;
    pr.ksplit = ks; pr.kchunk = per * 64; pr.unit0 = units;
    units += pr.ntile * ks;
  }
  gr.units = units;
  {   // hint bookkeeping: every hinted problem's units keep their live-step lists (<= TN_LIST_MAX steps per unit)
    int st_ = 0;
    for (int i = 0; i < count; ++i)
      if (gr.p[i].k_live) {
        const bool ok = gr.p[i].kchunk / 64 <= TN_LIST_MAX && (steps[i] + gr.p[i].ksplit - 1) / gr.p[i].ksplit <= TN_LIST_MAX;
        st_ = ok ? (st_ < 0 ? -1 : 1) : -1;
      }
    afm_note_hint(st_);
  }
  for (int i = count; i < AFM_TN_GROUP_MAX; ++i) { gr.p[i] = gr.p[0]; gr.p[i].unit0 = 0x7fffffff; }
  static AfmOncePerDevice attr;
  if (attr.need()) {
    (void)hipFuncSetAttribute((const void*)k_gemm_tn_group256, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 512 * 2 + TN_LIST_MAX_BYTES);
    (void)hipFuncSetAttribute((const void*)k_gemm_tn_groupw4, hipFuncAttributeMaxDynamicSharedMemorySize, TW4_RING + TN_LIST_MAX_BYTES);
  }
  // every problem made of whole 256 x 256 tiles (all of a layer stack's weight gradients): the four-wave unit; a descriptor with
  // reserved = 105 / 107 keeps the eight-wave one (A / B tests)
  // The four-wave unit (afm_gemm_tnw4_impl.h) is NOT the default: measured (tools/experiments/tnw4_gemm.py + r5_tn_ab.sh, one process,
  // order swapped, 131 072 tokens, fp16) it equals the eight-wave unit within the run-to-run spread on every layer set (c2 encoder layer
  // 0.81-0.85 ms vs 0.81-0.83, c4's 2.35-2.37 vs 2.35, decoder sets 0.277-0.33 vs 0.277-0.29), and inside bench.py's step the grouped
  // launch came out SLOWER (1.18 vs 0.81 ms at the c2 encoder layer, c2 2 870 vs 3 130 samples/s): a transposed fragment costs two
  // LDS instructions, so the lone wave of a SIMD issues 126 instructions beside its 64 MFMAs per slice and has no slack left for the
  // eight LDS-DMA pieces.  reserved = 108 on every descriptor selects it (tests, A / B runs).
  bool w4 = true;
  for (int i = 0; i < count; ++i) w4 = w4 && !(ds[i]->M & 255) && !(ds[i]->N & 255) && ds[i]->reserved == 108;
  if (w4) {
    AFM_LAUNCH(k_gemm_tn_groupw4, dim3(units), dim3(256), TW4_RING + TN_LIST_MAX_BYTES, st, gr);
    afm_set_last_algo("mfma_tn_groupw4");
    return AFM_OK;
  }
  AFM_LAUNCH(k_gemm_tn_group256, dim3(units), dim3(512), 2 * 64 * 512 * 2 + TN_LIST_MAX_BYTES, st, gr);
  afm_set_last_algo("mfma_tn_group256");
  return AFM_OK;
}
