// "bf16x3" GEMMs for gfx950: operands are split bf16 pairs (AFM_BF16X2: value = hi + lo, include/afm_hip.h), every
// product runs as three bf16 MFMAs  hi*hi + hi*lo + lo*hi  (v_mfma_f32_16x16x32_bf16, fp32 accumulate): results at
// fp32-GEMM accuracy (the dropped lo*lo term is 2^-18 relative) on the bf16 matrix cores.
//
//   NT  C[m][n] = sum_k A[m][k] B[n][k]     forward / dgrad: persistent 256 x 128 tiles, 8 MFMA waves + 4 loader waves
//   TN  C[m][n] += sum_k A[k][m] B[k][n]    wgrad: 256 x 128 tiles, split-K over the token rows, fp32 atomics
//
// Both reuse the LDS images of the single-pass kernels (afm_gemm_mfma.hip) with the k-step halved to 32: one LDS row
// of the NT image holds [hi k0..31 | lo k0..31] (128 bytes, the XOR swizzle of the 64-deep single-pass row), the TN
// stage holds 32 hi rows then 32 lo rows.  The fragment reads are therefore the single-pass reads, k-slice 0 returning
// the hi fragments and k-slice 1 the lo fragments; the LDS-DMA pieces differ only in their per-lane SOURCE address.
// Per byte staged the kernel issues 1.5x the MFMAs of the single-pass form, so it sits closer to the MFMA roof.
#include "afm_common.h"

struct X3Args {
  int M, N, K;
  int lda, ldb, ldc;
  const bf16* A;
  const bf16* B;
  void* C;
  const float* bias;
  const void* residual;
  void* pre_act;
  float* a_colsum;
  int act, accumulate;
  int tiles_m, tiles_n;
  int ksplit, kchunk;   // TN only
  int bias_in_lds;
  int glu_f;            // gated-FFN interleave (include/afm_hip.h)
  int sg_hi_only;       // stored gradient factors (GELU_SG / GLU_SG): hi plane only (the bf16 backward of mixed mode reads nothing else)
  int abl;              // timing experiments (tools/bench_gemm_x3.py): 1 skip LDS reads + MFMAs, 2 skip LDS-DMA, 4 skip epilogue,
                        // 8 skip the LDS-DMA of the B rows, 16 read the B fragments once per tile
  DropDev dd;
};

template <int N> __device__ __forceinline__ void x3_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef __attribute__((ext_vector_type(4))) short x3_s16x4;

enum { X3_F32 = 0, X3_X2 = 2 };
enum { XE_GENERIC = 0, XE_PLAIN = 1, XE_GELU = 3, XE_GELU_BWD = 4, XE_GELU_SG = 5, XE_MUL = 6, XE_GLU = 7, XE_GLU_SG = 8, XE_GLU_BWD = 9 };
__device__ __forceinline__ int x3_glu_deint(int n, int f) { return ((n >> 3) << 2) + (n & 3) + ((n >> 2) & 1) * f; }

// keep * scale of N consecutive elements from idx0 (a multiple of N): the element-wise stream of afm_common.h (one block hash, N / 2 pair
// mixes); ones where the site does not drop
template <int N>
__device__ __forceinline__ void x3_keep_scale(const DropDev& d, bool drop_on, uint32_t idx0, float (&out)[N]) {
  if (drop_on) afm_keep_scale<N>(d, (uint64_t)idx0, out);
  else {
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = 1.0f;
  }
}

// ------------------------------------------------------------------------------------------ fragment epilogue
// Lane holds C[m][n0 .. n0+3] (edge tiles / shapes without whole tiles): every run-time option of afm_gemm.
template <int CT>
__device__ __forceinline__ float x3_ldc(const void* p, int64_t ci, int lo) {
  if (CT == X3_F32) return ((const float*)p)[ci];
  return (float)((const bf16*)p)[ci] + (float)((const bf16*)p)[ci + lo];
}
template <int CT>
__device__ __forceinline__ void x3_stc(void* p, int64_t ci, int lo, float v) {
  if (CT == X3_F32) { ((float*)p)[ci] = v; return; }
  bf16 h, l;
  afm_split(v, h, l);
  ((bf16*)p)[ci] = h; ((bf16*)p)[ci + lo] = l;
}
template <int CT>
__device__ __forceinline__ void x3_epilogue4(const X3Args& g, int m, int n0, f32x4 v) {
  if (m >= g.M || n0 >= g.N) return;
  const int lo = g.ldc >> 1;
  const int nv = min(4, g.N - n0);
  for (int r = 0; r < nv; ++r) {
    float x = v[r];
    const int n = n0 + r;
    const int64_t ci = (int64_t)m * g.ldc + n;
    const uint64_t di = (uint64_t)m * (uint64_t)g.N + (uint64_t)n;
    if (g.bias) x += g.bias[n];
    if (g.act == AFM_ACT_GELU_BWD) {
      x = afm_drop(g.dd, di, x) * afm_gelu_grad(x3_ldc<CT>(g.pre_act, ci, lo));
    } else if (g.act == AFM_ACT_GELU_SAVE_GRAD) {
      float y, yp;
      afm_gelu_both(x, y, yp);
      const float k = afm_drop(g.dd, di, 1.0f);
      x3_stc<CT>(g.pre_act, ci, lo, yp * k);
      x = y * k;
    } else if (g.act == AFM_ACT_MUL_SAVED) {
      x *= x3_ldc<CT>(g.pre_act, ci, lo);
    } else {
      if (g.pre_act) x3_stc<CT>(g.pre_act, ci, lo, x);
      if (g.act == AFM_ACT_RELU) x = fmaxf(x, 0.f);
      else if (g.act == AFM_ACT_GELU) x = afm_gelu(x);
      x = afm_drop(g.dd, di, x);
    }
    if (g.residual) x += x3_ldc<CT>(g.residual, ci, lo);
    if (g.accumulate) x += x3_ldc<CT>(g.C, ci, lo);
    x3_stc<CT>(g.C, ci, lo, x);
  }
}

// ------------------------------------------------------------------------------------------ staged epilogue
// Whole tiles: a wave's 64 x 64 accumulator block goes through a wave-private LDS patch 16 rows at a time so every
// global access is 16 bytes per lane on whole 128-byte lines (as afm_gemm_mfma.hip's staged epilogue); the split-pair
// output writes the hi and the lo plane of 8 rows x 64 columns with one store instruction each.
#define X3_STG_LD 68
template <int CT, int EPI, int WM>
__device__ __forceinline__ void x3_epilogue_staged(const X3Args& g, float* stg, const float* bias_lds,
                                                   f32x4 (&acc)[4][WM], int mw, int nw, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  if constexpr (CT == X3_X2) {
    const int c8 = (lane & 7) * 8, r8 = lane >> 3;
    const int n = nw + c8, lo = g.ldc >> 1;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == XE_GLU || EPI == XE_GLU_SG) {   // u / v biases of hidden units (n >> 1) .. +3, reference order [b1 ; bg]
      if (g.bias) { b0 = *(const f32x4*)(g.bias + (n >> 1)); b1 = *(const f32x4*)(g.bias + g.glu_f + (n >> 1)); }
    } else if (EPI != XE_GELU_BWD && EPI != XE_MUL && EPI != XE_GLU_BWD && g.bias) { b0 = *(const f32x4*)(bias_lds + n); b1 = *(const f32x4*)(bias_lds + n + 4); }
    const bool drop_on = g.dd.thresh != 0;
    bf16* const cbase = (bf16*)g.C + (int64_t)(mw + r8) * g.ldc + n;
    bf16* const pbase = (bf16*)g.pre_act + (int64_t)(mw + r8) * g.ldc + n;
    const uint32_t dbase = (uint32_t)(mw + r8) * (uint32_t)g.N + (uint32_t)n;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)(stg + fr * X3_STG_LD + j * 16 + fq * 4) = acc[j][i];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int q = i * 2 + hf, row = hf * 8 + r8;
        const f32x4 v0 = *(const f32x4*)(stg + row * X3_STG_LD + c8) + b0;
        const f32x4 v1 = *(const f32x4*)(stg + row * X3_STG_LD + c8 + 4) + b1;
        float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const int64_t ro = (int64_t)(q * 8) * g.ldc;
        const uint32_t di = dbase + (uint32_t)(q * 8) * (uint32_t)g.N;
        if constexpr (EPI == XE_GLU || EPI == XE_GLU_SG) {
          // x[0..3] = u, x[4..7] = v of hidden units (n >> 1) .. +3; C / the dropout stream are f = N/2 wide (ldc = 2 f: two planes)
          const int64_t rowi = mw + r8 + q * 8;
          const int hcol = n >> 1;
          const uint32_t dg0 = (uint32_t)rowi * (uint32_t)(g.N >> 1) + (uint32_t)hcol;
          float gv[4], sv[8], kp[4];
          x3_keep_scale<4>(g.dd, drop_on, dg0, kp);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float y, yp;
            afm_gelu_both(x[k], y, yp);
            const float keep = kp[k];
            gv[k] = y * x[4 + k] * keep; sv[k] = yp * x[4 + k] * keep; sv[4 + k] = y * keep;
          }
          bf16 h0, l0, h1, l1, h2, l2, h3, l3;
          afm_split(gv[0], h0, l0); afm_split(gv[1], h1, l1); afm_split(gv[2], h2, l2); afm_split(gv[3], h3, l3);
          bf16* cp = (bf16*)g.C + rowi * g.ldc + hcol;
          *(bf16x4*)cp = (bf16x4){h0, h1, h2, h3};
          *(bf16x4*)(cp + lo) = (bf16x4){l0, l1, l2, l3};
          if (EPI == XE_GLU_SG) {
            bf16* pp = (bf16*)g.pre_act + rowi * (2 * g.N) + n;        // saved tensor: M x N pairs, row stride 2 N
            if (g.sg_hi_only) {
              *(bf16x8*)pp = (bf16x8){(bf16)sv[0], (bf16)sv[1], (bf16)sv[2], (bf16)sv[3], (bf16)sv[4], (bf16)sv[5], (bf16)sv[6], (bf16)sv[7]};
            } else {
              bf16x8 sh, sl;
              afm_split8(sv, sh, sl);
              *(bf16x8*)pp = sh; *(bf16x8*)(pp + g.N) = sl;
            }
          }
          continue;
        }
        if constexpr (EPI == XE_GLU_BWD) {
          // x[0..7] = dg of hidden units n .. n+7; saved / output columns 2n .. 2n+15 of the 2f-wide pair tensors (ldc = 4 f)
          const int64_t rowi = mw + r8 + q * 8;
          const bf16* sp = (const bf16*)g.pre_act + rowi * g.ldc + 2 * n;
          bf16* cp = (bf16*)g.C + rowi * g.ldc + 2 * n;
          float o[16];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const bf16x8 sh = *(const bf16x8*)(sp + 8 * hh), sl = *(const bf16x8*)(sp + 8 * hh + lo);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              o[8 * hh + k] = x[4 * hh + k] * ((float)sh[k] + (float)sl[k]);
              o[8 * hh + 4 + k] = x[4 * hh + k] * ((float)sh[4 + k] + (float)sl[4 + k]);
            }
          }
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const float oo[8] = {o[8 * hh], o[8 * hh + 1], o[8 * hh + 2], o[8 * hh + 3], o[8 * hh + 4], o[8 * hh + 5], o[8 * hh + 6], o[8 * hh + 7]};
            bf16x8 oh, ol;
            afm_split8(oo, oh, ol);
            *(bf16x8*)(cp + 8 * hh) = oh; *(bf16x8*)(cp + 8 * hh + lo) = ol;
          }
          continue;
        }
        if (EPI == XE_GELU) {
          if (g.pre_act) {
            bf16x8 h, l;
            afm_split8(x, h, l);
            *(bf16x8*)(pbase + ro) = h; *(bf16x8*)(pbase + ro + lo) = l;
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = afm_gelu(x[k]);
          if (drop_on) {
            float kp[8];
            x3_keep_scale<8>(g.dd, true, di, kp);
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] *= kp[k];
          }
        }
        if (EPI == XE_GELU_SG) {
          float gp[8], kp[8];
          x3_keep_scale<8>(g.dd, drop_on, di, kp);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float y, yp;
            afm_gelu_both(x[k], y, yp);
            x[k] = y * kp[k]; gp[k] = yp * kp[k];
          }
          if (g.sg_hi_only) {
            *(bf16x8*)(pbase + ro) = (bf16x8){(bf16)gp[0], (bf16)gp[1], (bf16)gp[2], (bf16)gp[3], (bf16)gp[4], (bf16)gp[5], (bf16)gp[6], (bf16)gp[7]};
          } else {
            bf16x8 h, l;
            afm_split8(gp, h, l);
            *(bf16x8*)(pbase + ro) = h; *(bf16x8*)(pbase + ro + lo) = l;
          }
        }
        if (EPI == XE_MUL || EPI == XE_GELU_BWD) {
          const bf16x8 uh = *(const bf16x8*)(pbase + ro), ul = *(const bf16x8*)(pbase + ro + lo);
          float kp[8];
          x3_keep_scale<8>(g.dd, drop_on && EPI == XE_GELU_BWD, di, kp);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float u = (float)uh[k] + (float)ul[k];
            if (EPI == XE_MUL) x[k] *= u;
            else x[k] = x[k] * kp[k] * afm_gelu_grad(u);
          }
        }
        bf16x8 h, l;
        afm_split8(x, h, l);
        *(bf16x8*)(cbase + ro) = h; *(bf16x8*)(cbase + ro + lo) = l;
      }
    }
  } else {   // fp32 output: bias, residual, accumulate (logits, residual branches, encoder-output gradient)
    const int c4 = (lane & 15) * 4, r4 = lane >> 4;
    const int n = nw + c4;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) b0 = *(const f32x4*)(bias_lds + n);
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)(stg + fr * X3_STG_LD + j * 16 + fq * 4) = acc[j][i];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int row = qq * 4 + r4;
        const int64_t ci = (int64_t)(mw + i * 16 + row) * g.ldc + n;
        f32x4 v = *(const f32x4*)(stg + row * X3_STG_LD + c4) + b0;
        if (g.residual) v += *(const f32x4*)((const float*)g.residual + ci);
        if (g.accumulate) v += *(const f32x4*)((const float*)g.C + ci);
        *(f32x4*)((float*)g.C + ci) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ NT
template <int CT, int EPI>
__global__ __launch_bounds__(768) void k_x3_nt(X3Args g) {
  constexpr int NWN = 2, NW = 8, NL = 4, WM = 4, S = 3;
  constexpr int TBM = 256, TBN = 128;
  constexpr int NI = (TBM + TBN) / 8, NIL = NI / NL;
  constexpr int STAGE = (TBM + TBN) * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* bias_lds = (float*)(lds + S * STAGE);
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntiles = g.tiles_m * g.tiles_n;
  constexpr bool GLU_EPI = EPI == XE_GLU || EPI == XE_GLU_SG || EPI == XE_GLU_BWD;   // these read the bias from global memory
  if (g.bias_in_lds && !GLU_EPI) {
    for (int n = t; n < g.N; n += 64 * (NW + NL)) bias_lds[n] = g.bias ? g.bias[n] : 0.f;
    __syncthreads();
  }
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const int tpx = (ntiles + 7) >> 3;
  const int tlo = xcd * tpx, thi = min(ntiles, tlo + tpx);
  auto tile_of = [&](int it) { const int tt = tlo + it * nbx + bx; return tt < thi ? tt : -1; };
  auto tile_full = [&](int tile) {
    const int m0 = (tile / g.tiles_n) * TBM, n0 = (tile % g.tiles_n) * TBN;
    return g.bias_in_lds && m0 + TBM <= g.M && n0 + TBN <= g.N;
  };
  const int nk = g.K / 32;

  if (w >= NW) {
    // ---------------------------------------------------------------- loader wave
    const int lw = w - NW;
    const bf16* src[NIL];
    auto set_src = [&](int tile) {
      const int m0 = (tile / g.tiles_n) * TBM, n0 = (tile % g.tiles_n) * TBN;
#pragma unroll
      for (int j = 0; j < NIL; ++j) {
        const int ii = lw + NL * j;
        const int r8 = lane >> 3, ch = (lane & 7) ^ r8;          // data chunk landing in this lane's LDS slot
        const int koff = (ch & 3) * 8;                            // chunks 0-3: hi plane, 4-7: lo plane
        if (ii < TBM / 8) src[j] = g.A + (int64_t)min(m0 + ii * 8 + r8, g.M - 1) * g.lda + koff + (ch >> 2) * (g.lda >> 1);
        else src[j] = g.B + (int64_t)min(n0 + (ii - TBM / 8) * 8 + r8, g.N - 1) * g.ldb + koff + (ch >> 2) * (g.ldb >> 1);
      }
    };
    int is_it = 0, is_kt = 0, is_slot = 0;
    int is_tile = tile_of(0);
    if (is_tile >= 0) set_src(is_tile);
    int ahead = 0;
    auto issue_one = [&]() {
      if (is_tile < 0) return;
      unsigned char* st = lds + is_slot * STAGE;
#pragma unroll
      for (int j = 0; j < NIL; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + is_kt * 32),
                                         (__attribute__((address_space(3))) void*)(st + (lw + NL * j) * 1024), 16, 0, 0);
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
        if (ahead - 1 >= S - 2) x3_wait_vmcnt<NIL * (S - 2)>(); else x3_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        --ahead;
        issue_one();
      }
      if (tile_full(tile)) __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ------------------------------------------------------------------ compute wave
  const int wm = w / NWN, wn = w % NWN;
  auto off = [](int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); };
  const int fr = lane & 15, fq = lane >> 4;
  int slot = 0;
  for (int it = 0;; ++it) {
    const int tile = tile_of(it);
    if (tile < 0) break;
    const int m0 = (tile / g.tiles_n) * TBM, n0 = (tile % g.tiles_n) * TBN;
    f32x4 acc[4][WM];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < WM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_barrier();
      const unsigned char* a = lds + slot * STAGE;
      const unsigned char* b = a + TBM * 128;
      slot = slot + 1 == S ? 0 : slot + 1;
      bf16x8 ah[WM], bh[4], bl[4];
#pragma unroll
      for (int i = 0; i < WM; ++i) ah[i] = *(const bf16x8*)(a + off(wm * 16 * WM + i * 16 + fr, fq));
#pragma unroll
      for (int j = 0; j < 4; ++j) bl[j] = *(const bf16x8*)(b + off(wn * 64 + j * 16 + fr, 4 + fq));
#pragma unroll
      for (int j = 0; j < 4; ++j) bh[j] = *(const bf16x8*)(b + off(wn * 64 + j * 16 + fr, fq));
      // small terms first: a_hi * b_lo, a_lo * b_hi, then a_hi * b_hi
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[j][i], 0, 0, 0);
      bf16x8 al[WM];
#pragma unroll
      for (int i = 0; i < WM; ++i) al[i] = *(const bf16x8*)(a + off(wm * 16 * WM + i * 16 + fr, 4 + fq));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[j][i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[j][i], 0, 0, 0);
    }
    if (tile_full(tile)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      float* stg = (float*)(lds + (slot == 0 ? S - 1 : slot - 1) * STAGE) + w * (16 * X3_STG_LD);
      x3_epilogue_staged<CT, EPI, WM>(g, stg, bias_lds, acc, m0 + wm * 16 * WM, n0 + wn * 64, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i)
          x3_epilogue4<CT>(g, m0 + wm * 16 * WM + i * 16 + fr, n0 + wn * 64 + j * 16 + fq * 4, acc[j][i]);
    }
  }
}

template <int CT, int EPI>
static int launch_x3_nt(X3Args& g, hipStream_t st, bool staged_ok) {
  constexpr int TBM = 256, TBN = 128, S = 3;
  constexpr int ring = S * (TBM + TBN) * 128;
  g.tiles_m = (g.M + TBM - 1) / TBM; g.tiles_n = (g.N + TBN - 1) / TBN;
  const int bias_bytes = ((g.N * 4 + 15) / 16) * 16;
  constexpr bool GLU_EPI = EPI == XE_GLU || EPI == XE_GLU_SG || EPI == XE_GLU_BWD;
  g.bias_in_lds = staged_ok && (GLU_EPI || ring + bias_bytes <= 160 * 1024) ? 1 : 0;
  const int shm = ring + (g.bias_in_lds && !GLU_EPI ? bias_bytes : 0);
  auto kern = k_x3_nt<CT, EPI>;
  static AfmOncePerDevice attr_done;   // per instantiation
  if (attr_done.need()) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  int grid = 256;
  const int ntiles = g.tiles_m * g.tiles_n;
  if (grid > ((ntiles + 7) / 8) * 8) grid = ((ntiles + 7) / 8) * 8;
  AFM_LAUNCH(kern, dim3(grid), dim3(768), shm, st, g);
  return AFM_OK;
}

// ------------------------------------------------------------------------------------------ NT, few rows (decode)
// Incremental decode multiplies 128-640 rows: on 256 x 128 tiles that is 4-16 workgroups on the chip, each grinding through a
// tile that is half padding (48 us per GEMM whatever the shape).  Here: 64 x 64 tiles, four waves of 32 x 32, every wave moves
// its own four LDS-DMA pieces per k-step, ring of four 16-KiB stages (three steps in flight), one workgroup per tile, so the
// grid is (M/64) x (N/64) workgroups and a GEMM is a handful of microseconds of LDS-DMA latency.  Fragment epilogue (bias,
// GELU / ReLU, dropout, residual, accumulate, saved pre-activation; pair or fp32 output).
template <int CT>
__global__ __launch_bounds__(256) void k_x3_nt_small(X3Args g) {
  constexpr int TB = 64, S = 4, STAGE = 2 * TB * 128, NP = 2 * TB / 8 / 4;   // pieces per wave and stage = 4
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tile = blockIdx.x;
  const int m0 = (tile / g.tiles_n) * TB, n0 = (tile % g.tiles_n) * TB;
  const int nk = g.K / 32;
  const bf16* src[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int ii = w + 4 * j;                                   // piece 0..15: 0..7 rows of A, 8..15 rows of B
    const int r8 = lane >> 3, ch = (lane & 7) ^ r8;
    const int koff = (ch & 3) * 8;
    if (ii < TB / 8) src[j] = g.A + (int64_t)min(m0 + ii * 8 + r8, g.M - 1) * g.lda + koff + (ch >> 2) * (g.lda >> 1);
    else src[j] = g.B + (int64_t)min(n0 + (ii - TB / 8) * 8 + r8, g.N - 1) * g.ldb + koff + (ch >> 2) * (g.ldb >> 1);
  }
  auto issue = [&](int kt) {
    unsigned char* st = lds + (kt % S) * STAGE;
#pragma unroll
    for (int j = 0; j < NP; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * 32),
                                       (__attribute__((address_space(3))) void*)(st + (w + 4 * j) * 1024), 16, 0, 0);
  };
  auto off = [](int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); };
  const int wm = w >> 1, wn = w & 1, fr = lane & 15, fq = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) issue(s);
  for (int kt = 0; kt < nk; ++kt) {
    const int later = min(S - 2, nk - 1 - kt);                  // younger steps already issued behind step kt
    if (later >= 2) x3_wait_vmcnt<2 * NP>(); else if (later == 1) x3_wait_vmcnt<NP>(); else x3_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + S - 1 < nk) issue(kt + S - 1);
    const unsigned char* a = lds + (kt % S) * STAGE;
    const unsigned char* b = a + TB * 128;
    bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = *(const bf16x8*)(a + off(wm * 32 + i * 16 + fr, fq)); al[i] = *(const bf16x8*)(a + off(wm * 32 + i * 16 + fr, 4 + fq));
      bh[i] = *(const bf16x8*)(b + off(wn * 32 + i * 16 + fr, fq)); bl[i] = *(const bf16x8*)(b + off(wn * 32 + i * 16 + fr, 4 + fq));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[j][i], 0, 0, 0);
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[j][i], 0, 0, 0);
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[j][i], 0, 0, 0);
      }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      x3_epilogue4<CT>(g, m0 + wm * 32 + i * 16 + fr, n0 + wn * 32 + j * 16 + fq * 4, acc[j][i]);
}

template <int CT>
static int launch_x3_nt_small(X3Args& g, hipStream_t st) {
  g.tiles_m = (g.M + 63) / 64; g.tiles_n = (g.N + 63) / 64;
  auto kern = k_x3_nt_small<CT>;
  static AfmOncePerDevice attr_done;   // per instantiation
  if (attr_done.need()) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  }
  AFM_LAUNCH(kern, dim3(g.tiles_m * g.tiles_n), dim3(256), 4 * 2 * 64 * 128, st, g);
  return AFM_OK;
}

// ------------------------------------------------------------------------------------------ NT, 256 x 256 tiles
// Whole-tile problems with many tiles (the encoder's token count): 8 waves of 128 x 64 (2 x 4), no loader waves -- per
// LDS-DMA piece the split-pair form issues three times the MFMAs of the single-pass kernel, so the waves can afford to
// issue their own pieces -- two 64-KiB ring slots, 96 MFMAs per wave between barriers, a quarter less L2->LDS traffic
// per FLOP than 256 x 128.  Whole tiles only (the dispatcher sends everything else to k_x3_nt).
template <int CT, int EPI>
__global__ __launch_bounds__(512) void k_x3_nt256(X3Args g) {
  constexpr int NWN = 4, NW = 8, WM = 8, S = 2;
  constexpr int TBM = 256, TBN = 256;
  constexpr int NI = (TBM + TBN) / 8, NIW = NI / NW;
  constexpr int STAGE = (TBM + TBN) * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* bias_lds = (float*)(lds + S * STAGE);
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w / NWN, wn = w % NWN;
  const int ntiles = g.tiles_m * g.tiles_n;
  for (int n = t; n < g.N; n += 64 * NW) bias_lds[n] = g.bias ? g.bias[n] : 0.f;
  __syncthreads();
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const int tpx = (ntiles + 7) >> 3;
  const int tlo = xcd * tpx, thi = min(ntiles, tlo + tpx);
  auto tile_of = [&](int it) { const int tt = tlo + it * nbx + bx; return tt < thi ? tt : -1; };
  const int nk = g.K / 32;

  const bf16* src[NIW];
  auto set_src = [&](int tile) {
    const int m0 = (tile / g.tiles_n) * TBM, n0 = (tile % g.tiles_n) * TBN;
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
      const int ii = w + NW * j;
      const int r8 = lane >> 3, ch = (lane & 7) ^ r8;
      const int koff = (ch & 3) * 8;
      if (ii < TBM / 8) src[j] = g.A + (int64_t)(m0 + ii * 8 + r8) * g.lda + koff + (ch >> 2) * (g.lda >> 1);
      else src[j] = g.B + (int64_t)(n0 + (ii - TBM / 8) * 8 + r8) * g.ldb + koff + (ch >> 2) * (g.ldb >> 1);
    }
  };
  int is_it = 0, is_kt = 0, is_slot = 0;
  int is_tile = tile_of(0);
  if (is_tile >= 0) set_src(is_tile);
  auto issue_one = [&]() {
    if (is_tile < 0) return;
    unsigned char* st = lds + is_slot * STAGE;
    if (!(g.abl & 2)) {
#pragma unroll
      for (int j = 0; j < NIW; ++j)
        if (!((g.abl & 8) && w + NW * j >= TBM / 8))
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + is_kt * 32),
                                           (__attribute__((address_space(3))) void*)(st + (w + NW * j) * 1024), 16, 0, 0);
    }
    is_slot ^= 1;
    if (++is_kt == nk) {
      is_kt = 0;
      is_tile = tile_of(++is_it);
      if (is_tile >= 0) set_src(is_tile);
    }
  };
  auto off = [](int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); };
  if (g.abl >> 4) {
    const int ph = bx & 3;
    for (int i = 0; i < ph * (g.abl >> 4); ++i) __builtin_amdgcn_s_sleep(127);
  }
  issue_one();
  const int fr = lane & 15, fq = lane >> 4;
  int slot = 0;
  for (int it = 0;; ++it) {
    const int tile = tile_of(it);
    if (tile < 0) break;
    const int m0 = (tile / g.tiles_n) * TBM, n0 = (tile % g.tiles_n) * TBN;
    f32x4 acc[4][WM];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < WM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 bh[4], bl[4];
    for (int kt = 0; kt < nk; ++kt) {
      // only this step's pieces are in flight (two slots); behind them, at a tile boundary, sit the >= 32 stores of the
      // previous tile's epilogue, which may stay outstanding under this tile's first MFMAs (in-order counter)
      if (it > 0 && kt == 0) x3_wait_vmcnt<32>(); else x3_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      issue_one();
      const unsigned char* a = lds + slot * STAGE;
      const unsigned char* b = a + TBM * 128;
      slot ^= 1;
      if (g.abl & 1) continue;
      if (!(g.abl & 16) || kt == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bh[j] = *(const bf16x8*)(b + off(wn * 64 + j * 16 + fr, fq));
          bl[j] = *(const bf16x8*)(b + off(wn * 64 + j * 16 + fr, 4 + fq));
        }
      }
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const bf16x8 ah = *(const bf16x8*)(a + off(wm * 16 * WM + i * 16 + fr, fq));
        const bf16x8 al = *(const bf16x8*)(a + off(wm * 16 * WM + i * 16 + fr, 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah, acc[j][i], 0, 0, 0);
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al, acc[j][i], 0, 0, 0);
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah, acc[j][i], 0, 0, 0);
        }
      }
    }
    // stage through the slot read by the last k-step (the other one is receiving the next tile's first step)
    __builtin_amdgcn_s_barrier();
    float* stg = (float*)(lds + (slot ^ 1) * STAGE) + w * (16 * X3_STG_LD);
    if (g.abl & 4) {   // keep the accumulators alive, store nothing (the next tile's counted wait assumes >= 32 stores: drain)
      if (acc[0][0][0] == 123.456f) ((float*)g.C)[0] = 1.f;
      x3_wait_vmcnt<0>();
    } else {
      x3_epilogue_staged<CT, EPI, WM>(g, stg, bias_lds, acc, m0 + wm * 16 * WM, n0 + wn * 64, lane);
    }
  }
}

template <int CT, int EPI>
static int launch_x3_nt256(X3Args& g, hipStream_t st) {
  constexpr int ring = 2 * 512 * 128;
  g.tiles_m = g.M / 256; g.tiles_n = g.N / 256;
  const int bias_bytes = ((g.N * 4 + 15) / 16) * 16;
  if (ring + bias_bytes > 160 * 1024) return AFM_ERR_UNSUPPORTED;
  g.bias_in_lds = 1;
  auto kern = k_x3_nt256<CT, EPI>;
  static AfmOncePerDevice attr_done;   // per instantiation
  if (attr_done.need()) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  int grid = 256;
  const int ntiles = g.tiles_m * g.tiles_n;
  if (grid > ((ntiles + 7) / 8) * 8) grid = ((ntiles + 7) / 8) * 8;
  AFM_LAUNCH(kern, dim3(grid), dim3(512), ring + bias_bytes, st, g);
  return AFM_OK;
}

// ------------------------------------------------------------------------------------------ TN (wgrad)
// Stage = 64 LDS rows x (256 A columns + 128 B columns): rows 0-31 the hi plane of 32 token rows, rows 32-63 their lo
// plane; 16-byte chunks XOR-swizzled as in k_gemm_tn_ring; fragments by ds_read_b64_tr_b16 (inline asm, hand-waited).
__device__ __forceinline__ int x3_tn_swz(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }

__global__ __launch_bounds__(512, 2) void k_x3_tn(X3Args g) {
  constexpr int S = 3, TBM = 256, TBN = 128, NW = 8, NIW = 6;
  constexpr int ABYTES = 64 * TBM * 2, STAGE = 64 * (TBM + TBN) * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int ntile = g.tiles_m * g.tiles_n;
  // tile index fastest: the workgroups of one XCD share a k-chunk (dy / x rows served from that XCD's L2)
  const int q8 = (ntile * g.ksplit) >> 3, r8 = (ntile * g.ksplit) & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tile = bid % ntile, ks_id = bid / ntile;
  const int m0 = (tile / g.tiles_n) * TBM, n0 = (tile % g.tiles_n) * TBN;
  const int kbeg = ks_id * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  const int nk = (kend - kbeg) / 32;

  const bf16* src[NIW];
  int64_t pitch[NIW];
#pragma unroll
  for (int j = 0; j < NIW; ++j) {
    const int ii = w + NW * j;
    if (ii < 32) {
      const int r = ii * 2 + (lane >> 5);                 // LDS row 0..63: plane r >> 5, token row r & 31
      const int c = (lane & 31) ^ x3_tn_swz(r);
      src[j] = g.A + (int64_t)(kbeg + (r & 31)) * g.lda + (r >> 5) * (g.lda >> 1) + min(m0 + c * 8, g.M - 8);
      pitch[j] = (int64_t)32 * g.lda;
    } else {
      const int r = (ii - 32) * 4 + (lane >> 4);
      const int c = (lane & 15) ^ x3_tn_swz(r);
      src[j] = g.B + (int64_t)(kbeg + (r & 31)) * g.ldb + (r >> 5) * (g.ldb >> 1) + min(n0 + c * 8, g.N - 8);
      pitch[j] = (int64_t)32 * g.ldb;
    }
  }
  auto issue = [&](int kt) {
    unsigned char* st = lds + (kt % S) * STAGE;
#pragma unroll
    for (int j = 0; j < NIW; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * pitch[j]),
                                       (__attribute__((address_space(3))) void*)(st + (w + NW * j) * 1024), 16, 0, 0);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient: the k-steps are dealt round-robin over the 2 x tiles_n waves of the tile row that hold the same A fragments
  // (afm_gemm_mfma_impl.h, k_gemm_tn_ring), instead of one wave per tile row summing all of them
  const bool do_cs = g.a_colsum != nullptr;
  const int cs_slots = 2 * g.tiles_n;
  int cs_next = (tile % g.tiles_n) * 2 + wn;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) issue(s);
  const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  int laneA[4], laneB[4];
  {
    const int lrow = grp * 8 + q, swz = x3_tn_swz(lrow), sub = (p & 1) << 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cha = ((wm * 64 + i * 16) >> 3) + (p >> 1), chb = ((wn * 64 + i * 16) >> 3) + (p >> 1);
      laneA[i] = lrow * 512 + ((cha ^ swz) << 4) + sub;
      laneB[i] = ABYTES + lrow * 256 + ((chb ^ swz) << 4) + sub;
    }
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int later = min(S - 2, nk - 1 - kt);
    if (later >= S - 2) x3_wait_vmcnt<NIW * (S - 2)>(); else x3_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + S - 1 < nk) issue(kt + S - 1);
    const unsigned sbase = (unsigned)(uintptr_t)(lds + (kt % S) * STAGE);
    unsigned va[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { va[i] = sbase + (unsigned)laneA[i]; vb[i] = sbase + (unsigned)laneB[i]; }
    x3_s16x4 a0[2][4], a1[2][4], b0[2][4], b1[2][4];      // [plane][fragment]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a0[0][i]) : "v"(va[i]));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(a1[0][i]) : "v"(va[i]));
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0[0][i]) : "v"(vb[i]));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(b1[0][i]) : "v"(vb[i]));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:16384" : "=v"(a0[1][i]) : "v"(va[i]));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:18432" : "=v"(a1[1][i]) : "v"(va[i]));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(b0[1][i]) : "v"(vb[i]));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9216" : "=v"(b1[1][i]) : "v"(vb[i]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 af[2][4], bfr[2][4];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4 x0 = __builtin_bit_cast(bf16x4, a0[pl][i]), x1 = __builtin_bit_cast(bf16x4, a1[pl][i]);
        const bf16x4 y0 = __builtin_bit_cast(bf16x4, b0[pl][i]), y1 = __builtin_bit_cast(bf16x4, b1[pl][i]);
        af[pl][i] = (bf16x8){x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        bfr[pl][i] = (bf16x8){y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
      }
    if (do_cs && kt == cs_next) {   // column sums of dy = hi + lo (lane: column fr, 8 token rows)
      cs_next += cs_slots;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[i] += (float)af[0][i][j] + (float)af[1][i][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bfr[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][i], bfr[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
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
      if (fq == 0 && mm < g.M) atomicAdd(g.a_colsum + (g.glu_f ? x3_glu_deint(mm, g.glu_f) : mm), s);
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
          float* c = C + (int64_t)(g.glu_f ? x3_glu_deint(mm, g.glu_f) : mm) * g.ldc + n;
          if (g.ksplit > 1) atomicAdd(c, acc[i][j][r]);
          else *c = acc[i][j][r] + (g.accumulate ? *c : 0.f);
        }
      }
    }
}

// Same kernel on 256 x 256 tiles (8 waves of 128 x 64, two 64-KiB ring slots): a quarter less L2->LDS fill and a quarter
// fewer transposed LDS reads per FLOP, 96 MFMAs per wave between barriers; needs more split-K (fewer tiles).
__global__ __launch_bounds__(512) void k_x3_tn256(X3Args g) {
  constexpr int S = 2, TBM = 256, TBN = 256, NW = 8, NIW = 8;
  constexpr int ABYTES = 64 * TBM * 2, STAGE = 64 * (TBM + TBN) * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int ntile = g.tiles_m * g.tiles_n;
  const int q8 = (ntile * g.ksplit) >> 3, r8 = (ntile * g.ksplit) & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tile = bid % ntile, ks_id = bid / ntile;
  const int m0 = (tile / g.tiles_n) * TBM, n0 = (tile % g.tiles_n) * TBN;
  const int kbeg = ks_id * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  const int nk = (kend - kbeg) / 32;

  const bf16* src[NIW];
  int64_t pitch[NIW];
#pragma unroll
  for (int j = 0; j < NIW; ++j) {
    const int ii = w + NW * j;
    const int r = (ii & 31) * 2 + (lane >> 5);          // LDS row 0..63: plane r >> 5, token row r & 31
    const int c = (lane & 31) ^ x3_tn_swz(r);
    if (ii < 32) {
      src[j] = g.A + (int64_t)(kbeg + (r & 31)) * g.lda + (r >> 5) * (g.lda >> 1) + min(m0 + c * 8, g.M - 8);
      pitch[j] = (int64_t)32 * g.lda;
    } else {
      src[j] = g.B + (int64_t)(kbeg + (r & 31)) * g.ldb + (r >> 5) * (g.ldb >> 1) + min(n0 + c * 8, g.N - 8);
      pitch[j] = (int64_t)32 * g.ldb;
    }
  }
  auto issue = [&](int kt) {
    unsigned char* st = lds + (kt % S) * STAGE;
#pragma unroll
    for (int j = 0; j < NIW; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * pitch[j]),
                                       (__attribute__((address_space(3))) void*)(st + (w + NW * j) * 1024), 16, 0, 0);
  };
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_cs = g.a_colsum != nullptr;     // k-steps dealt over the 4 x tiles_n waves holding the same A fragments (see k_x3_tn)
  const int cs_slots = 4 * g.tiles_n;
  int cs_next = (tile % g.tiles_n) * 4 + wn;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (nk > 0) issue(0);
  const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  int laneA[8], laneB[4];
  {
    const int lrow = grp * 8 + q, swz = x3_tn_swz(lrow), sub = (p & 1) << 3;
#pragma unroll
    for (int i = 0; i < 8; ++i) laneA[i] = lrow * 512 + (((((wm * 128 + i * 16) >> 3) + (p >> 1)) ^ swz) << 4) + sub;
#pragma unroll
    for (int j = 0; j < 4; ++j) laneB[j] = ABYTES + lrow * 512 + (((((wn * 64 + j * 16) >> 3) + (p >> 1)) ^ swz) << 4) + sub;
  }
  for (int kt = 0; kt < nk; ++kt) {
    x3_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) issue(kt + 1);
    const unsigned sbase = (unsigned)(uintptr_t)(lds + (kt % S) * STAGE);
    x3_s16x4 b0[2][4], b1[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned vb = sbase + (unsigned)laneB[j];
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0[0][j]) : "v"(vb));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(b1[0][j]) : "v"(vb));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:16384" : "=v"(b0[1][j]) : "v"(vb));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:18432" : "=v"(b1[1][j]) : "v"(vb));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 bfr[2][4];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x4 y0 = __builtin_bit_cast(bf16x4, b0[pl][j]), y1 = __builtin_bit_cast(bf16x4, b1[pl][j]);
        bfr[pl][j] = (bf16x8){y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
      }
    const bool cs_now = do_cs && kt == cs_next;
    if (cs_now) cs_next += cs_slots;
#pragma unroll
    for (int ih = 0; ih < 2; ++ih) {        // four A fragments at a time (register budget)
      x3_s16x4 a0[2][4], a1[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned va = sbase + (unsigned)laneA[ih * 4 + i];
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a0[0][i]) : "v"(va));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(a1[0][i]) : "v"(va));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:16384" : "=v"(a0[1][i]) : "v"(va));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:18432" : "=v"(a1[1][i]) : "v"(va));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4 h0 = __builtin_bit_cast(bf16x4, a0[0][i]), h1 = __builtin_bit_cast(bf16x4, a1[0][i]);
        const bf16x4 l0 = __builtin_bit_cast(bf16x4, a0[1][i]), l1 = __builtin_bit_cast(bf16x4, a1[1][i]);
        const bf16x8 ah = (bf16x8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        const bf16x8 al = (bf16x8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        if (cs_now) {
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[ih * 4 + i] += (float)ah[e] + (float)al[e];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[ih * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bfr[1][j], acc[ih * 4 + i][j], 0, 0, 0);
          acc[ih * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bfr[0][j], acc[ih * 4 + i][j], 0, 0, 0);
          acc[ih * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bfr[0][j], acc[ih * 4 + i][j], 0, 0, 0);
        }
      }
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
      if (fq == 0 && mm < g.M) atomicAdd(g.a_colsum + (g.glu_f ? x3_glu_deint(mm, g.glu_f) : mm), s);
    }
  }
  float* C = (float*)g.C;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = m0 + wm * 128 + i * 16 + fq * 4 + r;
        if (mm < g.M && n < g.N) {
          float* c = C + (int64_t)(g.glu_f ? x3_glu_deint(mm, g.glu_f) : mm) * g.ldc + n;
          if (g.ksplit > 1) atomicAdd(c, acc[i][j][r]);
          else *c = acc[i][j][r] + (g.accumulate ? *c : 0.f);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------ dispatch
static inline bool x3_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int afm_gemm_x3_try(const afm_gemm_desc* d, hipStream_t st) {
  if (d->a_dtype != AFM_BF16X2 || d->b_dtype != AFM_BF16X2) return AFM_ERR_UNSUPPORTED;
  X3Args g;
  g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
  g.A = (const bf16*)d->A; g.B = (const bf16*)d->B; g.C = d->C;
  g.bias = d->bias; g.residual = d->residual; g.pre_act = d->pre_act; g.a_colsum = d->a_colsum;
  g.act = d->act; g.accumulate = d->accumulate;
  g.dd = afm_make_drop(&d->drop);
  g.tiles_m = g.tiles_n = 0; g.ksplit = 1; g.kchunk = d->K; g.bias_in_lds = 0;
  g.abl = (d->reserved >= 320 && d->reserved < 352) ? d->reserved - 320 : 0;   // + 8: no LDS-DMA of the B rows, + 16: B fragments read once per tile
  if (d->reserved >= 400 && d->reserved < 464) g.abl = (d->reserved - 400) << 4;   // stagger experiment: sleeps per phase
  g.glu_f = d->glu_rows;
  g.sg_hi_only = d->reserved2 & 1;
  // 16-byte pieces of both planes: pointers 16-byte aligned, plane offsets (ld / 2) multiples of 8 elements
  if (!x3_al16(d->A) || !x3_al16(d->B) || (d->lda & 15) || (d->ldb & 15)) return AFM_ERR_UNSUPPORTED;
  if (!d->transA && d->transB) {   // NT
    if ((d->K & 31) || d->K < 32 || d->N < 16) return AFM_ERR_UNSUPPORTED;
    if (d->c_dtype == AFM_BF16) return AFM_ERR_UNSUPPORTED;
    if (d->bias && !x3_al16(d->bias)) return AFM_ERR_UNSUPPORTED;
    if (!x3_al16(d->C) || (d->residual && !x3_al16(d->residual)) || (d->pre_act && !x3_al16(d->pre_act))) return AFM_ERR_UNSUPPORTED;
    const bool small_idx = (uint64_t)d->M * (uint64_t)d->N <= 0x100000000ull;
    if (d->act >= AFM_ACT_GLU) {
      // fused gated FFN: whole 256 x 128 tiles, split-pair in / out, contiguous C and saved tensor
      const int ncol_c = d->act == AFM_ACT_GLU_BWD ? 2 * d->N : d->N / 2;
      if ((d->M & 255) || (d->N & 127) || d->c_dtype != AFM_BF16X2 || d->residual || d->accumulate || d->ldc != 2 * ncol_c ||
          (d->drop.p > 0.f && !small_idx) || (d->act == AFM_ACT_GLU_BWD && (d->bias || d->drop.p > 0.f)))
        return AFM_ERR_UNSUPPORTED;
      int r;
      if (d->act == AFM_ACT_GLU) r = launch_x3_nt<X3_X2, XE_GLU>(g, st, true);
      else if (d->act == AFM_ACT_GLU_SAVE) r = launch_x3_nt<X3_X2, XE_GLU_SG>(g, st, true);
      else r = launch_x3_nt<X3_X2, XE_GLU_BWD>(g, st, true);
      if (r != AFM_OK) return r;
      afm_set_last_algo("mfma_nt_x3_glu");
      return AFM_OK;
    }
    // few rows (incremental decode): 64 x 64 tiles over the whole chip instead of a handful of 256 x 128 tiles; reserved = 33 forces it
    const int64_t tiles_256x128 = (int64_t)((d->M + 255) / 256) * ((d->N + 127) / 128);
    if ((d->reserved == 33 || (d->reserved == 0 && tiles_256x128 < 64 && d->M <= 1024)) && d->act <= AFM_ACT_GELU && (d->ldc & 1) == 0 &&
        !(d->c_dtype == AFM_F32 && (d->act != AFM_ACT_NONE || d->pre_act || d->drop.p > 0.f)) &&
        (uint64_t)d->M * (uint64_t)d->N <= 0x100000000ull) {
      const int r = d->c_dtype == AFM_F32 ? launch_x3_nt_small<X3_F32>(g, st) : launch_x3_nt_small<X3_X2>(g, st);
      if (r != AFM_OK) return r;
      afm_set_last_algo("mfma_nt_x3_small");
      return AFM_OK;
    }
    // 256 x 256 tiles: whole tiles, enough of them to fill the chip twice over; reserved = 31 / 32 force a form (tools)
    const bool big = d->reserved != 31 && !(d->M & 255) && !(d->N & 255) && (d->ldc % 8) == 0 &&
                     (d->reserved == 32 || g.abl || (int64_t)(d->M >> 8) * (d->N >> 8) >= 512);
    int r;
    if (d->c_dtype == AFM_F32) {
      if (d->act != AFM_ACT_NONE || d->pre_act || d->drop.p > 0.f) return AFM_ERR_UNSUPPORTED;   // exact-fp32 kernel
      const bool staged = (d->N % 4) == 0 && (d->ldc % 4) == 0;
      if (big && staged) r = launch_x3_nt256<X3_F32, XE_GENERIC>(g, st);
      else r = launch_x3_nt<X3_F32, XE_GENERIC>(g, st, staged);
    } else if (big && !d->residual && !d->accumulate && (small_idx || d->drop.p <= 0.f) && (d->ldc & 15) == 0 &&
               d->act != AFM_ACT_RELU && !(d->act == AFM_ACT_NONE && (d->pre_act || d->drop.p > 0.f))) {
      if (d->act == AFM_ACT_GELU_SAVE_GRAD) r = launch_x3_nt256<X3_X2, XE_GELU_SG>(g, st);
      else if (d->act == AFM_ACT_MUL_SAVED) r = launch_x3_nt256<X3_X2, XE_MUL>(g, st);
      else if (d->act == AFM_ACT_GELU_BWD) r = launch_x3_nt256<X3_X2, XE_GELU_BWD>(g, st);
      else if (d->act == AFM_ACT_GELU) r = launch_x3_nt256<X3_X2, XE_GELU>(g, st);
      else r = launch_x3_nt256<X3_X2, XE_PLAIN>(g, st);
    } else {
      if ((d->ldc & 15) || d->act == AFM_ACT_RELU) return AFM_ERR_UNSUPPORTED;
      // compile-time staged epilogues for the training step's forms; anything else keeps the fragment epilogue
      const bool plainish = !d->residual && !d->accumulate && (d->N % 8) == 0 && (small_idx || d->drop.p <= 0.f);
      if (d->act == AFM_ACT_GELU_SAVE_GRAD) r = launch_x3_nt<X3_X2, XE_GELU_SG>(g, st, plainish);
      else if (d->act == AFM_ACT_MUL_SAVED) r = launch_x3_nt<X3_X2, XE_MUL>(g, st, plainish);
      else if (d->act == AFM_ACT_GELU_BWD) r = launch_x3_nt<X3_X2, XE_GELU_BWD>(g, st, plainish);
      else if (d->act == AFM_ACT_GELU) r = launch_x3_nt<X3_X2, XE_GELU>(g, st, plainish);
      else r = launch_x3_nt<X3_X2, XE_PLAIN>(g, st, plainish && !d->pre_act && d->drop.p <= 0.f);
    }
    if (r != AFM_OK) return r;
    const bool ran256 = big && (d->c_dtype == AFM_F32 ? (d->N % 4) == 0 && (d->ldc % 4) == 0
                                : (!d->residual && !d->accumulate && (small_idx || d->drop.p <= 0.f) && (d->ldc & 15) == 0 &&
                                   d->act != AFM_ACT_RELU && !(d->act == AFM_ACT_NONE && (d->pre_act || d->drop.p > 0.f))));
    afm_set_last_algo(ran256 ? "mfma_nt_x3_256" : "mfma_nt_x3");     // (_256: the 256 x 256-tile form)
    return AFM_OK;
  }
  if (d->transA && !d->transB) {   // TN: the wgrad form
    if (d->c_dtype != AFM_F32 || d->bias || d->residual || d->pre_act || d->act != AFM_ACT_NONE || d->drop.p > 0.f ||
        (d->a_colsum && ((uintptr_t)d->a_colsum & 3)))
      return AFM_ERR_UNSUPPORTED;
    if ((d->M & 7) || (d->N & 7) || (d->K & 31) || d->K < 32 || d->M < 64 || d->N < 64) return AFM_ERR_UNSUPPORTED;
    // 256 x 256 tiles for long token counts and gradient matrices with >= 8 such tiles (as the single-pass kernel);
    // reserved = 105 / 106 force a form (tools)
    const bool want256 = d->reserved == 105 || (d->reserved != 106 && d->K >= 65536 &&
                                                 (int64_t)((d->M + 255) / 256) * ((d->N + 255) / 256) >= 8);
    if (want256 && d->M >= 256 && d->N >= 256) {
      g.tiles_m = (d->M + 255) / 256; g.tiles_n = (d->N + 255) / 256;
      const int tiles = g.tiles_m * g.tiles_n;
      int ksplit = tiles >= 256 ? 1 : 256 / tiles;
      const int maxs = d->K / 512;
      if (ksplit > maxs) ksplit = maxs;
      if (ksplit < 1) ksplit = 1;
      int kchunk = ((d->K / 32 + ksplit - 1) / ksplit) * 32;
      ksplit = (d->K + kchunk - 1) / kchunk;
      g.ksplit = ksplit; g.kchunk = kchunk;
      if (ksplit > 1 && !d->accumulate) {
        if (hipMemset2DAsync(d->C, sizeof(float) * d->ldc, 0, sizeof(float) * d->N, d->M, st) != hipSuccess) return AFM_ERR_LAUNCH;
      }
      static AfmOncePerDevice attr256;
      if (attr256.need()) {
        (void)hipFuncSetAttribute((const void*)k_x3_tn256, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 512 * 2);
      }
      AFM_LAUNCH(k_x3_tn256, dim3(tiles * ksplit), dim3(512), 2 * 64 * 512 * 2, st, g);
      afm_set_last_algo(ksplit > 1 ? "mfma_tn_x3_256_splitk" : "mfma_tn_x3_256");
      return AFM_OK;
    }
    g.tiles_m = (d->M + 255) / 256; g.tiles_n = (d->N + 127) / 128;
    const int tiles = g.tiles_m * g.tiles_n;
    int ksplit = tiles >= 256 ? 1 : 256 / tiles;
    const int maxs = d->K / 512;              // >= 16 k-steps per workgroup
    if (ksplit > maxs) ksplit = maxs;
    if (ksplit < 1) ksplit = 1;
    int kchunk = ((d->K / 32 + ksplit - 1) / ksplit) * 32;
    ksplit = (d->K + kchunk - 1) / kchunk;
    g.ksplit = ksplit; g.kchunk = kchunk;
    if (ksplit > 1 && !d->accumulate) {
      if (hipMemset2DAsync(d->C, sizeof(float) * d->ldc, 0, sizeof(float) * d->N, d->M, st) != hipSuccess) return AFM_ERR_LAUNCH;
    }
    static AfmOncePerDevice attr_done;
    if (attr_done.need()) {
      (void)hipFuncSetAttribute((const void*)k_x3_tn, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 384 * 2);
    }
    AFM_LAUNCH(k_x3_tn, dim3(tiles * ksplit), dim3(512), 3 * 64 * 384 * 2, st, g);
    afm_set_last_algo(ksplit > 1 ? "mfma_tn_x3_splitk" : "mfma_tn_x3");
    return AFM_OK;
  }
  return AFM_ERR_UNSUPPORTED;
}
