// "bf16x3" flash attention for gfx950, head size 64: Q / K / V / O / dO and the gradients are split bf16 pairs
// (AFM_BF16X2, value = hi + lo), every product of the single-pass kernels (afm_attn_mfma.hip) runs as three
// v_mfma_f32_32x32x16_bf16 passes  hi*hi + hi*lo + lo*hi  with fp32 accumulation; probabilities and score gradients
// are split in registers before they become MFMA operands.  Same orientation and LDS images as the single-pass
// kernels (S^T = K Q^T with the query on the lane, LDS-DMA tile ring, transposed reads from inline asm); each tile
// image exists twice (hi plane, lo plane).  Masks, the 16-bit dropout stream and lse / delta are identical.
//
//   forward   4 waves x 32 queries, 64-key tiles (K row + V tr images, hi and lo: 32 KiB per stage, 2 stages)
//   dQ        4 waves x 32 queries, 32-key tiles (K dual-use image, V row image, hi and lo: 16 KiB per stage)
//   dK / dV   4 waves x 32 keys,    32-query tiles (Q and dO dual-use images, hi and lo: 16 KiB per stage; the wave's
//             K / V lo fragments parked in LDS)
#include "afm_attn_tiles.h"

#define IMG64 (64 * DH * 2)   // one 64-row image
#define IMG32 (32 * DH * 2)   // one 32-row image

__device__ __forceinline__ void split8(const f32x16& x, int s, bf16x8& hi, bf16x8& lo) {   // x: VALU results (see afm_split2)
  afm_u32x4 h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) { uint32_t a, b; afm_split2(x[8 * s + 2 * j], x[8 * s + 2 * j + 1], a, b); h[j] = a; l[j] = b; }
  hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);
}
// the three passes of one split product: acc += A * B with A = (ah, al), B = (bh, bl)
__device__ __forceinline__ f32x16 mfma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 c) {
  c = mfma32(ah, bl, c);
  c = mfma32(al, bh, c);
  return mfma32(ah, bh, c);
}
// split-pair row store of a lane's 4 consecutive columns
__device__ __forceinline__ void store4_x2(bf16* p, int lo, float a, float b, float c, float d) {
  bf16 h0, l0, h1, l1, h2, l2, h3, l3;
  afm_split(a, h0, l0); afm_split(b, h1, l1); afm_split(c, h2, l2); afm_split(d, h3, l3);
  *(bf16x4*)p = (bf16x4){h0, h1, h2, h3};
  *(bf16x4*)(p + lo) = (bf16x4){l0, l1, l2, l3};
}
__device__ __forceinline__ uint32_t mask32_of(const unsigned long long* maskw, int kt) {   // 32-key tile kt
  return (uint32_t)(maskw[kt >> 1] >> (32 * (kt & 1)));
}

// ------------------------------------------------------------------------------------------ forward
template <int DROP>
__global__ __launch_bounds__(256, 2) void k_attn_fwd_x3(AttnM a, const bf16* __restrict__ Q,
                                                        const bf16* __restrict__ K,
                                                        const bf16* __restrict__ V, bf16* __restrict__ O,
                                                        float* __restrict__ lse) {
  constexpr int STAGE = 4 * IMG64;   // K hi, K lo (row images), V hi, V lo (tr images)
  constexpr int NS = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned long long* maskw = (unsigned long long*)(lds + NS * STAGE);
  const int t = threadIdx.x, lane = t & 63, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tq + 127) / 128);
  const int hd = blk_.hd, b = blk_.b;
  const int q0 = blk_.xb * 128 + w * 32;
  const int q = q0 + (lane & 31);
  const int qc = q < a.Tq ? q : a.Tq - 1;
  const int lok = a.ldk >> 1, lov = a.ldv >> 1;
  const bf16* Kb = K + (int64_t)b * a.Tk * a.ldk + hd * DH;
  const bf16* Vb = V + (int64_t)b * a.Tk * a.ldv + hd * DH;
  int kend = a.Tk;
  if (a.causal) kend = min(a.Tk, blk_.xb * 128 + 128);
  const int ntiles = (kend + KT - 1) / KT;
  bf16x8 qh[4], ql[4];
  {
    const bf16* qp = Q + ((int64_t)b * a.Tq + qc) * a.ldq + hd * DH + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) { qh[s] = *(const bf16x8*)(qp + 16 * s); ql[s] = *(const bf16x8*)(qp + (a.ldq >> 1) + 16 * s); }
  }
  build_mask_words(maskw, a.key_pad, b, a.Tk, ntiles, w, lane);
  __syncthreads();
  auto issue = [&](int kt) {
    unsigned char* st = lds + (kt % NS) * STAGE;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      dma_piece<false>(st, Kb, a.ldk, kt * KT, a.Tk, w + 4 * u, lane);
      dma_piece<false>(st + IMG64, Kb + lok, a.ldk, kt * KT, a.Tk, w + 4 * u, lane);
      dma_piece<true>(st + 2 * IMG64, Vb, a.ldv, kt * KT, a.Tk, w + 4 * u, lane);
      dma_piece<true>(st + 3 * IMG64, Vb + lov, a.ldv, kt * KT, a.Tk, w + 4 * u, lane);
    }
  };
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const uint32_t rowbase = afm_row_hash(a.dd, (uint64_t)(b * a.H + hd) * a.Tq + qc);   // the lane's ROW HASH (two-level dropout stream, afm_common.h)
  issue(0);
  __builtin_assume(ntiles >= 1);
  for (int kt = 0; kt < ntiles; ++kt) {
    const int kb = kt * KT;
    attn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < ntiles) issue(kt + 1);
    if ((a.causal && kb > q0 + 31) || maskw[kt] == ~0ull) continue;
    const unsigned char* Kh = lds + (kt % NS) * STAGE;
    const unsigned char* Kl = Kh + IMG64;
    const unsigned char* Vh = Kh + 2 * IMG64;
    const unsigned long long mword = maskw[kt];
    const unsigned long long pad = mword >> (4 * h);
    f32x16 s[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[blk][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        s[blk] = mfma3(frag_row(Kh, 32 * blk, ks, lane), frag_row(Kl, 32 * blk, ks, lane), qh[ks], ql[ks], s[blk]);
    }
    const bool diag = a.causal && (kb + KT - 1 > q0);
    if (mword != 0ull || diag) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ko = 32 * blk + ACC_ROW(r);
          bool msk = (pad >> ko) & 1ull;
          if (a.causal) msk = msk || (kb + ko + 4 * h > q);
          s[blk][r] = msk ? -INFINITY : s[blk][r];
        }
    }
    float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = max3_raw(mt, s[0][r], s[1][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * a.scale_log2;
    const float mn = fmaxf(m, mt);
    const float ms = mn == -INFINITY ? 0.f : mn;
    const float alpha = fast_exp2(m - ms);
    m = mn;
    float ls = 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(fmaf(s[blk][r], a.scale_log2, -ms));
        s[blk][r] = p;
        ls += p;
      }
    l = l * alpha + ls;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
    if (DROP == DROP_HASH) {
      drop_block(a.dd, rowbase, kb, h, s[0]);
      drop_block(a.dd, rowbase, kb + 32, h, s[1]);
    }
    if (DROP == DROP_BITS) {   // the same dropout, and the keep bits of both 32-key blocks go to the keep-bit tensor
      unsigned long long* bb = bits_block(a, b * a.H + hd, q0 >> 5, kb >> 5);
      drop_block_emit(a.dd, rowbase, kb, h, s[0], bb);
      drop_block_emit(a.dd, rowbase, kb + 32, h, s[1], bb + 16);
    }
    unsigned vh0, vh1;
    tr_lane_addr(Vh, lane, vh0, vh1);
    const unsigned vl0 = vh0 + IMG64, vl1 = vh1 + IMG64;
    TrQuad vqh[2], vql[2];
    vqh[0] = tr_issue(vh0, vh1, 0); vql[0] = tr_issue(vl0, vl1, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // 16-key slice i = 2 blk + ks
      if (i < 3) { vqh[(i + 1) & 1] = tr_issue(vh0, vh1, 16 * (i + 1)); vql[(i + 1) & 1] = tr_issue(vl0, vl1, 16 * (i + 1)); }
      bf16x8 ph, pl;
      split8(s[i >> 1], i & 1, ph, pl);
      if (i < 3) tr_wait<8>(); else tr_wait<0>();
      const TrQuad& vh = vqh[i & 1];
      const TrQuad& vl = vql[i & 1];
      o[0] = mfma3(tr_join(vh.lo0, vh.hi0), tr_join(vl.lo0, vl.hi0), ph, pl, o[0]);
      o[1] = mfma3(tr_join(vh.lo1, vh.hi1), tr_join(vl.lo1, vl.hi1), ph, pl, o[1]);
    }
  }
  if (DROP == DROP_BITS) bits_flush();
  l += __shfl_xor(l, 32, 64);
  const float inv = l > 0.f ? a.dd.scale16 / l : 0.f;
  if (q < a.Tq) {
    bf16* op = O + ((int64_t)b * a.Tq + q) * a.ldo + hd * DH + 4 * h;
    const int loo = a.ldo >> 1;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        store4_x2(op + 32 * db + 8 * g4, loo, o[db][4 * g4 + 0] * inv, o[db][4 * g4 + 1] * inv,
                  o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv);
    if (h == 0) lse[((int64_t)b * a.H + hd) * a.Tq + q] = l > 0.f ? (m + __log2f(l)) * 0.69314718055994531f : INFINITY;
  }
}

// ------------------------------------------------------------------------------------------ dQ
template <int DROP>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dq_x3(AttnM a, const bf16* __restrict__ Q,
                                                           const bf16* __restrict__ K,
                                                           const bf16* __restrict__ V,
                                                           const bf16* __restrict__ O,
                                                           const bf16* __restrict__ dO,
                                                           const float* __restrict__ lse,
                                                           float* __restrict__ delta, bf16* __restrict__ dQ) {
  constexpr int KT2 = 32, NS = 2;
  constexpr int STAGE = 4 * IMG32;   // K hi / lo as dual-use images (row reads for S^T, transposed reads for dQ^T), V row hi / lo
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned long long* maskw = (unsigned long long*)(lds + NS * STAGE);
  const int t = threadIdx.x, lane = t & 63, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tq + 127) / 128);
  const int hd = blk_.hd, b = blk_.b;
  const int q0 = blk_.xb * 128 + w * 32;
  const int q = q0 + (lane & 31);
  const int qc = q < a.Tq ? q : a.Tq - 1;
  const int lok = a.ldk >> 1, lov = a.ldv >> 1, loq = a.ldq >> 1, loo = a.ldo >> 1;
  const bf16* Kb = K + (int64_t)b * a.Tk * a.ldk + hd * DH;
  const bf16* Vb = V + (int64_t)b * a.Tk * a.ldv + hd * DH;
  bf16x8 qh[4], ql[4], dh_[4], dl_[4];
  float dl = 0.f;
  {
    const bf16* qp = Q + ((int64_t)b * a.Tq + qc) * a.ldq + hd * DH + 8 * h;
    const bf16* dop = dO + ((int64_t)b * a.Tq + qc) * a.ldo + hd * DH + 8 * h;
    const bf16* op = O + ((int64_t)b * a.Tq + qc) * a.ldo + hd * DH + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qh[s] = *(const bf16x8*)(qp + 16 * s); ql[s] = *(const bf16x8*)(qp + loq + 16 * s);
      dh_[s] = *(const bf16x8*)(dop + 16 * s); dl_[s] = *(const bf16x8*)(dop + loo + 16 * s);
      const bf16x8 oh = *(const bf16x8*)(op + 16 * s), ol = *(const bf16x8*)(op + loo + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += ((float)dh_[s][j] + (float)dl_[s][j]) * ((float)oh[j] + (float)ol[j]);
    }
  }
  dl += __shfl_xor(dl, 32, 64);
  const int64_t lrow = ((int64_t)b * a.H + hd) * a.Tq + qc;
  if (q < a.Tq && h == 0) delta[lrow] = dl;
  const float L = lse[lrow];
  const float L2 = L == INFINITY ? INFINITY : L * 1.4426950408889634f;
  f32x16 dq[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }
  const uint32_t rowbase = afm_row_hash(a.dd, (uint64_t)lrow);   // the lane's ROW HASH (two-level dropout stream, afm_common.h)

  int kend = a.Tk;
  if (a.causal) kend = min(a.Tk, blk_.xb * 128 + 128);
  const int ntiles = (kend + KT2 - 1) / KT2;
  build_mask_words(maskw, a.key_pad, b, a.Tk, (kend + 63) / 64, w, lane);
  __syncthreads();
  auto issue = [&](int kt) {   // 4 images x 4 pieces: wave w moves piece w of every image
    unsigned char* st = lds + (kt % NS) * STAGE;
    dma_piece_dual(st, Kb, a.ldk, kt * KT2, a.Tk, w, lane);
    dma_piece_dual(st + IMG32, Kb + lok, a.ldk, kt * KT2, a.Tk, w, lane);
    dma_piece<false>(st + 2 * IMG32, Vb, a.ldv, kt * KT2, a.Tk, w, lane);
    dma_piece<false>(st + 3 * IMG32, Vb + lov, a.ldv, kt * KT2, a.Tk, w, lane);
  };
  const unsigned t0 = tr_dual_t0(lane);
  issue(0);
  __builtin_assume(ntiles >= 1);
  for (int kt = 0; kt < ntiles; ++kt) {
    const int kb = kt * KT2;
    attn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < ntiles) issue(kt + 1);
    const uint32_t mword = mask32_of(maskw, kt);
    if ((a.causal && kb > q0 + 31) || mword == 0xFFFFFFFFu) continue;
    const unsigned char* Krh = lds + (kt % NS) * STAGE;
    const unsigned char* Krl = Krh + IMG32;
    const unsigned char* Vrh = Krh + 2 * IMG32;
    const unsigned char* Vrl = Krh + 3 * IMG32;
    const uint32_t pad = mword >> (4 * h);
    KeepMasks km;
    if (DROP == DROP_BITS) keep_masks_issue(km, bits_block(a, b * a.H + hd, q0 >> 5, kt));
    f32x16 s, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s = mfma3(frag_row_dual(Krh, 0, ks, lane), frag_row_dual(Krl, 0, ks, lane), qh[ks], ql[ks], s);
      dp = mfma3(frag_row(Vrh, 0, ks, lane), frag_row(Vrl, 0, ks, lane), dh_[ks], dl_[ks], dp);
    }
    if (DROP == DROP_HASH) {
      drop_block(a.dd, rowbase, kb, h, dp);
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] *= a.dd.scale16;
    }
    if (DROP == DROP_BITS) drop_apply_masks(dp, km, a.dd.scale16);
    if (mword != 0u || (a.causal && (kb + KT2 - 1 > q0))) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ko = ACC_ROW(r);
        bool msk = (pad >> ko) & 1u;
        if (a.causal) msk = msk || (kb + ko + 4 * h > q);
        s[r] = msk ? -INFINITY : s[r];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = fast_exp2(fmaf(s[r], a.scale_log2, -L2));
      s[r] = p * (dp[r] - dl);
    }
    const unsigned sb = (unsigned)(uintptr_t)Krh;
    unsigned xa[4], xb[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { xa[d] = sb + (t0 ^ (d << 4)); xb[d] = sb + (t0 ^ (d << 4) ^ 64); }
    const TrQuad a0h = tr_quad_dual<0, 0>(xa, xb), a0l = tr_quad_dual<IMG32, 0>(xa, xb);
    const TrQuad a1h = tr_quad_dual<0, 1>(xa, xb), a1l = tr_quad_dual<IMG32, 1>(xa, xb);
    bf16x8 d0h, d0l, d1h, d1l;
    split8(s, 0, d0h, d0l);
    split8(s, 1, d1h, d1l);
    tr_wait<8>();
    dq[0] = mfma3(tr_join(a0h.lo0, a0h.hi0), tr_join(a0l.lo0, a0l.hi0), d0h, d0l, dq[0]);
    dq[1] = mfma3(tr_join(a0h.lo1, a0h.hi1), tr_join(a0l.lo1, a0l.hi1), d0h, d0l, dq[1]);
    tr_wait<0>();
    dq[0] = mfma3(tr_join(a1h.lo0, a1h.hi0), tr_join(a1l.lo0, a1l.hi0), d1h, d1l, dq[0]);
    dq[1] = mfma3(tr_join(a1h.lo1, a1h.hi1), tr_join(a1l.lo1, a1l.hi1), d1h, d1l, dq[1]);
  }
  if (q < a.Tq) {
    bf16* dqp = dQ + ((int64_t)b * a.Tq + q) * a.lddq + hd * DH + 4 * h;
    const int lod = a.lddq >> 1;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        store4_x2(dqp + 32 * db + 8 * g4, lod, dq[db][4 * g4 + 0] * a.scale, dq[db][4 * g4 + 1] * a.scale,
                  dq[db][4 * g4 + 2] * a.scale, dq[db][4 * g4 + 3] * a.scale);
  }
}

// ------------------------------------------------------------------------------------------ dK, dV
// Workgroup = 4 waves x 32 keys, 32-query tiles.  Stage = Q hi, Q lo, dO hi, dO lo as DUAL-USE images (row reads for
// S = Q K^T / dP = dO V^T, transposed reads for dV^T += dO^T P / dK^T += Q^T dS) + lse / delta: 16.5 KiB, two stages.
// The wave's K / V fragments: hi planes in registers, lo planes parked in LDS (8 KiB per wave, re-read per tile), so
// the kernel holds dK, dV, K hi, V hi (96 registers) and stays clear of scratch at two waves per SIMD.
template <int DROP>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dkv_x3(AttnM a, const bf16* __restrict__ Q,
                                                            const bf16* __restrict__ K,
                                                            const bf16* __restrict__ V,
                                                            const bf16* __restrict__ dO,
                                                            const float* __restrict__ lse,
                                                            const float* __restrict__ delta,
                                                            bf16* __restrict__ dK, bf16* __restrict__ dV) {
  constexpr int QT = 32, NS = 2;
  constexpr int STAGE = 4 * IMG32 + 2 * 64 * 4 + 4 * 256;   // + one 256-byte keep-bit block per wave
  constexpr int KVL = 4 * 2 * 4096;             // parked lo fragments: [wave][K | V][slice][lane] x 16 B
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* ring = lds + KVL;
  const int t = threadIdx.x, lane = t & 63, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tk + 127) / 128);
  const int hd = blk_.hd, b = blk_.b;
  const int k0 = blk_.xb * 128 + w * 32;
  const int key = k0 + (lane & 31);
  const int kc = key < a.Tk ? key : a.Tk - 1;
  const bool kmasked = key >= a.Tk || (a.key_pad && a.key_pad[(int64_t)b * a.Tk + kc]);
  const int loq = a.ldq >> 1, loo = a.ldo >> 1;
  const bf16* Qb = Q + (int64_t)b * a.Tq * a.ldq + hd * DH;
  const bf16* Db = dO + (int64_t)b * a.Tq * a.ldo + hd * DH;
  bf16x8 kh[4], vh[4];
  unsigned char* my_l = lds + w * 8192 + lane * 16;
  {
    const bf16* kp = K + ((int64_t)b * a.Tk + kc) * a.ldk + hd * DH + 8 * h;
    const bf16* vp = V + ((int64_t)b * a.Tk + kc) * a.ldv + hd * DH + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kh[s] = *(const bf16x8*)(kp + 16 * s);
      vh[s] = *(const bf16x8*)(vp + 16 * s);
      *(bf16x8*)(my_l + s * 1024) = *(const bf16x8*)(kp + (a.ldk >> 1) + 16 * s);
      *(bf16x8*)(my_l + 4096 + s * 1024) = *(const bf16x8*)(vp + (a.ldv >> 1) + 16 * s);
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
  const bool wave_all_masked = __all(kmasked);

  int qbeg = 0;
  if (a.causal) qbeg = (blk_.xb * 128) / QT * QT;
  const int ntiles = (a.Tq - qbeg + QT - 1) / QT;
  const int64_t lbase = ((int64_t)b * a.H + hd) * a.Tq;
  __syncthreads();   // plain loads retired / parked fragments visible before the LDS-DMA ring starts
  auto issue = [&](int qt) {   // 4 images x 4 pieces: wave w moves piece w of every image
    unsigned char* st = ring + (qt % NS) * STAGE;
    const int row0 = qbeg + qt * QT;
    dma_piece_dual(st, Qb, a.ldq, row0, a.Tq, w, lane);
    dma_piece_dual(st + IMG32, Qb + loq, a.ldq, row0, a.Tq, w, lane);
    dma_piece_dual(st + 2 * IMG32, Db, a.ldo, row0, a.Tq, w, lane);
    dma_piece_dual(st + 3 * IMG32, Db + loo, a.ldo, row0, a.Tq, w, lane);
    if (w < 2) {
      int qq = row0 + lane;
      qq = qq < a.Tq ? qq : a.Tq - 1;
      const float* src = (w == 0 ? lse : delta) + lbase + qq;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + 4 * IMG32 + w * 64 * 4), 4, 0, 0);
    }
    if (DROP == DROP_BITS) {   // this wave's (query block, key block) of the keep-bit tensor: 32 dwords (upper lanes: duplicates)
      // key block clamped: with Tk < 128 the waves past Tk skip all work but still issue their pieces (read past the tensor)
      const uint32_t* src = (const uint32_t*)bits_block(a, b * a.H + hd, row0 >> 5, min(k0 >> 5, a.nk32 - 1)) + (lane & 31);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + 4 * IMG32 + 2 * 64 * 4 + w * 256), 4, 0, 0);
    }
  };
  const unsigned t0 = tr_dual_t0(lane);
  if (ntiles > 0) issue(0);
  __builtin_assume(ntiles >= 1);
  for (int qt = 0; qt < ntiles; ++qt) {
    const int qb = qbeg + qt * QT;
    attn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (qt + 1 < ntiles) issue(qt + 1);
    const unsigned char* Qh = ring + (qt % NS) * STAGE;
    const unsigned char* Ql = Qh + IMG32;
    const unsigned char* Dh = Qh + 2 * IMG32;
    const unsigned char* Dl = Qh + 3 * IMG32;
    const float* Ls = (const float*)(Qh + 4 * IMG32);
    const float* Ds = Ls + 64;
    const bool ragged = qb + QT > a.Tq;
    if ((a.causal && qb + QT - 1 < k0) || wave_all_masked) continue;
    f32x16 s, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kl = *(const bf16x8*)(my_l + ks * 1024), vl = *(const bf16x8*)(my_l + 4096 + ks * 1024);
      s = mfma3(frag_row_dual(Qh, 0, ks, lane), frag_row_dual(Ql, 0, ks, lane), kh[ks], kl, s);       // S[q][key]
      dp = mfma3(frag_row_dual(Dh, 0, ks, lane), frag_row_dual(Dl, 0, ks, lane), vh[ks], vl, dp);     // dP[q][key]
    }
    if (a.causal || ragged) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = qb + ACC_ROW(r) + 4 * h;
        const bool msk = (a.causal && key > qq) || qq >= a.Tq;
        s[r] = msk ? -INFINITY : s[r];
      }
    }
    // in place: s <- p = exp2(s * scale - lse[q]),  dp <- dS = p (keep dP - delta[q]),  s <- keep p
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 Lq = *(const f32x4*)(Ls + 8 * g4 + 4 * h) * 1.4426950408889634f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s[4 * g4 + j] = fast_exp2(fmaf(s[4 * g4 + j], a.scale_log2, -Lq[j]));
    }
    if (DROP == DROP_BITS) {   // one dword per lane: bit q = keep(query q of the tile, this lane's key)
      const uint32_t word = ((const uint32_t*)(Qh + 4 * IMG32 + 2 * 64 * 4 + w * 256))[bits_word_of_key(lane & 31)] >> (4 * h);
      const int sbits = __builtin_bit_cast(int, a.dd.scale16);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 Dq = *(const f32x4*)(Ds + 8 * g4 + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g4 + j;
          const float kp = __builtin_bit_cast(float, __builtin_amdgcn_sbfe((int)word, ACC_ROW(r), 1) & sbits);   // scale or 0
          dp[r] = (dp[r] * kp - Dq[j]) * s[r];
          s[r] *= kp;
        }
      }
    } else if (DROP == DROP_HASH) {   // keep bits as in k_attn_bwd_dkv_mfma: the lanes of a key pair share one hash (DPP exchange)
      const uint64_t tb = (uint64_t)(lbase + qb + 4 * h + (lane & 1));
      const uint32_t po = afm_pair_offset((uint32_t)key >> 1);
      const uint32_t hshift = (lane & 1) << 4;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 Dq = *(const f32x4*)(Ds + 8 * g4 + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const int r = 4 * g4 + j;
          const uint32_t own = afm_pair_mix(afm_row_hash(a.dd, tb + (uint64_t)ACC_ROW(r)) + po);
          const uint32_t oth = (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);
          const uint32_t h0 = (lane & 1) ? oth : own, h1 = (lane & 1) ? own : oth;
          const float kp0 = ((h0 >> hshift) & 0xFFFFu) >= a.dd.thresh16 ? a.dd.scale16 : 0.f;
          const float kp1 = ((h1 >> hshift) & 0xFFFFu) >= a.dd.thresh16 ? a.dd.scale16 : 0.f;
          dp[r] = (dp[r] * kp0 - Dq[j]) * s[r];
          dp[r + 1] = (dp[r + 1] * kp1 - Dq[j + 1]) * s[r + 1];
          s[r] *= kp0; s[r + 1] *= kp1;
        }
      }
    } else {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 Dq = *(const f32x4*)(Ds + 8 * g4 + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) dp[4 * g4 + j] = (dp[4 * g4 + j] - Dq[j]) * s[4 * g4 + j];
      }
    }
    // transposed-read address registers of this stage: base + (T0 ^ (delta << 4)) [^ 64 for the upper column half]
    const unsigned sb = (unsigned)(uintptr_t)Qh;
    unsigned xa[4], xb[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { xa[d] = sb + (t0 ^ (d << 4)); xb[d] = sb + (t0 ^ (d << 4) ^ 64); }
    {
      const TrQuad dh0 = tr_quad_dual<2 * IMG32, 0>(xa, xb), dl0 = tr_quad_dual<3 * IMG32, 0>(xa, xb);
      const TrQuad qh0 = tr_quad_dual<0, 0>(xa, xb), ql0 = tr_quad_dual<IMG32, 0>(xa, xb);
      bf16x8 ph, pl, sh, sl_;
      split8(s, 0, ph, pl);
      split8(dp, 0, sh, sl_);
      tr_wait<8>();
      dv[0] = mfma3(tr_join(dh0.lo0, dh0.hi0), tr_join(dl0.lo0, dl0.hi0), ph, pl, dv[0]);
      dv[1] = mfma3(tr_join(dh0.lo1, dh0.hi1), tr_join(dl0.lo1, dl0.hi1), ph, pl, dv[1]);
      const TrQuad dh1 = tr_quad_dual<2 * IMG32, 1>(xa, xb), dl1 = tr_quad_dual<3 * IMG32, 1>(xa, xb);
      tr_wait<8>();
      dk[0] = mfma3(tr_join(qh0.lo0, qh0.hi0), tr_join(ql0.lo0, ql0.hi0), sh, sl_, dk[0]);
      dk[1] = mfma3(tr_join(qh0.lo1, qh0.hi1), tr_join(ql0.lo1, ql0.hi1), sh, sl_, dk[1]);
      const TrQuad qh1 = tr_quad_dual<0, 1>(xa, xb), ql1 = tr_quad_dual<IMG32, 1>(xa, xb);
      split8(s, 1, ph, pl);
      split8(dp, 1, sh, sl_);
      tr_wait<8>();
      dv[0] = mfma3(tr_join(dh1.lo0, dh1.hi0), tr_join(dl1.lo0, dl1.hi0), ph, pl, dv[0]);
      dv[1] = mfma3(tr_join(dh1.lo1, dh1.hi1), tr_join(dl1.lo1, dl1.hi1), ph, pl, dv[1]);
      tr_wait<0>();
      dk[0] = mfma3(tr_join(qh1.lo0, qh1.hi0), tr_join(ql1.lo0, ql1.hi0), sh, sl_, dk[0]);
      dk[1] = mfma3(tr_join(qh1.lo1, qh1.hi1), tr_join(ql1.lo1, ql1.hi1), sh, sl_, dk[1]);
    }
  }
  if (kmasked) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
  }
  if (key < a.Tk) {
    bf16* dkp = dK + ((int64_t)b * a.Tk + key) * a.lddk + hd * DH + 4 * h;
    bf16* dvp = dV + ((int64_t)b * a.Tk + key) * a.lddv + hd * DH + 4 * h;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        store4_x2(dkp + 32 * db + 8 * g4, a.lddk >> 1, dk[db][4 * g4 + 0] * a.scale, dk[db][4 * g4 + 1] * a.scale,
                  dk[db][4 * g4 + 2] * a.scale, dk[db][4 * g4 + 3] * a.scale);
        store4_x2(dvp + 32 * db + 8 * g4, a.lddv >> 1, dv[db][4 * g4 + 0], dv[db][4 * g4 + 1], dv[db][4 * g4 + 2], dv[db][4 * g4 + 3]);
      }
  }
}

// ------------------------------------------------------------------------------------------ dispatch
static bool eligible_x3(const afm_attn_shape* s, const void* const* ptrs, int nptr, const int* lds, int nld) {
  if (s->dtype != AFM_BF16X2 || s->dh != DH) return false;
  if (s->sqb || s->skb || s->svb || s->sob) return false;
  if (s->causal && s->Tq != s->Tk) return false;
  if (s->drop.p > 0.f && (s->Tk & 1)) return false;
  if (s->drop.p > 0.f && (uint64_t)s->B * s->H * s->Tq * (uint64_t)s->Tk > 0xFFFFFFFFull) return false;
  for (int i = 0; i < nptr; ++i) if ((uintptr_t)ptrs[i] & 15) return false;
  for (int i = 0; i < nld; ++i) if (lds[i] & 15) return false;    // both planes 16-byte aligned
  return true;
}
static AttnM make_m_x3(const afm_attn_shape* s) {
  AttnM a;
  a.B = s->B; a.H = s->H; a.Tq = s->Tq; a.Tk = s->Tk;
  a.ldq = s->ldq; a.ldk = s->ldk; a.ldv = s->ldv; a.ldo = s->ldo;
  a.lddq = a.lddk = a.lddv = 0;
  a.causal = s->causal; a.scale = s->scale; a.scale_log2 = s->scale * 1.4426950408889634f;
  a.key_pad = s->key_pad; a.dd = afm_make_drop(&s->drop);
  a.bits = a.dd.thresh16 ? (unsigned long long*)s->drop_bits : nullptr;
  a.nq32 = ((s->Tq + 127) / 128) * 4; a.nk32 = ((s->Tk + 63) / 64) * 2;      // whole workgroups / whole 64-key tiles
  return a;
}

int afm_attn_fwd_x3_try(const afm_attn_shape* s, const void* Q, const void* K, const void* V, void* O,
                        float* lse, hipStream_t st) {
  const void* ptrs[] = {Q, K, V, O};
  const int lds[] = {s->ldq, s->ldk, s->ldv, s->ldo};
  if (!eligible_x3(s, ptrs, 4, lds, 4)) return AFM_ERR_UNSUPPORTED;
  const AttnM a = make_m_x3(s);
  const dim3 grid(((s->Tq + 127) / 128) * s->H * s->B);
  const int shm = 2 * 4 * IMG64 + ((s->Tk + KT - 1) / KT) * 8;
  if (shm > 80 * 1024) return AFM_ERR_UNSUPPORTED;
  static AfmOncePerDevice attr;
  if (attr.need()) {
    (void)hipFuncSetAttribute((const void*)k_attn_fwd_x3<DROP_HASH>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)k_attn_fwd_x3<DROP_BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)k_attn_fwd_x3<DROP_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  }
  if (a.dd.thresh16 && a.bits) AFM_LAUNCH(k_attn_fwd_x3<DROP_BITS>, grid, dim3(256), shm, st, a, (const bf16*)Q, (const bf16*)K, (const bf16*)V, (bf16*)O, lse);
  else if (a.dd.thresh16) AFM_LAUNCH(k_attn_fwd_x3<DROP_HASH>, grid, dim3(256), shm, st, a, (const bf16*)Q, (const bf16*)K, (const bf16*)V, (bf16*)O, lse);
  else AFM_LAUNCH(k_attn_fwd_x3<DROP_NONE>, grid, dim3(256), shm, st, a, (const bf16*)Q, (const bf16*)K, (const bf16*)V, (bf16*)O, lse);
  afm_set_last_algo("attn_mfma_x3");
  return AFM_OK;
}

int afm_attn_bwd_x3_try(const afm_attn_shape* s, const void* Q, const void* K, const void* V, const void* O,
                        const void* dO, const float* lse, float* delta, void* dQ, void* dK, void* dV,
                        int lddq, int lddk, int lddv, hipStream_t st) {
  const void* ptrs[] = {Q, K, V, O, dO, dQ, dK, dV};
  const int lds[] = {s->ldq, s->ldk, s->ldv, s->ldo, lddq, lddk, lddv};
  if (!eligible_x3(s, ptrs, 8, lds, 7)) return AFM_ERR_UNSUPPORTED;
  AttnM a = make_m_x3(s);
  a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  const dim3 gq(((s->Tq + 127) / 128) * s->H * s->B), gk(((s->Tk + 127) / 128) * s->H * s->B);
  const int shm_q = 2 * 4 * IMG32 + ((s->Tk + 63) / 64) * 8;
  const int shm_k = 4 * 2 * 4096 + 2 * (4 * IMG32 + 2 * 64 * 4 + 4 * 256);
  if (shm_q > 80 * 1024) return AFM_ERR_UNSUPPORTED;
  static AfmOncePerDevice attr;
  if (attr.need()) {
    (void)hipFuncSetAttribute((const void*)k_attn_bwd_dq_x3<DROP_HASH>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)k_attn_bwd_dq_x3<DROP_BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)k_attn_bwd_dq_x3<DROP_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_x3<DROP_HASH>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_x3<DROP_BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_x3<DROP_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  }
  const bool run_q = s->reserved != 2, run_k = s->reserved != 1;   // reserved = 1 / 2: only the dQ / only the dK-dV kernel (timing)
  if (!run_q) {}
  else if (a.dd.thresh16 && a.bits) AFM_LAUNCH(k_attn_bwd_dq_x3<DROP_BITS>, gq, dim3(256), shm_q, st, a, (const bf16*)Q, (const bf16*)K, (const bf16*)V, (const bf16*)O, (const bf16*)dO, lse, delta, (bf16*)dQ);
  else if (a.dd.thresh16) AFM_LAUNCH(k_attn_bwd_dq_x3<DROP_HASH>, gq, dim3(256), shm_q, st, a, (const bf16*)Q, (const bf16*)K, (const bf16*)V, (const bf16*)O, (const bf16*)dO, lse, delta, (bf16*)dQ);
  else AFM_LAUNCH(k_attn_bwd_dq_x3<DROP_NONE>, gq, dim3(256), shm_q, st, a, (const bf16*)Q, (const bf16*)K, (const bf16*)V, (const bf16*)O, (const bf16*)dO, lse, delta, (bf16*)dQ);
  if (!run_k) {}
  else if (a.dd.thresh16 && a.bits) AFM_LAUNCH(k_attn_bwd_dkv_x3<DROP_BITS>, gk, dim3(256), shm_k, st, a, (const bf16*)Q, (const bf16*)K, (const bf16*)V, (const bf16*)dO, lse, delta, (bf16*)dK, (bf16*)dV);
  else if (a.dd.thresh16) AFM_LAUNCH(k_attn_bwd_dkv_x3<DROP_HASH>, gk, dim3(256), shm_k, st, a, (const bf16*)Q, (const bf16*)K, (const bf16*)V, (const bf16*)dO, lse, delta, (bf16*)dK, (bf16*)dV);
  else AFM_LAUNCH(k_attn_bwd_dkv_x3<DROP_NONE>, gk, dim3(256), shm_k, st, a, (const bf16*)Q, (const bf16*)K, (const bf16*)V, (const bf16*)dO, lse, delta, (bf16*)dK, (bf16*)dV);
  afm_set_last_algo("attn_mfma_x3");
  return AFM_OK;
}
