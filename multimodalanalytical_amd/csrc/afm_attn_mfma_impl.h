// Single-pass 16-bit MFMA flash attention (written against `e16`, compiled as bf16 and as fp16: afm_common.h) for gfx950, head size 64 (d_model 512/8 and 768/12 of the reference's
// model yamls), forward and backward, with key-padding / causal masks and attention-probability
// dropout fused; scores never leave the chip.
//
// All products use v_mfma_f32_32x32x16_bf16 in the "swapped" orientation: the score tile is
// computed TRANSPOSED (S^T = K Q^T), so the softmax axis (keys) runs over a lane's registers and
// the query sits on the lane.  The row maximum / sum are then register loops plus one exchange
// with lane^32, the probability tile (an accumulator) is directly the B operand of the next MFMA
// (O^T = V^T P^T: cdna_hip_programming.md section 3, "an accumulator tile as the next MFMA's
// operand"), and the per-query rescale of O^T is a per-lane multiply.  V^T fragments come from the
// row-major V tile in LDS through ds_read_b64_tr_b16.
//
// forward:   workgroup = 4 waves x 32 queries = 128 queries of one (batch, head); 64-key tiles.
// backward:  two kernels, no atomics, deterministic:
//   dQ   : same geometry as forward; per tile S^T, dP^T = V dO^T, dS^T, dQ^T += K^T dS^T.
//   dK/dV: workgroup = 4 waves x 32 keys; 64-query tiles; S = Q K^T with the KEY on the lane,
//          dV^T += dO^T P, dK^T += Q^T dS.
#include "afm_attn_tiles.h"

namespace AFM_E16_NS {

// ------------------------------------------------------------------------------------------ forward
#ifndef AFM_FWD_OCC
#define AFM_FWD_OCC 4     // waves per SIMD the forward is compiled for: 4 = 128 registers (125 used, no scratch with the keep-bit tensor) and, with
                          // RS = 2, four workgroups' LDS per CU: 0.50 -> 0.48 ms against 3 at RS = 3
#endif
template <int DROP>
__global__ __launch_bounds__(256, AFM_FWD_OCC) void k_attn_fwd_mfma(AttnM a, const e16* __restrict__ Q,
                                                          const e16* __restrict__ K,
                                                          const e16* __restrict__ V, e16* __restrict__ O,
                                                          float* __restrict__ lse) {
  constexpr int STAGE = 2 * KT * DH * 2;   // K row image + V tr image
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned long long* maskw = (unsigned long long*)(lds + RS * STAGE);
  const int t = threadIdx.x, lane = t & 63, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tq + 127) / 128);
  const int hd = blk_.hd, b = blk_.b;
  const int q0 = blk_.xb * 128 + w * 32;          // this wave's first query
  const int q = q0 + (lane & 31);
  const int qc = q < a.Tq ? q : a.Tq - 1;
  // self-attention over a padded batch in a training step: nobody reads the outputs of padded query rows (qskip: afm_attn_shape.reserved
  // bit 6 in the forward sense).  A workgroup whose 128 queries are all padding writes the all-masked-row convention (O = 0, lse = +inf:
  // finite values for whoever still loads the rows, P = 0 for a backward that does not skip them) and leaves; no keep bits are written.
  const int64_t rq = attn_row0(a.q_off, b, a.Tq), rk = attn_row0(a.k_off, b, a.Tk);
  const int lim_q = attn_slot(a.q_off, b, a.Tq);      // rows of this sample that are its own (packed: its slot)
  {
    int64_t tail0;
    if (attn_tail_block(a.q_off, a.B, b, blk_.xb, a.Tq, tail0)) {      // packed rows, a block beyond the sample's slot: zeros to its block of the dead tail
      const int64_t trow = tail0 + w * 32 + (lane & 31);
      if (trow >= attn_fill_end(a.nofill, a.q_off, a.B)) return;
      e16* op = O + trow * a.ldo + hd * DH + 4 * h;
      const e16x4 z = {(e16)0.f, (e16)0.f, (e16)0.f, (e16)0.f};
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) *(e16x4*)(op + 32 * db + 8 * g4) = z;
      return;
    }
  }
  if (a.qskip && __syncthreads_and(q >= a.Tq || a.key_pad[(int64_t)b * a.Tk + qc] != 0)) {
    if (q < lim_q) {
      e16* op = O + (rq + q) * a.ldo + hd * DH + 4 * h;
      const e16x4 z = {(e16)0.f, (e16)0.f, (e16)0.f, (e16)0.f};
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) *(e16x4*)(op + 32 * db + 8 * g4) = z;
      if (h == 0) lse[((int64_t)b * a.H + hd) * a.Tq + q] = INFINITY;
    }
    return;
  }
  const e16* Kb = K + rk * a.ldk + hd * DH;
  const e16* Vb = V + rk * a.ldv + hd * DH;
  int kend = a.Tk;
  if (a.causal) kend = min(a.Tk, blk_.xb * 128 + 128);  // keys beyond the block's last query are masked
  const int ntiles = (kend + KT - 1) / KT;
  // Q fragments (B operand of S^T = K Q^T): Q[q][16 s + 8 h ..]
  e16x8 qf[4];
  {
    const e16* qp = Q + (rq + qc) * a.ldq + hd * DH + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {     // pre-multiplied by scale * log2(e): S^T comes out of the MFMA chain in log2 units (round 3)
      const e16x8 x = ld8_once(qp + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (e16)((float)x[j] * a.scale_log2);
    }
  }
  build_mask_words(maskw, a.key_pad, b, a.Tk, ntiles, w, lane);
  int* const tl = (int*)(maskw + (a.Tk + KT - 1) / KT) + 1;      // key tiles with at least one real key (build_tile_list)
  __syncthreads();
  build_tile_list(tl, a.key_pad ? maskw : nullptr, 0, ntiles, w, lane);
  __syncthreads();   // plain loads above are retired here (vmcnt(0)), before any LDS-DMA is in flight
  const int nlive = __builtin_amdgcn_readfirstlane(tl[-1]);
  auto issue = [&](int j) {
    unsigned char* st = lds + (j % RS) * STAGE;
    const int kt = tl[j];
    dma_piece<false>(st, Kb, a.ldk, kt * KT, a.Tk, w, lane);
    dma_piece<false>(st, Kb, a.ldk, kt * KT, a.Tk, w + 4, lane);
    dma_piece<true>(st + KT * DH * 2, Vb, a.ldv, kt * KT, a.Tk, w, lane);
    dma_piece<true>(st + KT * DH * 2, Vb, a.ldv, kt * KT, a.Tk, w + 4, lane);
  };
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m = -INFINITY, l = 0.f;   // running maximum (log2 units; -inf until the row has seen an unmasked key) and row sum
  const uint32_t rowbase = afm_row_hash(a.dd, (uint64_t)(b * a.H + hd) * a.Tq + qc);   // the lane's ROW HASH (two-level dropout stream, afm_common.h)
#pragma unroll
  for (int s = 0; s < RS - 1; ++s)
    if (s < nlive) issue(s);
  // nlive >= 1 always (build_tile_list): without the guard the loop's exit block has one predecessor and the
  // accumulators need no phi copies at the latch (they cost 32-64 v_mov per tile)
  __builtin_assume(nlive >= 1);
  for (int j = 0; j < nlive; ++j) {
    const int kt = __builtin_amdgcn_readfirstlane(tl[j]);
    const int kb = kt * KT;
    if (nlive - 1 - j >= RS - 2) attn_wait_vmcnt<4 * (RS - 2)>(); else attn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (j + RS - 1 < nlive) issue(j + RS - 1);
    // wave-uniform skips: tile entirely above this wave's diagonal / every key of the tile is padding (the list's fallback tile)
    if ((a.causal && kb > q0 + 31) || maskw[kt] == ~0ull) continue;
    const unsigned char* Kimg = lds + (j % RS) * STAGE;
    const unsigned char* Vimg = Kimg + KT * DH * 2;
    const unsigned long long mword = maskw[kt];
    const unsigned long long pad = mword >> (4 * h);
    KeepMasks km[2];
    if (DROP == DROP_READ) {   // the tensor was filled beforehand (afm_attn_drop_bits_fill): 2 x 16 lane masks as SGPR pairs
      const unsigned long long* bb = bits_block(a, b * a.H + hd, q0 >> 5, kb >> 5);
      keep_masks_issue(km[0], bb);
      keep_masks_issue(km[1], bb + 16);
    }
    f32x16 s[2];
    const float init = m == -INFINITY ? 0.f : -m;      // S' = S - m: the running maximum is the chain's initial value
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[blk][i] = init;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[blk] = mfma32(frag_row(Kimg, 32 * blk, ks, lane), qf[ks], s[blk]);
    }
    // masks only where the tile has any (wave-uniform test): most tiles of a padded batch have none
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
    // row maximum with v_max3_f32 from inline asm: fmaxf() makes hipcc canonicalise every MFMA output
    // first (one extra `v_max_f32 x, x, x` per score); the scores are finite or -inf, never NaN
    float mt = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = max3_raw(mt, s[0][r], s[1][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));        // maximum of S' over the tile's keys
    // a row's first unmasked key sets m to the true maximum; afterwards m moves only when the row grew by more than 2^8
    // (P <= 2^8 is exact in either 16-bit format and the fp32 sums do not care): O and l are rescaled in that rare tile only
    const bool unset = m == -INFINITY;
    if (__any((unset && mt != -INFINITY) || mt > 8.f)) {
      const float dlt = unset ? (mt == -INFINITY ? 0.f : mt) : fmaxf(mt, 0.f);
      const float alpha = unset ? 1.f : fast_exp2(-dlt);     // (an unset row has accumulated nothing yet)
      m = (unset && mt == -INFINITY) ? m : (unset ? dlt : m + dlt);
      l *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; s[0][i] -= dlt; s[1][i] -= dlt; }
    }
    float ls = 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(s[blk][r]);   // exp2(-inf) = 0 for masked keys
        s[blk][r] = p;
        ls += p;
      }
    l += ls;
    if (DROP == DROP_HASH) {   // compile-time: a run-time branch costs 32 register copies at its join
      drop_block(a.dd, rowbase, kb, h, s[0]);
      drop_block(a.dd, rowbase, kb + 32, h, s[1]);
    }
    if (DROP == DROP_BITS) {   // the same dropout, and the keep bits of both 32-key blocks go to the keep-bit tensor
      unsigned long long* bb = bits_block(a, b * a.H + hd, q0 >> 5, kb >> 5);
      drop_block_emit(a.dd, rowbase, kb, h, s[0], bb);
      drop_block_emit(a.dd, rowbase, kb + 32, h, s[1], bb + 16);
    }
    if (DROP == DROP_READ) {   // one select per score
      drop_select_masks(s[0], km[0], 0.f);
      drop_select_masks(s[1], km[1], 0.f);
    }
    // O^T += V^T P^T over the four 16-key slices, the V^T reads one slice ahead of the MFMAs
    unsigned va0, va1;
    tr_lane_addr(Vimg, lane, va0, va1);
    TrQuad vq[2];
    vq[0] = tr_issue(va0, va1, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // slice i = 2 blk + ks
      if (i < 3) vq[(i + 1) & 1] = tr_issue(va0, va1, 16 * (i + 1));
      // packed conversions (half an instruction per score) where they fit the 128 registers: with dropout since the two-level hash
      const e16x8 pf = DROP == DROP_NONE ? cvt8(s[i >> 1], i & 1) : cvt8_pk(s[i >> 1], i & 1);
      if (i < 3) tr_wait<4>(); else tr_wait<0>();
      o[0] = mfma32(tr_join(vq[i & 1].lo0, vq[i & 1].hi0), pf, o[0]);
      o[1] = mfma32(tr_join(vq[i & 1].lo1, vq[i & 1].hi1), pf, o[1]);
    }
  }
  if (DROP == DROP_BITS) bits_flush();
  l += __shfl_xor(l, 32, 64);
  const float inv = l > 0.f ? a.dd.scale16 / l : 0.f;
  if (q < a.Tq) {
    // packed rows: a wave whose rows lie beyond the slot (a partly used last block) zeroes its rows of the dead tail instead
    const bool own = q < lim_q;
    const int64_t orow = own ? rq + q : attn_tail0(a.q_off, a.B, b, a.Tq) + (q - lim_q);
    if (own || orow < attn_fill_end(a.nofill, a.q_off, a.B)) {
      e16* op = O + orow * a.ldo + hd * DH + 4 * h;
      const float sc = own ? inv : 0.f;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          e16x4 v = {(e16)(o[db][4 * g4 + 0] * sc), (e16)(o[db][4 * g4 + 1] * sc),
                      (e16)(o[db][4 * g4 + 2] * sc), (e16)(o[db][4 * g4 + 3] * sc)};
          if (!own) v = (e16x4){(e16)0.f, (e16)0.f, (e16)0.f, (e16)0.f};      // (o may hold anything for rows of another sample)
          *(e16x4*)(op + 32 * db + 8 * g4) = v;
        }
    }
    // (rows beyond a packed slot are another sample's: +inf, the all-masked-row convention, makes them P = 0 for the dK/dV kernel, whose
    // 64-query tiles may reach past a slot that ends on a 32-row boundary)
    if (h == 0) lse[((int64_t)b * a.H + hd) * a.Tq + q] = (l > 0.f && own) ? (m + __log2f(l)) * 0.69314718055994531f : INFINITY;
  }
}

// ------------------------------------------------------------------------------------------ forward, 8 staggered waves
// Round-3 form for long query sequences (Tq >= 256).  Measurements behind it (tools/experiments/pingpong.hip):
// ONE wave issues at most one vector instruction per ~5 cycles, whereas the SIMD's vector pipe takes two to three waves' worth;
// the matrix pipe is shared by the SIMD's waves.  Two waves that run the same tile loop half a tile apart -- one in its MFMA
// segment (S = K Q^T, O += V^T P) while the other is in its softmax -- overlap almost completely, two waves in lockstep do not.
//   workgroup = 8 waves x 32 queries = 256 queries of one (batch, head); waves w and w + 4 share a SIMD
//   waves 4-7 run HALF A TILE LATE: the tile loop has ONE barrier per tile, which the early waves meet at the top of tile t and
//   the late waves between the softmax and the P V product of tile t, so the offset is kept for the whole kernel
//   K / V tiles (64 keys) arrive by LDS-DMA in a 3-stage ring shared by all 8 waves (half the L2 -> LDS traffic per query)
// Vector work per score is cut as well: Q is multiplied by scale * log2(e) once per kernel, and the running maximum enters
// the S^T accumulators as their initial value (S' = S - m), so a probability is one v_exp_f32; O and l are rescaled only when a
// row maximum grew by more than 2^8 (wave-uniform branch; P <= 2^8 is exact in either 16-bit format and fp32 sums do not care).
template <int DROP>
__global__ __launch_bounds__(512, 2) void k_attn_fwd_st(AttnM a, const e16* __restrict__ Q,
                                                        const e16* __restrict__ K,
                                                        const e16* __restrict__ V, e16* __restrict__ O,
                                                        float* __restrict__ lse) {
  constexpr int STAGE = 2 * KT * DH * 2;   // K row image + V tr image
  constexpr int NST = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned long long* maskw = (unsigned long long*)(lds + NST * STAGE);
  const int t = threadIdx.x, lane = t & 63, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool late = w >= 4;
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tq + 255) / 256);
  const int hd = blk_.hd, b = blk_.b;
  const int q0 = blk_.xb * 256 + w * 32;
  const int q = q0 + (lane & 31);
  const int qc = q < a.Tq ? q : a.Tq - 1;
  const bool wave_on = q0 < a.Tq;           // waves past the last query only keep the ring and the barriers going
  const e16* Kb = K + (int64_t)b * a.Tk * a.ldk + hd * DH;
  const e16* Vb = V + (int64_t)b * a.Tk * a.ldv + hd * DH;
  int kend = a.Tk;
  if (a.causal) kend = min(a.Tk, blk_.xb * 256 + 256);
  const int ntiles = (kend + KT - 1) / KT;
  // Q fragments, pre-multiplied by scale * log2(e): S comes out of the MFMA chain in log2 units
  e16x8 qf[4];
  {
    const e16* qp = Q + ((int64_t)b * a.Tq + qc) * a.ldq + hd * DH + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const e16x8 x = ld8_once(qp + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (e16)((float)x[j] * a.scale_log2);
    }
  }
  build_mask_words(maskw, a.key_pad, b, a.Tk, ntiles, w & 3, lane);   // (both halves write the same words)
  __syncthreads();   // plain loads retired before any LDS-DMA is in flight
  auto issue = [&](int kt) {      // 16 pieces per tile: wave w moves piece w of the K image and of the V image
    unsigned char* st = lds + (kt % NST) * STAGE;
    dma_piece<false>(st, Kb, a.ldk, kt * KT, a.Tk, w, lane);
    dma_piece<true>(st + KT * DH * 2, Vb, a.ldv, kt * KT, a.Tk, w, lane);
  };
  f32x16 o[2], s[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m = -INFINITY, l = 0.f;   // running maximum (log2 units; -inf until the row has seen an unmasked key) and row sum
  e16x8 pf[4];                    // the tile's probabilities as MFMA operands (written by the vector segment)
  const uint32_t rowbase = afm_row_hash(a.dd, (uint64_t)(b * a.H + hd) * a.Tq + qc);   // the lane's ROW HASH (two-level dropout stream, afm_common.h)
  issue(0);
  if (ntiles > 1) issue(1);
  attn_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // one barrier per tile.  At barrier kt the early waves are about to run { O += V^T P (kt - 1) ; S (kt) } -- their MFMA segment --
  // and the late waves softmax (kt), then their MFMA segment { O += V^T P (kt) ; S (kt + 1) }: tile kt + 1 has landed for
  // everybody, tiles kt - 1 .. kt + 1 stay, stage (kt + 2) % 4 (tile kt - 2) is free for the next DMA
  auto sync = [&](int kt) {
#if !defined(AFM_ABL) || (AFM_ABL != 2 && AFM_ABL != 4)
    attn_wait_vmcnt<0>();
#endif
#if !defined(AFM_ABL) || (AFM_ABL != 1 && AFM_ABL != 4)
    __builtin_amdgcn_s_barrier();
#endif
#if !defined(AFM_ABL) || (AFM_ABL != 2 && AFM_ABL != 4)
    if (kt + 2 < ntiles) issue(kt + 2);
#endif
  };
  auto live = [&](int kt) {   // wave-uniform: the tile has unmasked keys for this wave
    return wave_on && !(a.causal && kt * KT > q0 + 31) && maskw[kt] != ~0ull;
  };
  auto scores = [&](int kt) {
    const unsigned char* Kimg = lds + (kt % NST) * STAGE;
    const float init = m == -INFINITY ? 0.f : -m;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[blk][i] = init;
#pragma unroll
#if defined(AFM_ABL) && AFM_ABL >= 3
      for (int ks = 0; ks < 4; ++ks) s[blk] = mfma32(qf[(ks + blk) & 3], qf[ks], s[blk]);
#else
      for (int ks = 0; ks < 4; ++ks) s[blk] = mfma32(frag_row(Kimg, 32 * blk, ks, lane), qf[ks], s[blk]);
      asm volatile("" ::: "memory");    // keep the second block's four fragment reads behind the first block's (16 fewer live registers)
#endif
    }
  };
  auto softmax = [&](int kt) {
    const int kb = kt * KT;
    const unsigned long long mword = maskw[kt];
    const bool diag = a.causal && (kb + KT - 1 > q0);
    if (mword != 0ull || diag) {
      const unsigned long long pad = mword >> (4 * h);
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
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));        // maximum of S' = S - m over the tile's keys
    // a row's first unmasked key sets m to the true maximum; afterwards m moves only when the row grew by more than 2^8
    const bool unset = m == -INFINITY;
    if (__any((unset && mt != -INFINITY) || mt > 8.f)) {
      const float dlt = unset ? (mt == -INFINITY ? 0.f : mt) : fmaxf(mt, 0.f);
      const float alpha = unset ? 1.f : fast_exp2(-dlt);     // (an unset row has accumulated nothing yet)
      m = (unset && mt == -INFINITY) ? m : (unset ? dlt : m + dlt);
      l *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; s[0][i] -= dlt; s[1][i] -= dlt; }
    }
    float ls = 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(s[blk][r]);       // exp2(-inf) = 0 for masked keys
        s[blk][r] = p;
        ls += p;
      }
    l += ls;
    if (DROP == DROP_HASH) {
      drop_block(a.dd, rowbase, kb, h, s[0]);
      drop_block(a.dd, rowbase, kb + 32, h, s[1]);
    }
    if (DROP == DROP_BITS) {
      unsigned long long* bb = bits_block(a, b * a.H + hd, q0 >> 5, kb >> 5);
      drop_block_emit(a.dd, rowbase, kb, h, s[0], bb);
      drop_block_emit(a.dd, rowbase, kb + 32, h, s[1], bb + 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = cvt8(s[i >> 1], i & 1);
  };
  auto pv = [&](int kt) {
    const unsigned char* Vimg = lds + (kt % NST) * STAGE + KT * DH * 2;
#if defined(AFM_ABL) && AFM_ABL >= 3
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[0] = mfma32(qf[i], pf[i], o[0]); o[1] = mfma32(qf[(i + 1) & 3], pf[i], o[1]); }
#else
    unsigned va0, va1;
    tr_lane_addr(Vimg, lane, va0, va1);
    TrQuad vq[2];
    vq[0] = tr_issue(va0, va1, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < 3) vq[(i + 1) & 1] = tr_issue(va0, va1, 16 * (i + 1));
      if (i < 3) tr_wait<4>(); else tr_wait<0>();
      o[0] = mfma32(tr_join(vq[i & 1].lo0, vq[i & 1].hi0), pf[i], o[0]);
      o[1] = mfma32(tr_join(vq[i & 1].lo1, vq[i & 1].hi1), pf[i], o[1]);
    }
#endif
  };
  __builtin_assume(ntiles >= 1);
  // Each wave alternates an MFMA segment { P V of the previous tile ; S of the next } with a vector segment (softmax); the
  // barrier sits at the START of the early waves' MFMA segment and at the START of the late waves' vector segment, so the
  // two waves of a SIMD are held in opposite segments (two MFMA segments side by side would serialise on the matrix pipe).
  bool pl = false;      // the previous tile was live: its P V product is pending
  if (!late) {
    for (int kt = 0; kt < ntiles; ++kt) {
      sync(kt);
      if (pl) pv(kt - 1);
      pl = live(kt);
      if (pl) { scores(kt); softmax(kt); }
    }
    if (pl) pv(ntiles - 1);
  } else {
    __builtin_amdgcn_s_setprio(1);      // the younger half loses issue arbitration otherwise (MI355X_MICROARCH.md, two waves per SIMD)
    bool lv = live(0);
    if (lv) scores(0);
    for (int kt = 0; kt < ntiles; ++kt) {
      sync(kt);
      if (lv) { softmax(kt); pv(kt); }
      lv = kt + 1 < ntiles && live(kt + 1);
      if (lv) scores(kt + 1);
    }
  }
  if (DROP == DROP_BITS) bits_flush();
  l += __shfl_xor(l, 32, 64);
  const float inv = l > 0.f ? a.dd.scale16 / l : 0.f;
  if (q < a.Tq) {
    e16* op = O + ((int64_t)b * a.Tq + q) * a.ldo + hd * DH + 4 * h;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        e16x4 v = {(e16)(o[db][4 * g4 + 0] * inv), (e16)(o[db][4 * g4 + 1] * inv),
                    (e16)(o[db][4 * g4 + 2] * inv), (e16)(o[db][4 * g4 + 3] * inv)};
        *(e16x4*)(op + 32 * db + 8 * g4) = v;
      }
    if (h == 0) lse[((int64_t)b * a.H + hd) * a.Tq + q] = l > 0.f ? (m + __log2f(l)) * 0.69314718055994531f : INFINITY;
  }
}

// ------------------------------------------------------------------------------------------ dQ
// Per 64-key tile: S^T and dP^T (keys in registers, query on the lane), then
// dQ^T[d][q] += sum_key K^T[d][key] dS^T[key][q].  Also writes delta = rowsum(dO * O).
#ifndef AFM_DQ_OCC
#define AFM_DQ_OCC 3      // workgroups per CU the dQ kernel is compiled for (168 registers, 8 ... 40 bytes of spill).  2 (no spill) was measured in
                          // round 5 (tools/experiments/r5_dqocc.sh): keep-bit path 0.557 -> 0.611 ms, re-hash 0.706 -> 0.640, step 3 100 -> 3 058: stays 3
#endif
template <int DROP>
__global__ __launch_bounds__(256, AFM_DQ_OCC) void k_attn_bwd_dq_mfma(AttnM a, const e16* __restrict__ Q,
                                                          const e16* __restrict__ K,
                                                          const e16* __restrict__ V,
                                                          const e16* __restrict__ O,
                                                          const e16* __restrict__ dO,
                                                          const float* __restrict__ lse,
                                                          float* __restrict__ delta, e16* __restrict__ dQ) {
  constexpr int STAGE = 3 * KT * DH * 2;   // K row image, K tr image, V row image
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned long long* maskw = (unsigned long long*)(lds + RS * STAGE);
  const int t = threadIdx.x, lane = t & 63, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tq + 127) / 128);
  const int hd = blk_.hd, b = blk_.b;
  const int q0 = blk_.xb * 128 + w * 32;
  const int q = q0 + (lane & 31);
  const int qc = q < a.Tq ? q : a.Tq - 1;
  // (rk through readfirstlane: the K / V tile bases are then scalar for the compiler and the ring's pieces take the saddr form, dma_piece_s)
  const int64_t rq = attn_row0(a.q_off, b, a.Tq), rk = (int64_t)__builtin_amdgcn_readfirstlane((int)attn_row0(a.k_off, b, a.Tk));
  const int lim_q = attn_slot(a.q_off, b, a.Tq);      // rows of this sample that are its own (packed: its slot)
  {
    int64_t tail0;
    if (attn_tail_block(a.q_off, a.B, b, blk_.xb, a.Tq, tail0)) {      // packed rows, a block beyond the sample's slot: zeros to its block of the dead tail
      const int64_t trow = tail0 + w * 32 + (lane & 31);
      if (trow >= attn_fill_end(a.nofill, a.q_off, a.B)) return;
      e16* dqp = dQ + trow * a.lddq + hd * DH + 4 * h;
      const e16x4 z = {(e16)0.f, (e16)0.f, (e16)0.f, (e16)0.f};
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) *(e16x4*)(dqp + 32 * db + 8 * g4) = z;
      return;
    }
  }
  const e16* Kb = K + rk * a.ldk + hd * DH;
  const e16* Vb = V + rk * a.ldv + hd * DH;
  e16x8 qf[4], dof[4];
  float dl = 0.f;
  {
    const e16* qp = Q + (rq + qc) * a.ldq + hd * DH + 8 * h;
    const e16* dop = dO + (rq + qc) * a.ldo + hd * DH + 8 * h;
    const e16* op = O + (rq + qc) * a.ldo + hd * DH + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = ld8_once(qp + 16 * s);
      dof[s] = ld8_once(dop + 16 * s);
      const e16x8 ov = ld8_once(op + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += (float)dof[s][j] * (float)ov[j];
    }
  }
  dl += __shfl_xor(dl, 32, 64);
  const int64_t lrow = ((int64_t)b * a.H + hd) * a.Tq + qc;
  if (q < a.Tq && h == 0) delta[lrow] = q < lim_q ? -dl : 0.f;      // the workspace holds -delta: the dK/dV kernel starts its dP accumulators from it as is (rows beyond a packed slot: 0)
  const float L = lse[lrow];
  // Row constants as the initial accumulators (round 3): Q is pre-multiplied by scale * log2(e) and S^T starts at -lse (log2
  // units), so p = exp2(S') is one instruction; dO is pre-multiplied by the dropout scale and dP^T starts at -delta, so
  // dS = p * (keep ? acc : -delta) is a select and a multiply.
  const float nL2 = L == INFINITY ? -INFINITY : -L * 1.4426950408889634f;
  const float ndl = -dl;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      qf[s][j] = (e16)((float)qf[s][j] * a.scale_log2);
      if (DROP != DROP_NONE) dof[s][j] = (e16)((float)dof[s][j] * a.dd.scale16);
    }
  f32x16 dq[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }
  const uint32_t rowbase = afm_row_hash(a.dd, (uint64_t)lrow);   // the lane's ROW HASH (two-level dropout stream, afm_common.h)

  int kend = a.Tk;
  if (a.causal) kend = min(a.Tk, blk_.xb * 128 + 128);
  const int ntiles = (kend + KT - 1) / KT;
  build_mask_words(maskw, a.key_pad, b, a.Tk, ntiles, w, lane);
  // self-attention over a padded batch: the caller vouches (qskip) that padded query rows carry zero dO -- their dQ rows are exact
  // zeros whatever the keys, so a wave whose 32 queries are all padding only keeps the ring and the barriers going
  const bool wave_qskip = a.qskip && __all(q >= a.Tq || a.key_pad[(int64_t)b * a.Tk + qc] != 0);
  int* const tl = (int*)(maskw + (a.Tk + KT - 1) / KT) + 1;      // key tiles with at least one real key
  if (__syncthreads_and(wave_qskip)) {   // all 128 queries of the workgroup are padding: their dQ rows are zeros, nothing to load
    const int64_t zrow = attn_out_row(a.q_off, a.B, b, a.Tq, rq, q, lim_q);      // (rows beyond a packed slot: their share of the dead tail)
    if (zrow >= 0 && (q < lim_q || zrow < attn_fill_end(a.nofill, a.q_off, a.B))) {
      e16* dqp = dQ + zrow * a.lddq + hd * DH + 4 * h;
      const e16x4 z = {(e16)0.f, (e16)0.f, (e16)0.f, (e16)0.f};
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) *(e16x4*)(dqp + 32 * db + 8 * g4) = z;
    }
    return;
  }
  build_tile_list(tl, a.key_pad ? maskw : nullptr, 0, ntiles, w, lane);
  __syncthreads();   // retires the plain loads / the delta store before the LDS-DMA ring starts
  const int nlive = __builtin_amdgcn_readfirstlane(tl[-1]);
  auto issue = [&](int j) {
    unsigned char* st = lds + (j % RS) * STAGE;
    const int kt = tl[j];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#ifdef AFM_DQ_OLD_DMA      // (A / B builds: the 64-bit lane addresses of rounds 2-5)
      dma_piece<false>(st, Kb, a.ldk, kt * KT, a.Tk, w + 4 * u, lane);
      dma_piece<true>(st + KT * DH * 2, Kb, a.ldk, kt * KT, a.Tk, w + 4 * u, lane);
      dma_piece<false>(st + 2 * KT * DH * 2, Vb, a.ldv, kt * KT, a.Tk, w + 4 * u, lane);
#else
      dma_piece_s<0>(st, Kb, a.ldk, kt * KT, a.Tk, w + 4 * u, lane);
      dma_piece_s<1>(st + KT * DH * 2, Kb, a.ldk, kt * KT, a.Tk, w + 4 * u, lane);
      dma_piece_s<0>(st + 2 * KT * DH * 2, Vb, a.ldv, kt * KT, a.Tk, w + 4 * u, lane);
#endif
    }
  };
#pragma unroll
  for (int s = 0; s < RS - 1; ++s)
    if (s < nlive) issue(s);
  // nlive >= 1 always (build_tile_list): without the guard the loop's exit block has one predecessor and the
  // accumulators need no phi copies at the latch (they cost 32-64 v_mov per tile)
  __builtin_assume(nlive >= 1);
  for (int j = 0; j < nlive; ++j) {
    const int kt = __builtin_amdgcn_readfirstlane(tl[j]);
    const int kb = kt * KT;
    if (nlive - 1 - j >= RS - 2) attn_wait_vmcnt<6 * (RS - 2)>(); else attn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (j + RS - 1 < nlive) issue(j + RS - 1);
    if ((a.causal && kb > q0 + 31) || maskw[kt] == ~0ull || wave_qskip) continue;   // above the diagonal / all-padding key tile / all-padding queries
    const unsigned char* Krow = lds + (j % RS) * STAGE;
    const unsigned char* Ktr = Krow + KT * DH * 2;
    const unsigned char* Vrow = Krow + 2 * KT * DH * 2;
    const unsigned long long mword = maskw[kt];
    const unsigned long long pad = mword >> (4 * h);
    unsigned ka0, ka1;
    tr_lane_addr(Ktr, lane, ka0, ka1);
    KeepMasks km[2];
    if (DROP == DROP_BITS) {   // both 32-key blocks of the tile now; used after the S / dP products
      const unsigned long long* kbp = bits_block(a, b * a.H + hd, q0 >> 5, 2 * kt);
      keep_masks_issue(km[0], kbp);
      keep_masks_issue(km[1], kbp + 16);
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = nL2; dp[i] = ndl; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = mfma32(frag_row(Krow, 32 * blk, ks, lane), qf[ks], s);
        dp = mfma32(frag_row(Vrow, 32 * blk, ks, lane), dof[ks], dp);
      }
      if (DROP == DROP_HASH) drop_block_select(a.dd, rowbase, kb + 32 * blk, h, dp, ndl);
      if (DROP == DROP_BITS) drop_select_masks(dp, km[blk], ndl);
      if (mword != 0ull || (a.causal && (kb + KT - 1 > q0))) {   // wave-uniform: tile has masked keys
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ko = 32 * blk + ACC_ROW(r);
          bool msk = (pad >> ko) & 1ull;
          if (a.causal) msk = msk || (kb + ko + 4 * h > q);
          s[r] = msk ? -INFINITY : s[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(s[r]);   // masked: exp2(-inf) = 0
        s[r] = p * dp[r];   // dS^T = P (D dP - delta) (the 1/sqrt(dh) factor is applied once at the end)
      }
      {
        const TrQuad k0q = tr_issue(ka0, ka1, 32 * blk), k1q = tr_issue(ka0, ka1, 32 * blk + 16);
        const e16x8 ds0 = cvt8(s, 0), ds1 = cvt8(s, 1);
        tr_wait<4>();
        dq[0] = mfma32(tr_join(k0q.lo0, k0q.hi0), ds0, dq[0]);
        dq[1] = mfma32(tr_join(k0q.lo1, k0q.hi1), ds0, dq[1]);
        tr_wait<0>();
        dq[0] = mfma32(tr_join(k1q.lo0, k1q.hi0), ds1, dq[0]);
        dq[1] = mfma32(tr_join(k1q.lo1, k1q.hi1), ds1, dq[1]);
      }
    }
  }
  // packed rows: a wave whose rows lie beyond the slot (a partly used last block) zeroes its rows of the dead tail instead
  const bool own = q < lim_q;
  const int64_t drow = attn_out_row(a.q_off, a.B, b, a.Tq, rq, q, lim_q);
  if (drow >= 0 && (own || drow < attn_fill_end(a.nofill, a.q_off, a.B))) {
    e16* dqp = dQ + drow * a.lddq + hd * DH + 4 * h;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        e16x4 v = {(e16)(dq[db][4 * g4 + 0] * a.scale), (e16)(dq[db][4 * g4 + 1] * a.scale),
                    (e16)(dq[db][4 * g4 + 2] * a.scale), (e16)(dq[db][4 * g4 + 3] * a.scale)};
        if (!own) v = (e16x4){(e16)0.f, (e16)0.f, (e16)0.f, (e16)0.f};
        *(e16x4*)(dqp + 32 * db + 8 * g4) = v;
      }
  }
}

// ------------------------------------------------------------------------------------------ dQ, 8 staggered waves
// The forward's round-3 structure (k_attn_fwd_st) for the dQ kernel: 8 waves x 32 queries, waves 4-7 half a tile late, one
// barrier per 64-key tile.  A wave alternates an MFMA segment { dQ^T += K^T dS^T of the previous tile (8 MFMAs) ; S^T and dP^T of
// the next (16) } with a vector segment { p = exp2(S'), dropout select, dS = p * dP' , conversion to MFMA operands }: 768 matrix
// cycles against ~1000 vector-issue cycles, so the two waves of a SIMD keep both pipes busy.  4-stage ring of K row / K tr / V row
// images shared by the 8 waves.
template <int DROP>
__global__ __launch_bounds__(512, 2) void k_attn_bwd_dq_st(AttnM a, const e16* __restrict__ Q,
                                                           const e16* __restrict__ K,
                                                           const e16* __restrict__ V,
                                                           const e16* __restrict__ O,
                                                           const e16* __restrict__ dO,
                                                           const float* __restrict__ lse,
                                                           float* __restrict__ delta, e16* __restrict__ dQ) {
  constexpr int STAGE = 3 * KT * DH * 2;   // K row image, K tr image, V row image
  constexpr int NST = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned long long* maskw = (unsigned long long*)(lds + NST * STAGE);
  const int t = threadIdx.x, lane = t & 63, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool late = w >= 4;    // waves w and w + 4 share a SIMD (measured: pairing by parity or by w & 2 is 15 % slower)
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tq + 255) / 256);
  const int hd = blk_.hd, b = blk_.b;
  const int q0 = blk_.xb * 256 + w * 32;
  const int q = q0 + (lane & 31);
  const int qc = q < a.Tq ? q : a.Tq - 1;
  const bool wave_on = q0 < a.Tq;
  const e16* Kb = K + (int64_t)b * a.Tk * a.ldk + hd * DH;
  const e16* Vb = V + (int64_t)b * a.Tk * a.ldv + hd * DH;
  e16x8 qf[4], dof[4];
  float dl = 0.f;
  {
    const e16* qp = Q + ((int64_t)b * a.Tq + qc) * a.ldq + hd * DH + 8 * h;
    const e16* dop = dO + ((int64_t)b * a.Tq + qc) * a.ldo + hd * DH + 8 * h;
    const e16* op = O + ((int64_t)b * a.Tq + qc) * a.ldo + hd * DH + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = ld8_once(qp + 16 * s);
      dof[s] = ld8_once(dop + 16 * s);
      const e16x8 ov = ld8_once(op + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += (float)dof[s][j] * (float)ov[j];
    }
  }
  dl += __shfl_xor(dl, 32, 64);
  const int64_t lrow = ((int64_t)b * a.H + hd) * a.Tq + qc;
  if (q < a.Tq && h == 0) delta[lrow] = -dl;      // the workspace holds -delta: the dK/dV kernel starts its dP accumulators from it as is
  const float L = lse[lrow];
  const float nL2 = L == INFINITY ? -INFINITY : -L * 1.4426950408889634f;    // row constants as initial accumulators
  const float ndl = -dl;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      qf[s][j] = (e16)((float)qf[s][j] * a.scale_log2);
      if (DROP != DROP_NONE) dof[s][j] = (e16)((float)dof[s][j] * a.dd.scale16);
    }
  f32x16 dq[2], sc[2], dp[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }
  e16x8 ds[4];                       // dS^T of the tile as MFMA operands (written by the vector segment)
  const uint32_t rowbase = afm_row_hash(a.dd, (uint64_t)lrow);   // the lane's ROW HASH (two-level dropout stream, afm_common.h)
  int kend = a.Tk;
  if (a.causal) kend = min(a.Tk, blk_.xb * 256 + 256);
  const int ntiles = (kend + KT - 1) / KT;
  build_mask_words(maskw, a.key_pad, b, a.Tk, ntiles, w & 3, lane);
  __syncthreads();   // retires the plain loads / the delta store before the LDS-DMA ring starts
  auto issue = [&](int kt) {      // 24 pieces per tile: wave w moves piece w of each of the three images
    unsigned char* st = lds + (kt % NST) * STAGE;
    dma_piece<false>(st, Kb, a.ldk, kt * KT, a.Tk, w, lane);
    dma_piece<true>(st + KT * DH * 2, Kb, a.ldk, kt * KT, a.Tk, w, lane);
    dma_piece<false>(st + 2 * KT * DH * 2, Vb, a.ldv, kt * KT, a.Tk, w, lane);
  };
  issue(0);
  if (ntiles > 1) issue(1);
  attn_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  auto sync = [&](int kt) {         // see k_attn_fwd_st: tile kt + 1 landed for everybody, stage (kt + 2) % 4 is free
#if defined(AFM_ABL) && AFM_ABL == 8      // timing only: no barrier, no DMA
    return;
#endif
    attn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < ntiles) issue(kt + 2);
  };
  auto live = [&](int kt) { return wave_on && !(a.causal && kt * KT > q0 + 31) && maskw[kt] != ~0ull; };
  KeepMasks km[2];
  auto products = [&](int kt) {     // MFMA segment, second half: S^T = K Q^T - lse, dP'^T = scale V dO^T - delta
    const unsigned char* Krow = lds + (kt % NST) * STAGE;
    const unsigned char* Vrow = Krow + 2 * KT * DH * 2;
    if (DROP == DROP_BITS) {        // the tile's keep masks: scalar loads in flight during the MFMAs
      const unsigned long long* kbp = bits_block(a, b * a.H + hd, q0 >> 5, 2 * kt);
      keep_masks_issue(km[0], kbp);
      keep_masks_issue(km[1], kbp + 16);
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { sc[blk][i] = nL2; dp[blk][i] = ndl; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#if defined(AFM_ABL) && AFM_ABL == 6      // timing only: fragment reads without the MFMAs
        const e16x8 fk = frag_row(Krow, 32 * blk, ks, lane), fv = frag_row(Vrow, 32 * blk, ks, lane);
        asm volatile("" :: "v"(fk), "v"(fv));
#elif defined(AFM_ABL) && AFM_ABL == 7    // timing only: MFMAs on register operands, no fragment reads
        sc[blk] = mfma32(qf[(ks + 1) & 3], qf[ks], sc[blk]);
        dp[blk] = mfma32(dof[(ks + 1) & 3], dof[ks], dp[blk]);
#else
        sc[blk] = mfma32(frag_row(Krow, 32 * blk, ks, lane), qf[ks], sc[blk]);
        dp[blk] = mfma32(frag_row(Vrow, 32 * blk, ks, lane), dof[ks], dp[blk]);
#endif
      }
      asm volatile("" ::: "memory");     // fragment reads of the second block stay behind the first block's
    }
  };
  auto vector_seg = [&](int kt) {
#if defined(AFM_ABL) && AFM_ABL == 5      // timing only: no vector segment
    if (DROP == DROP_BITS) { keep_masks_wait(km[0]); keep_masks_wait(km[1]); }
    return;
#endif
    const int kb = kt * KT;
    const unsigned long long mword = maskw[kt];
    const unsigned long long pad = mword >> (4 * h);
    const bool masked = mword != 0ull || (a.causal && (kb + KT - 1 > q0));
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      if (DROP == DROP_HASH) drop_block_select(a.dd, rowbase, kb + 32 * blk, h, dp[blk], ndl);
      if (DROP == DROP_BITS) drop_select_masks(dp[blk], km[blk], ndl);
      if (masked) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ko = 32 * blk + ACC_ROW(r);
          bool msk = (pad >> ko) & 1ull;
          if (a.causal) msk = msk || (kb + ko + 4 * h > q);
          sc[blk][r] = msk ? -INFINITY : sc[blk][r];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[blk][r] = fast_exp2(sc[blk][r]) * dp[blk][r];     // dS^T = P (D dP - delta)
      ds[2 * blk] = cvt8(sc[blk], 0);
      ds[2 * blk + 1] = cvt8(sc[blk], 1);
    }
  };
  auto dq_mm = [&](int kt) {        // MFMA segment, first half: dQ^T += K^T dS^T
    const unsigned char* Ktr = lds + (kt % NST) * STAGE + KT * DH * 2;
    unsigned ka0, ka1;
    tr_lane_addr(Ktr, lane, ka0, ka1);
    TrQuad kq[2];
    kq[0] = tr_issue(ka0, ka1, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < 3) kq[(i + 1) & 1] = tr_issue(ka0, ka1, 16 * (i + 1));
      if (i < 3) tr_wait<4>(); else tr_wait<0>();
#if defined(AFM_ABL) && AFM_ABL == 6
      asm volatile("" :: "v"(kq[i & 1].lo0), "v"(kq[i & 1].hi0), "v"(kq[i & 1].lo1), "v"(kq[i & 1].hi1));
#elif defined(AFM_ABL) && AFM_ABL == 7
      dq[0] = mfma32(qf[i], ds[i], dq[0]);
      dq[1] = mfma32(dof[i], ds[i], dq[1]);
#else
      dq[0] = mfma32(tr_join(kq[i & 1].lo0, kq[i & 1].hi0), ds[i], dq[0]);
      dq[1] = mfma32(tr_join(kq[i & 1].lo1, kq[i & 1].hi1), ds[i], dq[1]);
#endif
    }
  };
  __builtin_assume(ntiles >= 1);
  bool pl = false;
  if (!late) {
    for (int kt = 0; kt < ntiles; ++kt) {
      sync(kt);
      if (pl) dq_mm(kt - 1);
      pl = live(kt);
      if (pl) { products(kt); vector_seg(kt); }
    }
    if (pl) dq_mm(ntiles - 1);
  } else {
    __builtin_amdgcn_s_setprio(1);     // (measured: 0.571 -> 0.621 ms without it)
    bool lv = live(0);
    if (lv) products(0);
    for (int kt = 0; kt < ntiles; ++kt) {
      sync(kt);
      if (lv) { vector_seg(kt); dq_mm(kt); }
      lv = kt + 1 < ntiles && live(kt + 1);
      if (lv) products(kt + 1);
    }
  }
  if (q < a.Tq) {
    e16* dqp = dQ + ((int64_t)b * a.Tq + q) * a.lddq + hd * DH + 4 * h;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        e16x4 v = {(e16)(dq[db][4 * g4 + 0] * a.scale), (e16)(dq[db][4 * g4 + 1] * a.scale),
                    (e16)(dq[db][4 * g4 + 2] * a.scale), (e16)(dq[db][4 * g4 + 3] * a.scale)};
        *(e16x4*)(dqp + 32 * db + 8 * g4) = v;
      }
  }
}

// ------------------------------------------------------------------------------------------ dK, dV
// Workgroup = 4 waves x 32 keys; loops over 64-query tiles.  S = Q K^T with the key on the lane
// (queries in registers), P = exp2(S - lse[q]), dP = dO V^T, dS = P (D dP - delta[q]);
// dV^T[d][key] += sum_q dO^T[d][q] (D P)[q][key],  dK^T[d][key] += sum_q Q^T[d][q] dS[q][key].
template <int DROP>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dkv_mfma(AttnM a, const e16* __restrict__ Q,
                                                           const e16* __restrict__ K,
                                                           const e16* __restrict__ V,
                                                           const e16* __restrict__ dO,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ delta,
                                                           e16* __restrict__ dK, e16* __restrict__ dV) {
  // stage: Q and dO as dual-use images (afm_attn_tiles.h: one image serves the row reads of S / dP and the transposed reads of
  // dK / dV; round 3: half the LDS and half the LDS-DMA pieces of the row + transposed pair), lse[64], delta[64]; 2-stage ring
  constexpr int IMG = KT * DH * 2;
  constexpr int STAGE = 2 * IMG + 2 * KT * 4 + 4 * 256;   // + two 128-byte keep-bit blocks (the tile's two 32-query blocks) per wave
  constexpr int DS = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tk + 127) / 128);
  const int hd = blk_.hd, b = blk_.b;
  const int k0 = blk_.xb * 128 + w * 32;
  const int key = k0 + (lane & 31);
  const int kc = key < a.Tk ? key : a.Tk - 1;
  const bool kmasked = key >= a.Tk || (a.key_pad && a.key_pad[(int64_t)b * a.Tk + kc]);
  const e16* Qb = Q + (int64_t)b * a.Tq * a.ldq + hd * DH;
  const e16* Db = dO + (int64_t)b * a.Tq * a.ldo + hd * DH;
  e16x8 kf[4], vf[4];
  {
    const e16* kp = K + ((int64_t)b * a.Tk + kc) * a.ldk + hd * DH + 8 * h;
    const e16* vp = V + ((int64_t)b * a.Tk + kc) * a.ldv + hd * DH + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) { kf[s] = ld8_once(kp + 16 * s); vf[s] = ld8_once(vp + 16 * s); }
    // round 3: K pre-multiplied by scale * log2(e) (S comes out in log2 units, -lse[q] is the chain's initial value: p = exp2(S')
    // is one instruction), V by the dropout scale (dP' = scale * dP with -delta[q] as initial value: dS = p * (keep ? acc : -delta));
    // dV accumulates the UNSCALED dropped probabilities and takes the dropout scale once at the end
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        kf[s][j] = (e16)((float)kf[s][j] * a.scale_log2);
        if (DROP != DROP_NONE) vf[s][j] = (e16)((float)vf[s][j] * a.dd.scale16);
      }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
  const bool wave_all_masked = __all(kmasked);

  int qbeg = 0;
  if (a.causal) qbeg = (blk_.xb * 128) / KT * KT;   // queries before the block's first key see none of it
  const int ntiles = (a.Tq - qbeg + KT - 1) / KT;
  const int64_t lbase = ((int64_t)b * a.H + hd) * a.Tq;
  // qskip (self-attention, Tq == Tk): 64-query tiles of nothing but padding carry dO = 0 and delta = 0, i.e. they add exact zeros
  // to dK and dV: skipped.  One mask word per query tile, behind the ring.
  unsigned long long* qmaskw = (unsigned long long*)(lds + DS * STAGE);
  int* const tl = (int*)(qmaskw + (a.Tq + KT - 1) / KT) + 1;      // the query tiles to visit (build_tile_list): all, or the ones with a real query
  if (a.qskip) build_mask_words(qmaskw, a.key_pad, b, a.Tq, (a.Tq + KT - 1) / KT, w, lane);
  if (__syncthreads_and(wave_all_masked)) {   // 128 padded keys: their dK / dV rows are zeros (see the end of the kernel), nothing to load
    if (key < a.Tk) {
      e16* dkp = dK + ((int64_t)b * a.Tk + key) * a.lddk + hd * DH + 4 * h;
      e16* dvp = dV + ((int64_t)b * a.Tk + key) * a.lddv + hd * DH + 4 * h;
      const e16x4 z = {(e16)0.f, (e16)0.f, (e16)0.f, (e16)0.f};
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) { *(e16x4*)(dkp + 32 * db + 8 * g4) = z; *(e16x4*)(dvp + 32 * db + 8 * g4) = z; }
    }
    return;
  }
  build_tile_list(tl, a.qskip ? qmaskw : nullptr, qbeg / KT, qbeg / KT + ntiles, w, lane);
  __syncthreads();   // K / V fragment loads retired before the LDS-DMA ring starts
  const int nlive = __builtin_amdgcn_readfirstlane(tl[-1]);
  auto issue = [&](int j) {
    unsigned char* st = lds + (j % DS) * STAGE;
    const int row0 = tl[j] * KT;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      dma_piece_dual(st, Qb, a.ldq, row0, a.Tq, w + 4 * u, lane);
      dma_piece_dual(st + IMG, Db, a.ldo, row0, a.Tq, w + 4 * u, lane);
    }
    if (w < 2) {   // lse / delta of the tile's 64 queries: one 4-byte piece each
      int qq = row0 + lane;
      qq = qq < a.Tq ? qq : a.Tq - 1;
      const float* src = (w == 0 ? lse : delta) + lbase + qq;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + 2 * IMG + w * KT * 4), 4, 0, 0);
    }
    if (DROP == DROP_BITS) {   // keep-bit blocks (query block of lanes 0-31 / 32-63, this wave's key block): 2 x 32 dwords
      // (bits_block returns a wave-uniform pointer: the second query block of lanes 32-63 is a per-lane offset on top of it)
      const uint32_t* src = (const uint32_t*)bits_block(a, b * a.H + hd, row0 >> 5, min(k0 >> 5, a.nk32 - 1)) + (lane >> 5) * (a.nk32 * 32) + (lane & 31);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + 2 * IMG + 2 * KT * 4 + w * 256), 4, 0, 0);
    }
  };
  const unsigned t0 = tr_dual_t0(lane);
  issue(0);
  __builtin_assume(nlive >= 1);   // see k_attn_fwd_mfma
  for (int j = 0; j < nlive; ++j) {
    const int qb = __builtin_amdgcn_readfirstlane(tl[j]) * KT;
    attn_wait_vmcnt<0>();          // this tile's pieces (the only ones in flight)
    __builtin_amdgcn_s_barrier();
    if (j + 1 < nlive) issue(j + 1);
    const unsigned char* Qrow = lds + (j % DS) * STAGE;
    const unsigned char* Drow = Qrow + IMG;
    const float* Ls = (const float*)(Qrow + 2 * IMG);   // lse (natural log units)
    const float* Ds = Ls + KT;
    const bool ragged = qb + KT > a.Tq;   // wave-uniform: tile holds rows past Tq (clamped duplicates)
    // wave-uniform skips: every query of the tile precedes this wave's keys / its 32 keys are all padding
    if ((a.causal && qb + KT - 1 < k0) || wave_all_masked || (a.qskip && qmaskw[qb / KT] == ~0ull)) continue;
    // transposed-read address registers of this stage: base + (T0 ^ (delta << 4)) [^ 64 for the upper column half]
    const unsigned sb = (unsigned)(uintptr_t)Qrow;
    unsigned xa[4], xb[4];
#pragma unroll
    for (int dd_ = 0; dd_ < 4; ++dd_) { xa[dd_] = sb + (t0 ^ (dd_ << 4)); xb[dd_] = sb + (t0 ^ (dd_ << 4) ^ 64); }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      f32x16 s, dp;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {    // initial accumulators: -lse[q] (log2 units; lse = +inf for an all-masked row) and -delta[q]
        const f32x4 Lq = *(const f32x4*)(Ls + 32 * blk + 8 * g4 + 4 * h) * -1.4426950408889634f;
        const f32x4 Dq = *(const f32x4*)(Ds + 32 * blk + 8 * g4 + 4 * h);      // -delta (the dQ kernel stores it negated)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[4 * g4 + j] = Lq[j]; dp[4 * g4 + j] = Dq[j]; }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = mfma32(frag_row_dual(Qrow, 32 * blk, ks, lane), kf[ks], s);     // S'[q][key] = S log2(e) / sqrt(dh) - lse[q]
        dp = mfma32(frag_row_dual(Drow, 32 * blk, ks, lane), vf[ks], dp);   // scale dP[q][key] - delta[q]
      }
      f32x16 pd;
      // Uniform conditions select whole loops (a branch per score would sit inside the unrolled body).
      if (a.causal || ragged) {   // rare: diagonal tiles of the decoder / the last, partly filled tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qq = qb + 32 * blk + ACC_ROW(r) + 4 * h;
          const bool msk = (a.causal && key > qq) || qq >= a.Tq;
          s[r] = msk ? -INFINITY : s[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) pd[r] = fast_exp2(s[r]);     // masked keys: outputs zeroed at the end
      if (DROP == DROP_BITS) {   // one dword per lane and 32-query block: bit q = keep(query q, this lane's key)
        const uint32_t word = ((const uint32_t*)(Qrow + 2 * IMG + 2 * KT * 4 + w * 256))[32 * blk + bits_word_of_key(lane & 31)] >> (4 * h);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 nd = *(const f32x4*)(Ds + 32 * blk + 8 * g4 + 4 * h);   // -delta again, from LDS: 16 registers less to hold
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * g4 + j;
            const bool kp = (int)(word << (31 - ACC_ROW(r))) < 0;      // shift the row's bit into the sign: one shift, one compare
            s[r] = pd[r] * (kp ? dp[r] : nd[j]);                       // dS = P (D dP - delta)
            pd[r] = kp ? pd[r] : 0.f;                                  // dropped P for dV
          }
        }
      } else if (DROP == DROP_HASH) {
        // Dropout keep bits (two-level stream, afm_common.h).  Element (q, key): the two lanes of a key pair (lane, lane^1) share
        // the pair mix of row (lbase + q)'s hash and take its low / high 16 bits.  The even lane hashes the even register rows,
        // the odd lane the odd rows, and a quad-permute DPP move hands each lane its partner's hash.  (The key sits on the lane
        // here, so every register row costs a row hash as well: this re-hash path is the slow one; training reads the keep bits.)
        const uint64_t tb = (uint64_t)(lbase + qb + 32 * blk + 4 * h + (lane & 1));
        const uint32_t po = afm_pair_offset((uint32_t)key >> 1);
        const uint32_t hshift = (lane & 1) << 4;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const uint32_t own = afm_pair_mix(afm_row_hash(a.dd, tb + (uint64_t)ACC_ROW(r)) + po);   // row r + (lane&1)
          const uint32_t oth = (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);  // lane^1
          const uint32_t h0 = (lane & 1) ? oth : own, h1 = (lane & 1) ? own : oth;
          const bool k0 = ((h0 >> hshift) & 0xFFFFu) >= a.dd.thresh16, k1 = ((h1 >> hshift) & 0xFFFFu) >= a.dd.thresh16;
          const float nd0 = Ds[32 * blk + 8 * (r >> 2) + 4 * h + (r & 3)], nd1 = Ds[32 * blk + 8 * ((r + 1) >> 2) + 4 * h + ((r + 1) & 3)];
          s[r] = pd[r] * (k0 ? dp[r] : nd0); s[r + 1] = pd[r + 1] * (k1 ? dp[r + 1] : nd1);
          pd[r] = k0 ? pd[r] : 0.f; pd[r + 1] = k1 ? pd[r + 1] : 0.f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = pd[r] * dp[r];
      }
      {
        // 16-bit operands first (the fp32 P / dS registers die here), then the transposed reads two quads at a time.
        // (Forcing three workgroups per CU -- 168 registers, with or without the V fragments parked in LDS -- was measured:
        // 1.02 .. 1.24 ms against 0.78 at two per CU; per product this kernel already runs at the dQ kernel's rate.)
        const e16x8 pf0 = cvt8_pk(pd, 0), dsf0 = cvt8_pk(s, 0), pf1 = cvt8_pk(pd, 1), dsf1 = cvt8_pk(s, 1);
        constexpr int QO = 0, DO_ = IMG;
#define AFM_DKV_QUADS(B32)                                                                                        \
        {                                                                                                         \
          const TrQuad d0 = tr_quad_dual<DO_ + B32, 0>(xa, xb), q0f = tr_quad_dual<QO + B32, 0>(xa, xb);          \
          tr_wait<4>();                                                                                           \
          dv[0] = mfma32(tr_join(d0.lo0, d0.hi0), pf0, dv[0]);                                                    \
          dv[1] = mfma32(tr_join(d0.lo1, d0.hi1), pf0, dv[1]);                                                    \
          const TrQuad d1 = tr_quad_dual<DO_ + B32, 1>(xa, xb);                                                   \
          tr_wait<4>();                                                                                           \
          dk[0] = mfma32(tr_join(q0f.lo0, q0f.hi0), dsf0, dk[0]);                                                 \
          dk[1] = mfma32(tr_join(q0f.lo1, q0f.hi1), dsf0, dk[1]);                                                 \
          const TrQuad q1f = tr_quad_dual<QO + B32, 1>(xa, xb);                                                   \
          tr_wait<4>();                                                                                           \
          dv[0] = mfma32(tr_join(d1.lo0, d1.hi0), pf1, dv[0]);                                                    \
          dv[1] = mfma32(tr_join(d1.lo1, d1.hi1), pf1, dv[1]);                                                    \
          tr_wait<0>();                                                                                           \
          dk[0] = mfma32(tr_join(q1f.lo0, q1f.hi0), dsf1, dk[0]);                                                 \
          dk[1] = mfma32(tr_join(q1f.lo1, q1f.hi1), dsf1, dk[1]);                                                 \
        }
        if (blk == 0) AFM_DKV_QUADS(0) else AFM_DKV_QUADS(32 * 128)
#undef AFM_DKV_QUADS
      }
    }
  }
  if (kmasked) {   // a padded key took no part in any softmax: its dK / dV rows are zero
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
  }
  if (DROP != DROP_NONE) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { dv[0][i] *= a.dd.scale16; dv[1][i] *= a.dd.scale16; }
  }
  if (key < a.Tk) {
    e16* dkp = dK + ((int64_t)b * a.Tk + key) * a.lddk + hd * DH + 4 * h;
    e16* dvp = dV + ((int64_t)b * a.Tk + key) * a.lddv + hd * DH + 4 * h;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        e16x4 x = {(e16)(dk[db][4 * g4 + 0] * a.scale), (e16)(dk[db][4 * g4 + 1] * a.scale),
                    (e16)(dk[db][4 * g4 + 2] * a.scale), (e16)(dk[db][4 * g4 + 3] * a.scale)};
        e16x4 y = {(e16)dv[db][4 * g4 + 0], (e16)dv[db][4 * g4 + 1], (e16)dv[db][4 * g4 + 2], (e16)dv[db][4 * g4 + 3]};
        *(e16x4*)(dkp + 32 * db + 8 * g4) = x;
        *(e16x4*)(dvp + 32 * db + 8 * g4) = y;
      }
  }
}

#include "afm_attn_pipe_impl.h"

// ------------------------------------------------------------------------------------------ keep-bit tensor, filled ahead
// The forward's own dropout bits cost it a hash per score pair (0.53 vs 0.40 ms at the c2 shape).  The bits depend on nothing but
// (seed, site, shape): this kernel writes the whole tensor -- one wave per (batch*head, 32-query block), all its 32-key blocks,
// the forward kernel's indexing to the letter -- so it can run on another stream under the HBM-bound LayerNorm that precedes the
// attention block, and the forward reads lane masks like the dQ kernel does (DROP_READ).
__global__ __launch_bounds__(256) void k_attn_bits_fill(AttnM a) {
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int64_t wave = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int64_t nrows = (int64_t)a.B * a.H * a.nq32;
  if (wave >= nrows) return;
  const int bh = (int)(wave / a.nq32), qb32 = (int)(wave % a.nq32);
  const int q = qb32 * 32 + (lane & 31);
  const int qc = q < a.Tq ? q : a.Tq - 1;
  const uint32_t rowbase = afm_row_hash(a.dd, (uint64_t)bh * a.Tq + qc);
  for (int kb32 = 0; kb32 < a.nk32; ++kb32)
    bits_emit_rows<0>(a.dd, pair_base(rowbase, kb32 * 32, h), bits_block(a, bh, qb32, kb32));
  bits_flush();
}

#include "afm_attn_m16_impl.h"
#include "afm_attn_pipe16_impl.h"
#include "afm_attn_fwd16_impl.h"
#include "afm_attn_sq_impl.h"
#include "afm_attn_fsq_impl.h"

}  // namespace AFM_E16_NS
using namespace AFM_E16_NS;

// ------------------------------------------------------------------------------------------ dispatch
// packed rows (q_off / k_off): the default single-pass kernels only (forward, dQ in both MFMA shapes, the pipelined 16 x 16 x 32 dK/dV), no
// causal mask, whole 128-row blocks, key_pad given on a packed key side
static bool packed_ok(const afm_attn_shape* s) {
  if (!s->q_off && !s->k_off) return true;
  if (s->causal || (s->q_off && (s->Tq & 127)) || (s->k_off && ((s->Tk & 127) || !s->key_pad))) return false;
  if (s->q_off && s->q_off != s->k_off) return false;      // (a packed query side is the encoder's self-attention: the same rows on both sides)
  return !(s->reserved & (16 | 128 | 256 | 512 | 4096 | 16384 | 65536 | 1024 | 2048));
}
static bool eligible(const afm_attn_shape* s, const void* const* ptrs, int nptr, const int* lds, int nld) {
  if (s->dtype != AFM_E16 || s->dh != DH) return false;
  if (!packed_ok(s)) return false;
  if (s->sqb || s->skb || s->svb || s->sob) return false;   // KV-cache strides: generic kernel
  if (s->causal && s->Tq != s->Tk) return false;            // the tile loops assume >= 1 tile per workgroup (self-attention)
  if (s->drop.p > 0.f && (s->Tk & 1)) return false;   // the pair hash needs even rows of the mask
  if (s->drop.p > 0.f && (uint64_t)s->B * s->H * s->Tq * (uint64_t)s->Tk > 0xFFFFFFFFull) return false;  // 32-bit mask index
  for (int i = 0; i < nptr; ++i) if ((uintptr_t)ptrs[i] & 15) return false;
  for (int i = 0; i < nld; ++i) if (lds[i] & 7) return false;
  return true;
}
static AttnM make_m(const afm_attn_shape* s) {
  AttnM a;
  a.B = s->B; a.H = s->H; a.Tq = s->Tq; a.Tk = s->Tk;
  a.ldq = s->ldq; a.ldk = s->ldk; a.ldv = s->ldv; a.ldo = s->ldo;
  a.lddq = a.lddk = a.lddv = 0;
  a.causal = s->causal; a.scale = s->scale; a.scale_log2 = s->scale * 1.4426950408889634f;
  a.key_pad = s->key_pad; a.dd = afm_make_drop(&s->drop);
  a.bits = a.dd.thresh16 ? (unsigned long long*)s->drop_bits : nullptr;
  a.nq32 = ((s->Tq + 127) / 128) * 4; a.nk32 = ((s->Tk + 63) / 64) * 2;      // whole workgroups / whole 64-key tiles
  a.qskip = (s->reserved & 64) && s->key_pad && s->Tq == s->Tk;
  a.q_off = s->q_off; a.k_off = s->k_off;
  a.nofill = (s->reserved & 131072) != 0;
  // packed self-attention: query blocks beyond a sample's slot are other samples' rows -- the padded-query skip is not optional there
  if (a.q_off && a.k_off && s->key_pad && s->Tq == s->Tk) a.qskip = 1;
  return a;
}

int AFM_E16_FN(afm_attn_fwd_mfma_try)(const afm_attn_shape* s, const void* Q, const void* K, const void* V, void* O,
                          float* lse, hipStream_t st) {
  const void* ptrs[] = {Q, K, V, O};
  const int lds[] = {s->ldq, s->ldk, s->ldv, s->ldo};
  if (!eligible(s, ptrs, 4, lds, 4)) return AFM_ERR_UNSUPPORTED;
  const AttnM a = make_m(s);
  // 8 staggered waves per workgroup (256 queries): built and verified in round 3, 10 % faster than the 4-wave kernel on random
  // operands in isolation (0.61 vs 0.68 ms at the c2 shape) but SLOWER inside the training step (0.67 vs 0.58 ms per launch in the
  // same rocprofv3 run, same box): not the default.  afm_attn_shape.reserved & 16 or AFM_ATTN_8WAVE=1 selects it.
  static const bool eight_wave = getenv("AFM_ATTN_8WAVE") != nullptr;
  const bool packed = s->q_off || s->k_off;      // (packed rows: the four-wave kernel only)
  if (!packed && s->Tq >= 256 && ((s->reserved & 16) || eight_wave)) {
    const dim3 grid8(((s->Tq + 255) / 256) * s->H * s->B);
    const int shm8 = 4 * 2 * KT * DH * 2 + ((s->Tk + KT - 1) / KT) * 8;
    if (shm8 > 80 * 1024) return AFM_ERR_UNSUPPORTED;
    static AfmOncePerDevice attr8;
    if (attr8.need()) {
      (void)hipFuncSetAttribute((const void*)k_attn_fwd_st<DROP_HASH>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute((const void*)k_attn_fwd_st<DROP_BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute((const void*)k_attn_fwd_st<DROP_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    }
    if (a.dd.thresh16 && a.bits) AFM_LAUNCH(k_attn_fwd_st<DROP_BITS>, grid8, dim3(512), shm8, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (e16*)O, lse);
    else if (a.dd.thresh16) AFM_LAUNCH(k_attn_fwd_st<DROP_HASH>, grid8, dim3(512), shm8, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (e16*)O, lse);
    else AFM_LAUNCH(k_attn_fwd_st<DROP_NONE>, grid8, dim3(512), shm8, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (e16*)O, lse);
    afm_set_last_algo("attn_mfma");
    return AFM_OK;
  }
  const dim3 grid(((s->Tq + 127) / 128) * s->H * s->B);
  const int shm = RS * 2 * KT * DH * 2 + ((s->Tk + KT - 1) / KT) * 12 + 8;      // ring, key-mask words, tile list
  if (shm > 64 * 1024) return AFM_ERR_UNSUPPORTED;
  if (s->reserved & 1024) {      // the forward on v_mfma_f32_16x16x32 (afm_attn_fwd16_impl.h: an A / B form); & 2048: three workgroups per CU (168 registers)
#define AFM_F16_LAUNCH(D) do { if (s->reserved & 2048) AFM_LAUNCH((k_attn_fwd_m16<D, 3>), grid, dim3(256), shm, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (e16*)O, lse); \
      else AFM_LAUNCH((k_attn_fwd_m16<D, 4>), grid, dim3(256), shm, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (e16*)O, lse); } while (0)
    if (a.dd.thresh16 && a.bits && (s->reserved & 32)) AFM_F16_LAUNCH(DROP_READ);
    else if (a.dd.thresh16 && a.bits) AFM_F16_LAUNCH(DROP_BITS);
    else if (a.dd.thresh16) AFM_F16_LAUNCH(DROP_HASH);
    else AFM_F16_LAUNCH(DROP_NONE);
#undef AFM_F16_LAUNCH
    afm_set_last_algo("attn_mfma");
    return AFM_OK;
  }
  if (a.dd.thresh16 && a.bits && (s->reserved & 32)) AFM_LAUNCH(k_attn_fwd_mfma<DROP_READ>, grid, dim3(256), shm, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (e16*)O, lse);
  else if (a.dd.thresh16 && a.bits) AFM_LAUNCH(k_attn_fwd_mfma<DROP_BITS>, grid, dim3(256), shm, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (e16*)O, lse);
  else if (a.dd.thresh16) AFM_LAUNCH(k_attn_fwd_mfma<DROP_HASH>, grid, dim3(256), shm, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (e16*)O, lse);
  else AFM_LAUNCH(k_attn_fwd_mfma<DROP_NONE>, grid, dim3(256), shm, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (e16*)O, lse);
  afm_set_last_algo("attn_mfma");
  return AFM_OK;
}

// Fill the keep-bit tensor of `s` (the bits the forward kernel would have written).  Only the shape and the dropout stream matter.
int AFM_E16_FN(afm_attn_bits_fill_try)(const afm_attn_shape* s, hipStream_t st) {
  if (s->dtype != AFM_E16 || s->dh != DH || s->drop.p <= 0.f || !s->drop_bits) return AFM_ERR_UNSUPPORTED;
  if (s->sqb || s->skb || s->svb || s->sob || (s->causal && s->Tq != s->Tk) || (s->Tk & 1)) return AFM_ERR_UNSUPPORTED;
  if ((uint64_t)s->B * s->H * s->Tq * (uint64_t)s->Tk > 0xFFFFFFFFull) return AFM_ERR_UNSUPPORTED;
  const AttnM a = make_m(s);
  const int64_t nrows = (int64_t)a.B * a.H * a.nq32;
  AFM_LAUNCH(k_attn_bits_fill, dim3((int)((nrows + 3) / 4)), dim3(256), 0, st, a);
  return AFM_OK;
}

int AFM_E16_FN(afm_attn_bwd_mfma_try)(const afm_attn_shape* s, const void* Q, const void* K, const void* V, const void* O,
                          const void* dO, const float* lse, float* delta, void* dQ, void* dK, void* dV,
                          int lddq, int lddk, int lddv, hipStream_t st) {
  const void* ptrs[] = {Q, K, V, O, dO, dQ, dK, dV};
  const int lds[] = {s->ldq, s->ldk, s->ldv, s->ldo, lddq, lddk, lddv};
  if (!eligible(s, ptrs, 8, lds, 7)) return AFM_ERR_UNSUPPORTED;
  AttnM a = make_m(s);
  a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  const dim3 gq(((s->Tq + 127) / 128) * s->H * s->B), gk(((s->Tk + 127) / 128) * s->H * s->B);
  const int shm_q = RS * 3 * KT * DH * 2 + ((s->Tk + KT - 1) / KT) * 12 + 8;
  if (shm_q > 80 * 1024) return AFM_ERR_UNSUPPORTED;
  static AfmOncePerDevice attr_q;
  if (attr_q.need()) {
    (void)hipFuncSetAttribute((const void*)k_attn_bwd_dq_mfma<DROP_HASH>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)k_attn_bwd_dq_mfma<DROP_BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)k_attn_bwd_dq_mfma<DROP_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  }
  const bool run_q = (s->reserved & 3) != 2, run_k = (s->reserved & 3) != 1;   // reserved & 3 = 1 / 2: only the dQ / only the dK-dV kernel (timing)
  // Round 6: short query sequences (the decoder's cross-attention), dQ, dK and dV in ONE kernel (afm_attn_fsq_impl.h): reserved & 262144
  if ((s->reserved & 262144) && (s->reserved & 3) == 0 && (!s->causal || (s->Tq == s->Tk && !s->k_off)) && s->Tq <= 128 && !s->q_off && (!s->k_off || s->key_pad) &&
      (!a.dd.thresh16 || a.bits)) {
    const int shm_f = RS * 2 * KT * DH * 2 + 4 * KT * DH * 2 + 3 * 4096 + ((s->Tk + KT - 1) / KT) * 12 + 16;      // ring, P / dS tiles, parked fragments, key-mask words and tile list
    if (shm_f <= 80 * 1024) {
      static AfmOncePerDevice attr_f;
      if (attr_f.need()) {
        (void)hipFuncSetAttribute((const void*)k_attn_bwd_fsq<DROP_BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)k_attn_bwd_fsq<DROP_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      }
      const dim3 gf(s->H * s->B);
#ifdef AFM_ATTN_ABLATIONS
#define AFM_FSQ_ABL_CASE(N) case N: (void)hipFuncSetAttribute((const void*)k_attn_bwd_fsq<DROP_BITS, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
      AFM_LAUNCH((k_attn_bwd_fsq<DROP_BITS, N>), gf, dim3(256), shm_f, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)O, (const e16*)dO, lse, delta, (e16*)dQ, (e16*)dK, (e16*)dV); return AFM_OK;
      if (a.dd.thresh16 && ((s->reserved >> 20) & 255)) {
        switch ((s->reserved >> 20) & 255) {
          AFM_FSQ_ABL_CASE(1) AFM_FSQ_ABL_CASE(2) AFM_FSQ_ABL_CASE(3) AFM_FSQ_ABL_CASE(4) AFM_FSQ_ABL_CASE(7) AFM_FSQ_ABL_CASE(8) AFM_FSQ_ABL_CASE(15) AFM_FSQ_ABL_CASE(16) AFM_FSQ_ABL_CASE(23) AFM_FSQ_ABL_CASE(31)
          default: return AFM_ERR_UNSUPPORTED;
        }
      }
#endif
      if (a.dd.thresh16) AFM_LAUNCH(k_attn_bwd_fsq<DROP_BITS>, gf, dim3(256), shm_f, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)O, (const e16*)dO, lse, delta, (e16*)dQ, (e16*)dK, (e16*)dV);
      else AFM_LAUNCH(k_attn_bwd_fsq<DROP_NONE>, gf, dim3(256), shm_f, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)O, (const e16*)dO, lse, delta, (e16*)dQ, (e16*)dK, (e16*)dV);
      afm_set_last_algo("attn_fsq");
      return AFM_OK;
    }
  }
  static const bool eight_wave = getenv("AFM_ATTN_8WAVE") != nullptr;
  const bool packed = s->q_off || s->k_off;
  // packed rows exist in the pipelined 16 x 16 x 32 dK/dV kernel only: refuse BEFORE anything is launched where it would not run
  // (dropout without a keep-bit tensor, a query length that is not whole tiles)
  if (packed && run_k && !(!s->causal && (s->Tq % KT) == 0 && (!a.dd.thresh16 || a.bits) &&
                           3 * 2 * KT * DH * 2 + 2 * (2048 + 4 * 1024) + (s->Tq / KT) * 12 + 8 <= 80 * 1024)) return AFM_ERR_UNSUPPORTED;
  const bool q8 = !packed && s->Tq >= 256 && ((s->reserved & 16) || eight_wave);   // the 8-wave staggered dQ kernel (see afm_attn_fwd_mfma_try: not the default)
  const int shm_q8 = 4 * 3 * KT * DH * 2 + ((s->Tk + KT - 1) / KT) * 8;
  if (run_q && q8) {
    static AfmOncePerDevice attr_q8;
    if (attr_q8.need()) {
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dq_st<DROP_HASH>, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dq_st<DROP_BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dq_st<DROP_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    }
    if (shm_q8 > 112 * 1024) return AFM_ERR_UNSUPPORTED;
    const dim3 gq8(((s->Tq + 255) / 256) * s->H * s->B);
    if (a.dd.thresh16 && a.bits) AFM_LAUNCH(k_attn_bwd_dq_st<DROP_BITS>, gq8, dim3(512), shm_q8, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)O, (const e16*)dO, lse, delta, (e16*)dQ);
    else if (a.dd.thresh16) AFM_LAUNCH(k_attn_bwd_dq_st<DROP_HASH>, gq8, dim3(512), shm_q8, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)O, (const e16*)dO, lse, delta, (e16*)dQ);
    else AFM_LAUNCH(k_attn_bwd_dq_st<DROP_NONE>, gq8, dim3(512), shm_q8, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)O, (const e16*)dO, lse, delta, (e16*)dQ);
  }
  else if (!run_q) {}
  // the 16x16x32 form of the dQ kernel (afm_attn_m16_impl.h); reserved & 2048: compiled for two workgroups per CU (256 registers).  Round 5,
  // c2 encoder shape, one process: without dropout 0.510 ... 0.516 ms (two workgroups per CU) against 0.571 ... 0.574 for the 32 x 32 x 16 kernel,
  // with the hash re-evaluated 0.556 ... 0.562 against 0.699 ... 0.701 -- the default in those two cases; with the keep bits READ it loses
  // (0.564 against 0.535: the lane-mask words do not match its score layout and are repacked per tile), so the training step's dropout
  // path keeps the 32 x 32 x 16 kernel.  reserved & 32768 keeps that kernel everywhere (A / B runs).
  // Round 6: with its ring pieces in saddr form and the transposed-read addresses recomputed per tile the three-workgroup build of the
  // 16 x 16 x 32 kernel spills nothing (168 registers), and with the keep bits it edges out the 32 x 32 x 16 kernel on long query
  // sequences (c2 encoder shape 591 ... 612 against 624 ... 627 us, c4's 229 against 250; at 128 queries it loses, 110 against 96) --
  // +0.2 % of the c2 step (3 189 -> 3 196).  NOT made the default for the keep-bit path: another rounding of dQ in every training step for
  // a gain inside the box-to-box spread (reserved & 1024 selects it).
  else if ((s->reserved & 1024) || (!(a.dd.thresh16 && a.bits) && !(s->reserved & 32768))) {
#define AFM_M16_LAUNCH(D, OCC) do { static AfmOncePerDevice at_; if (at_.need()) (void)hipFuncSetAttribute((const void*)k_attn_bwd_dq_m16<D, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
      AFM_LAUNCH((k_attn_bwd_dq_m16<D, OCC>), gq, dim3(256), shm_q, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)O, (const e16*)dO, lse, delta, (e16*)dQ); } while (0)
    const bool occ2 = (s->reserved & 2048) != 0 || (!(s->reserved & 1024) && !(a.dd.thresh16 && a.bits));      // (keep bits: the three-workgroup build)
    if (a.dd.thresh16 && a.bits) { if (occ2) AFM_M16_LAUNCH(DROP_BITS, 2); else AFM_M16_LAUNCH(DROP_BITS, 3); }
    else if (a.dd.thresh16) { if (occ2) AFM_M16_LAUNCH(DROP_HASH, 2); else AFM_M16_LAUNCH(DROP_HASH, 3); }
    else { if (occ2) AFM_M16_LAUNCH(DROP_NONE, 2); else AFM_M16_LAUNCH(DROP_NONE, 3); }
#undef AFM_M16_LAUNCH
  }
  else if (a.dd.thresh16 && a.bits) AFM_LAUNCH(k_attn_bwd_dq_mfma<DROP_BITS>, gq, dim3(256), shm_q, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)O, (const e16*)dO, lse, delta, (e16*)dQ);
  else if (a.dd.thresh16) AFM_LAUNCH(k_attn_bwd_dq_mfma<DROP_HASH>, gq, dim3(256), shm_q, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)O, (const e16*)dO, lse, delta, (e16*)dQ);
  else AFM_LAUNCH(k_attn_bwd_dq_mfma<DROP_NONE>, gq, dim3(256), shm_q, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)O, (const e16*)dO, lse, delta, (e16*)dQ);
  const int shm_k = 2 * (2 * KT * DH * 2 + 2 * KT * 4 + 4 * 256) + ((s->Tq + KT - 1) / KT) * 12 + 8;     // two stages of {Q, dO dual-use images, lse, delta, keep bits} + the query-tile mask words and list
  static AfmOncePerDevice attr_k;
  if (attr_k.need()) {
    (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_mfma<DROP_HASH>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_mfma<DROP_BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_mfma<DROP_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  }
  // the software-pipelined kernel (afm_attn_pipe_impl.h) where its conditions hold; reserved & 128 keeps the round-3 kernel (A / B tests),
  // reserved & 256 the eight-wave form (0.72 vs 0.70 ms at the c2 shape), reserved & 512 the form with 64 keys per wave and one wave per SIMD
  // (every LDS fragment feeds two MFMAs, but a lone wave hides nothing and its accumulators travel through v_accvgpr: 0.76 ms; 0.53 vs 0.37
  // over a padded batch) -- both bit-identical, both left as measured
  const bool piped = !s->causal && (s->Tq % KT) == 0 && (!a.dd.thresh16 || a.bits) && !(s->reserved & 128);
  const int pnw = (s->reserved & 256) ? 8 : 4;
  const int pkb = (pnw == 4 && (s->reserved & 512) && s->Tk >= 256) ? 2 : 1;
  const int shm_kp = 3 * 2 * KT * DH * 2 + 2 * (2048 + pnw * pkb * 1024) + (s->Tq / KT) * 12 + 8;
  const dim3 gkp(((s->Tk + 32 * pnw * pkb - 1) / (32 * pnw * pkb)) * s->H * s->B);
  if (run_k && piped) {
    static AfmOncePerDevice attr_kp;
    if (attr_kp.need()) {
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_pipe<DROP_BITS, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_pipe<DROP_NONE, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_pipe<DROP_BITS, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_pipe<DROP_NONE, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_pipe<DROP_BITS, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_pipe<DROP_NONE, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    }
  }
#define AFM_PIPE_LAUNCH(KERN, BLK) AFM_LAUNCH(KERN, gkp, dim3(BLK), shm_kp, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)dO, lse, delta, (e16*)dK, (e16*)dV)
#ifdef AFM_ATTN_ABLATIONS
#define AFM_PIPE_ABL_CASE(N) case N: (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_pipe<DROP_BITS, 4, 1, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
    AFM_PIPE_LAUNCH((k_attn_bwd_dkv_pipe<DROP_BITS, 4, 1, N>), 256); return AFM_OK;
  if (run_k && piped && pnw == 4 && pkb == 1 && a.dd.thresh16 && ((s->reserved >> 20) & 255)) {
    switch ((s->reserved >> 20) & 255) {
      AFM_PIPE_ABL_CASE(1) AFM_PIPE_ABL_CASE(2) AFM_PIPE_ABL_CASE(4) AFM_PIPE_ABL_CASE(6) AFM_PIPE_ABL_CASE(3) AFM_PIPE_ABL_CASE(7) AFM_PIPE_ABL_CASE(8)
      AFM_PIPE_ABL_CASE(16) AFM_PIPE_ABL_CASE(32) AFM_PIPE_ABL_CASE(23) AFM_PIPE_ABL_CASE(31) AFM_PIPE_ABL_CASE(63) AFM_PIPE_ABL_CASE(22) AFM_PIPE_ABL_CASE(54)
      AFM_PIPE_ABL_CASE(64) AFM_PIPE_ABL_CASE(80) AFM_PIPE_ABL_CASE(112) AFM_PIPE_ABL_CASE(120)
      default: return AFM_ERR_UNSUPPORTED;
    }
  }
#endif
  // Round 5: short query sequences (the decoder's cross-attention) -- all query tiles resident, the workgroup walks the head's key blocks
  // (afm_attn_sq_impl.h)
  const int ntq = (s->Tq + KT - 1) / KT;
  if (run_k && (s->reserved & 65536) && !s->causal && ntq <= 3 && s->Tk >= 256 && (!a.dd.thresh16 || a.bits)) {
    const int nkb = (s->Tk + 127) / 128, nbh = s->B * s->H;
    int nch = std::min(nkb, std::max(1, (1024 + nbh - 1) / nbh));
    const int nb = (nkb + nch - 1) / nch;
    nch = (nkb + nb - 1) / nb;
    const int shm_sq = ntq * (2 * KT * DH * 2 + 2 * KT * 4);      // per query tile: the Q and dO dual-use images, lse and delta (what the kernel lays out; ADVICE r05: twice that was requested)
    const dim3 gsq(nch * s->H * s->B);
#define AFM_SQ_LAUNCH(D, N) do { static AfmOncePerDevice at_; if (at_.need()) (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_sq<D, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
    AFM_LAUNCH((k_attn_bwd_dkv_sq<D, N>), gsq, dim3(256), shm_sq, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)dO, lse, delta, (e16*)dK, (e16*)dV, nb, nch); } while (0)
    if (a.dd.thresh16) { if (ntq == 1) AFM_SQ_LAUNCH(DROP_BITS, 1); else if (ntq == 2) AFM_SQ_LAUNCH(DROP_BITS, 2); else AFM_SQ_LAUNCH(DROP_BITS, 3); }
    else { if (ntq == 1) AFM_SQ_LAUNCH(DROP_NONE, 1); else if (ntq == 2) AFM_SQ_LAUNCH(DROP_NONE, 2); else AFM_SQ_LAUNCH(DROP_NONE, 3); }
#undef AFM_SQ_LAUNCH
    afm_set_last_algo("attn_mfma");
    return AFM_OK;
  }
  // Round 5: the software-pipelined kernel on v_mfma_f32_16x16x32 (afm_attn_pipe16_impl.h) is the default where the pipelined kernel
  // applies in its four-wave form -- one process, order swapped, c2 encoder shape, fp16: 0.652 ... 0.658 ms against 0.690 ... 0.694 with the
  // keep bits (-5.5 %), 0.564 ... 0.572 against 0.613 ... 0.622 without dropout (-8 %); results agree with the 32 x 32 x 16 kernels to rounding
  // (another accumulation order).  reserved & 16384 keeps the 32 x 32 x 16 pipelined kernel (A / B runs, its bit-identity test).
  if (run_k && !(s->reserved & (16384 | 4096)) && piped && shm_kp <= 80 * 1024 && pnw == 4 && pkb == 1) {
    static AfmOncePerDevice attr_kp16;
    if (attr_kp16.need()) {
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_pipe16<DROP_BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_pipe16<DROP_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    }
    if (a.dd.thresh16) AFM_LAUNCH(k_attn_bwd_dkv_pipe16<DROP_BITS>, gkp, dim3(256), shm_kp, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)dO, lse, delta, (e16*)dK, (e16*)dV);
    else AFM_LAUNCH(k_attn_bwd_dkv_pipe16<DROP_NONE>, gkp, dim3(256), shm_kp, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)dO, lse, delta, (e16*)dK, (e16*)dV);
  }
  else if (run_k && (s->reserved & 4096) && (!a.dd.thresh16 || a.bits)) {      // the 16x16x32 form of the round-3 dK/dV kernel (afm_attn_m16_impl.h: an A/B form)
    static AfmOncePerDevice attr_k16;
    if (attr_k16.need()) {
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_m16<DROP_BITS>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute((const void*)k_attn_bwd_dkv_m16<DROP_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    }
    if (a.dd.thresh16) AFM_LAUNCH(k_attn_bwd_dkv_m16<DROP_BITS>, gk, dim3(256), shm_k, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)dO, lse, delta, (e16*)dK, (e16*)dV);
    else AFM_LAUNCH(k_attn_bwd_dkv_m16<DROP_NONE>, gk, dim3(256), shm_k, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)dO, lse, delta, (e16*)dK, (e16*)dV);
  }
  else if (!run_k) {}
  else if (piped && shm_kp <= 80 * 1024 && a.dd.thresh16 && pnw == 8) AFM_PIPE_LAUNCH((k_attn_bwd_dkv_pipe<DROP_BITS, 8, 1>), 512);
  else if (piped && shm_kp <= 80 * 1024 && a.dd.thresh16 && pkb == 2) AFM_PIPE_LAUNCH((k_attn_bwd_dkv_pipe<DROP_BITS, 4, 2>), 256);
  else if (piped && shm_kp <= 80 * 1024 && a.dd.thresh16) AFM_PIPE_LAUNCH((k_attn_bwd_dkv_pipe<DROP_BITS, 4, 1>), 256);
  else if (piped && shm_kp <= 80 * 1024 && pnw == 8) AFM_PIPE_LAUNCH((k_attn_bwd_dkv_pipe<DROP_NONE, 8, 1>), 512);
  else if (piped && shm_kp <= 80 * 1024 && pkb == 2) AFM_PIPE_LAUNCH((k_attn_bwd_dkv_pipe<DROP_NONE, 4, 2>), 256);
  else if (piped && shm_kp <= 80 * 1024) AFM_PIPE_LAUNCH((k_attn_bwd_dkv_pipe<DROP_NONE, 4, 1>), 256);
  else if (a.dd.thresh16 && a.bits) AFM_LAUNCH(k_attn_bwd_dkv_mfma<DROP_BITS>, gk, dim3(256), shm_k, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)dO, lse, delta, (e16*)dK, (e16*)dV);
  else if (a.dd.thresh16) AFM_LAUNCH(k_attn_bwd_dkv_mfma<DROP_HASH>, gk, dim3(256), shm_k, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)dO, lse, delta, (e16*)dK, (e16*)dV);
  else AFM_LAUNCH(k_attn_bwd_dkv_mfma<DROP_NONE>, gk, dim3(256), shm_k, st, a, (const e16*)Q, (const e16*)K, (const e16*)V, (const e16*)dO, lse, delta, (e16*)dK, (e16*)dV);
  afm_set_last_algo("attn_mfma");
  return AFM_OK;
}
