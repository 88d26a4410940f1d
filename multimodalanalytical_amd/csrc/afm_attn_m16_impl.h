// (Dispatch as of the end of round 5: the dQ kernel below is the DEFAULT where no keep-bit tensor is read -- no dropout, or the hash
// re-evaluated -- in its two-workgroups-per-CU build; with the keep bits the 32 x 32 x 16 kernel stays.  The dK/dV restatement at the end of
// this file is an A / B form; the default dK/dV kernel is its hand-pipelined sibling, afm_attn_pipe16_impl.h.)
// The dQ kernel of the single-pass flash-attention backward on v_mfma_f32_16x16x32 (included by afm_attn_mfma_impl.h inside namespace
// AFM_E16_NS).  Round 5, VERDICT r04 item 1: the three attention kernels run v_mfma_f32_32x32x16; MI355X_MICROARCH.md ("DVFS give-back"
// item 7) measures the 16x16x32 shape at 1.12-1.15 x the FLOP/s in bare clock-limited loops, so the shape is A/B-tested here on the
// kernel that is simplest to restate: SAME output tile per wave (32 queries x 32-key blocks, 4 waves = 128 queries per workgroup, three
// workgroups per CU), same LDS images and LDS-DMA ring, same arithmetic per score; only the MFMA shape and what follows from it differ:
//   * S^T / dP^T of a 32-key block = 2 x 2 tiles of 16 keys x 16 queries, two k-steps of 32 over dh = 64: 8 + 8 MFMAs (were 4 + 4);
//     a lane owns TWO queries (columns lane & 15 of the two query tiles) and, per tile, keys 4 (lane >> 4) .. + 3;
//   * dS^T as the B operand of dQ^T += K^T dS^T: k-index 8 g + j <-> key 16 (j >> 2) + 4 g + (j & 3) of the block, i.e. the four
//     registers of key tile 0 followed by those of key tile 1 -- no lane movement; the A operand K^T[d][key] takes the same key order
//     from two ds_read_b64_tr_b16 (rows 4 g .. + 3 and 16 + 4 g .. + 3);
//   * the transposed K image needs its own swizzle for that read pattern (chunk ^ ((row >> 1) & 3) << 1: the four same-parity rows of a
//     32-lane half land in four different 32-byte chunk pairs): dma_piece_tr16.
// Dropout keep bits: the tensor's 64-bit words are lane masks of the 32x32 accumulator layout (the forward kernel writes them: word R of
// a 32x32 block, bit 32 h + q = keep(key (R & 3) + 8 (R >> 2) + 4 h, query q)).  The lane mask of register r of the 16x16 tile (ki, qi)
// -- lane 16 g + c <-> key 16 ki + 4 g + r, query 16 qi + c -- is four 16-bit fields of two of those words, Ra = r + 8 ki (g = 0, 1) and
// Rb = Ra + 4 (g = 2, 3): [Ra.lo | Ra.hi | Rb.lo | Rb.hi] taken at field qi, i.e. two scalar 16-bit packs per mask (keep_mask16x16).
// The tensor layout, hence the forward and the dK/dV kernels, stay as they are.

__device__ __forceinline__ void dma_piece_tr16(unsigned char* img, const e16* base, int ld, int row0, int nrows, int pi, int lane) {
  const int r = 8 * pi + (lane >> 3), slot = lane & 7;
  const int chunk = slot ^ (((r >> 1) & 3) << 1);
  int gr = row0 + r;
  gr = gr < nrows ? gr : nrows - 1;   // clamped rows are masked out by the caller
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (int64_t)gr * ld + chunk * 8),
                                   (__attribute__((address_space(3))) void*)(img + pi * 1024), 16, 0, 0);
}
// A-operand fragment by rows for 16x16x32: lane holds tile[R0 + (lane & 15)][32 ks + 8 (lane >> 4) .. + 7] of a row image
__device__ __forceinline__ e16x8 frag_row16(const unsigned char* img, int R0, int ks, int lane) {
  return *(const e16x8*)(img + img_row(R0 + (lane & 15), 4 * ks + (lane >> 4)));
}
__device__ __forceinline__ e16x8 cvt8_2x4(const f32x4& lo, const f32x4& hi) {
  return (e16x8){(e16)lo[0], (e16)lo[1], (e16)lo[2], (e16)lo[3], (e16)hi[0], (e16)hi[1], (e16)hi[2], (e16)hi[3]};
}
// keep ? x : alt for the lane's four consecutive keys key0 .. key0 + 3 of score-matrix row `rowhash` (two pair mixes)
__device__ __forceinline__ void drop_select4(const DropDev& dd, uint32_t rowhash, int key0, f32x4& x, float alt) {
  const uint32_t base = rowhash + afm_pair_offset((uint32_t)key0 >> 1);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const uint32_t hh = afm_pair_mix(base + (uint32_t)p * AFM_PAIR_STRIDE);
    x[2 * p] = (hh & 0xFFFFu) >= dd.thresh16 ? x[2 * p] : alt;
    x[2 * p + 1] = (hh >> 16) >= dd.thresh16 ? x[2 * p + 1] : alt;
  }
}

// lane mask of accumulator register r of the 16 x 16 tile (ki, qi) from the 32 x 32 block's 16 words (see the header)
__device__ __forceinline__ unsigned long long keep_mask16x16(const KeepMasks& m, int ki, int qi, int r) {
  const int Ra = r + 8 * ki, Rb = Ra + 4;
  const u32x16& va = Ra < 8 ? m.a : m.b;
  const u32x16& vb = Rb < 8 ? m.a : m.b;
  const uint32_t alo = va[2 * (Ra & 7)], ahi = va[2 * (Ra & 7) + 1], blo = vb[2 * (Rb & 7)], bhi = vb[2 * (Rb & 7) + 1];
  uint32_t nlo, nhi;      // one scalar pack each (hipcc spells the C form as three to four SALU instructions per half)
  if (qi) {
    asm("s_pack_hh_b32_b16 %0, %1, %2" : "=s"(nlo) : "s"(alo), "s"(ahi));
    asm("s_pack_hh_b32_b16 %0, %1, %2" : "=s"(nhi) : "s"(blo), "s"(bhi));
  } else {
    asm("s_pack_ll_b32_b16 %0, %1, %2" : "=s"(nlo) : "s"(alo), "s"(ahi));
    asm("s_pack_ll_b32_b16 %0, %1, %2" : "=s"(nhi) : "s"(blo), "s"(bhi));
  }
  return ((unsigned long long)nhi << 32) | nlo;
}

template <int DROP, int OCC = 3>
__global__ __launch_bounds__(256, OCC) void k_attn_bwd_dq_m16(AttnM a, const e16* __restrict__ Q,
                                                         const e16* __restrict__ K,
                                                         const e16* __restrict__ V,
                                                         const e16* __restrict__ O,
                                                         const e16* __restrict__ dO,
                                                         const float* __restrict__ lse,
                                                         float* __restrict__ delta, e16* __restrict__ dQ) {
  constexpr int STAGE = 3 * KT * DH * 2;   // K row image, K tr image, V row image
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned long long* maskw = (unsigned long long*)(lds + RS * STAGE);
  const int t = threadIdx.x, lane = t & 63, g = lane >> 4, c16 = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tq + 127) / 128);
  const int hd = blk_.hd, b = blk_.b;
  const int q0 = blk_.xb * 128 + w * 32;
  const int64_t rq = attn_row0(a.q_off, b, a.Tq), rk = (int64_t)__builtin_amdgcn_readfirstlane((int)attn_row0(a.k_off, b, a.Tk));      // (scalar tile bases: dma_piece_s)
  const int lim_q = attn_slot(a.q_off, b, a.Tq);      // rows of this sample that are its own (packed: its slot)
  {
    int64_t tail0;
    if (attn_tail_block(a.q_off, a.B, b, blk_.xb, a.Tq, tail0)) {      // packed rows, a block beyond the sample's slot: zeros to its block of the dead tail
      const int64_t fe = attn_fill_end(a.nofill, a.q_off, a.B);
      const e16x4 z = {(e16)0.f, (e16)0.f, (e16)0.f, (e16)0.f};
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {
        const int64_t trow = tail0 + w * 32 + 16 * qi + c16;
        if (trow >= fe) continue;
        e16* dqp = dQ + trow * a.lddq + hd * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(e16x4*)(dqp + 16 * dt) = z;
      }
      return;
    }
  }
  const e16* Kb = K + rk * a.ldk + hd * DH;
  const e16* Vb = V + rk * a.ldv + hd * DH;
  int q[2], qc[2];
  int64_t lrow[2];
  e16x8 qf[2][2], dof[2][2];
  float nL2[2], ndl[2];
  uint32_t rowbase[2];
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    q[qi] = q0 + 16 * qi + c16;
    qc[qi] = q[qi] < a.Tq ? q[qi] : a.Tq - 1;
    const e16* qp = Q + (rq + qc[qi]) * a.ldq + hd * DH + 8 * g;
    const e16* dop = dO + (rq + qc[qi]) * a.ldo + hd * DH + 8 * g;
    const e16* op = O + (rq + qc[qi]) * a.ldo + hd * DH + 8 * g;
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[qi][ks] = ld8_once(qp + 32 * ks);
      dof[qi][ks] = ld8_once(dop + 32 * ks);
      const e16x8 ov = ld8_once(op + 32 * ks);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += (float)dof[qi][ks][j] * (float)ov[j];
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    lrow[qi] = ((int64_t)b * a.H + hd) * a.Tq + qc[qi];
    if (q[qi] < a.Tq && g == 0) delta[lrow[qi]] = q[qi] < lim_q ? -dl : 0.f;      // the workspace holds -delta (the dK/dV kernel starts its dP accumulators from it; rows beyond a packed slot: 0)
    const float L = lse[lrow[qi]];
    nL2[qi] = L == INFINITY ? -INFINITY : -L * 1.4426950408889634f;
    ndl[qi] = -dl;
    rowbase[qi] = afm_row_hash(a.dd, (uint64_t)lrow[qi]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        qf[qi][ks][j] = (e16)((float)qf[qi][ks][j] * a.scale_log2);
        if (DROP != DROP_NONE) dof[qi][ks][j] = (e16)((float)dof[qi][ks][j] * a.dd.scale16);
      }
  }
  f32x4 dq[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) dq[dt][qi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int kend = a.Tk;
  if (a.causal) kend = min(a.Tk, blk_.xb * 128 + 128);
  const int ntiles = (kend + KT - 1) / KT;
  build_mask_words(maskw, a.key_pad, b, a.Tk, ntiles, w, lane);
  bool wave_qskip = false;
  if (a.qskip) {                       // (the caller vouches for key_pad and for zero dO at padded query rows)
    bool lane_pad = true;
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) lane_pad = lane_pad && (q[qi] >= a.Tq || a.key_pad[(int64_t)b * a.Tk + qc[qi]] != 0);
    wave_qskip = __all(lane_pad);
  }
  int* const tl = (int*)(maskw + (a.Tk + KT - 1) / KT) + 1;      // key tiles with at least one real key
  auto store_dq = [&](bool zeros) {
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      // packed rows: a wave whose rows lie beyond the slot (a partly used last block) zeroes its rows of the dead tail instead
      const bool own = q[qi] < lim_q;
      const int64_t drow = attn_out_row(a.q_off, a.B, b, a.Tq, rq, q[qi], lim_q);
      if (drow >= 0 && (own || drow < attn_fill_end(a.nofill, a.q_off, a.B))) {
        e16* dqp = dQ + drow * a.lddq + hd * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          e16x4 v = {(e16)0.f, (e16)0.f, (e16)0.f, (e16)0.f};
          if (!zeros && own) v = (e16x4){(e16)(dq[dt][qi][0] * a.scale), (e16)(dq[dt][qi][1] * a.scale), (e16)(dq[dt][qi][2] * a.scale), (e16)(dq[dt][qi][3] * a.scale)};
          *(e16x4*)(dqp + 16 * dt) = v;
        }
      }
    }
  };
  if (__syncthreads_and(wave_qskip)) {   // all 128 queries of the workgroup are padding: their dQ rows are zeros, nothing to load
    store_dq(true);
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
      dma_piece_s<0>(st, Kb, a.ldk, kt * KT, a.Tk, w + 4 * u, lane);
      dma_piece_s<3>(st + KT * DH * 2, Kb, a.ldk, kt * KT, a.Tk, w + 4 * u, lane);
      dma_piece_s<0>(st + 2 * KT * DH * 2, Vb, a.ldv, kt * KT, a.Tk, w + 4 * u, lane);
    }
  };
#pragma unroll
  for (int s = 0; s < RS - 1; ++s)
    if (s < nlive) issue(s);
  __builtin_assume(nlive >= 1);
  // transposed reads of the K image: lane (g, qq, p) supplies row 4 g + qq (+ the block's first row, + 16 for the second read), columns
  // 16 dt + 4 p .. + 3 of d-tile dt: chunk 2 dt + (p >> 1) at position chunk ^ sw, sw = (2 (g & 1) + (qq >> 1)) << 1, i.e. the chunk PAIR
  // dt ^ (sw >> 1): one lane address per d-tile, the rows as immediates
  // (one lane address, d-tile dt = an XOR of dt << 5 on it: the row term has no bits below 128, the chunk-pair term is (dt ^ s2) << 5)
  unsigned tra0;
  {
    const int qq = (lane >> 2) & 3, p = lane & 3;
    const int s2 = 2 * (g & 1) + (qq >> 1);
    tra0 = (4 * g + qq) * 128 + ((2 * s2 + (p >> 1)) << 4) + ((p & 1) << 3);
  }
  for (int j = 0; j < nlive; ++j) {
    const int kt = __builtin_amdgcn_readfirstlane(tl[j]);
    const int kb = kt * KT;
    if (nlive - 1 - j >= RS - 2) attn_wait_vmcnt<6 * (RS - 2)>(); else attn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (j + RS - 1 < nlive) issue(j + RS - 1);
    if ((a.causal && kb > q0 + 31) || maskw[kt] == ~0ull || wave_qskip) continue;   // above the diagonal / all-padding key tile / all-padding queries
    const unsigned char* Krow = lds + (j % RS) * STAGE;
    const unsigned ktr = (unsigned)(uintptr_t)(Krow + KT * DH * 2);
    const unsigned char* Vrow = Krow + 2 * KT * DH * 2;
    const unsigned long long mword = maskw[kt];
    KeepMasks km[2];
    if (DROP == DROP_BITS) {   // both 32-key blocks of the tile now; used after the S / dP products
      const unsigned long long* kbp = bits_block(a, b * a.H + hd, q0 >> 5, 2 * kt);
      keep_masks_issue(km[0], kbp);
      keep_masks_issue(km[1], kbp + 16);
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      f32x4 s[2][2], dp[2][2];
#pragma unroll
      for (int ki = 0; ki < 2; ++ki)
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
          s[ki][qi] = (f32x4){nL2[qi], nL2[qi], nL2[qi], nL2[qi]};
          dp[ki][qi] = (f32x4){ndl[qi], ndl[qi], ndl[qi], ndl[qi]};
        }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) {
          const e16x8 kfr = frag_row16(Krow, 32 * blk + 16 * ki, ks, lane);
          const e16x8 vfr = frag_row16(Vrow, 32 * blk + 16 * ki, ks, lane);
#pragma unroll
          for (int qi = 0; qi < 2; ++qi) {
            s[ki][qi] = mfma16(kfr, qf[qi][ks], s[ki][qi]);
            dp[ki][qi] = mfma16(vfr, dof[qi][ks], dp[ki][qi]);
          }
        }
      if (DROP == DROP_HASH) {
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
#pragma unroll
          for (int qi = 0; qi < 2; ++qi) drop_select4(a.dd, rowbase[qi], kb + 32 * blk + 16 * ki + 4 * g, dp[ki][qi], ndl[qi]);
      }
      if (DROP == DROP_BITS) {
        keep_masks_wait(km[blk]);
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
#pragma unroll
          for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              dp[ki][qi][r] = __builtin_amdgcn_inverse_ballot_w64(keep_mask16x16(km[blk], ki, qi, r)) ? dp[ki][qi][r] : ndl[qi];
      }
      if (mword != 0ull || (a.causal && (kb + KT - 1 > q0))) {   // wave-uniform: tile has masked keys
        const unsigned long long padg = mword >> (4 * g);          // bit 32 blk + 16 ki + r = this lane's key of register r, tile ki
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kc = 32 * blk + 16 * ki + r;                 // + 4 g = the key inside the 64-key tile
            const bool pad = (padg >> kc) & 1ull;
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) {
              bool msk = pad;
              if (a.causal) msk = msk || (kb + kc + 4 * g > q[qi]);
              s[ki][qi][r] = msk ? -INFINITY : s[ki][qi][r];
            }
          }
      }
#pragma unroll
      for (int ki = 0; ki < 2; ++ki)
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[ki][qi][r] = fast_exp2(s[ki][qi][r]) * dp[ki][qi][r];   // dS^T = P (D dP - delta)
      {
        s16x4 lo[4], hi[4];
        unsigned tra = tra0;
        asm volatile("" : "+v"(tra));      // opaque: tra0 ^ (dt << 5) is one instruction here; hoisted out of the tile loop the four of them were spilled (and reloaded behind vmcnt(0))
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const unsigned ad = ktr + (tra ^ (dt << 5));
          if (blk == 0) { AFM_TR_RD(lo[dt], ad, 0); AFM_TR_RD(hi[dt], ad, 2048); }
          else { AFM_TR_RD(lo[dt], ad, 4096); AFM_TR_RD(hi[dt], ad, 6144); }
        }
        const e16x8 ds0 = cvt8_2x4(s[0][0], s[1][0]), ds1 = cvt8_2x4(s[0][1], s[1][1]);
        tr_wait<4>();
        dq[0][0] = mfma16(tr_join(lo[0], hi[0]), ds0, dq[0][0]);
        dq[0][1] = mfma16(tr_join(lo[0], hi[0]), ds1, dq[0][1]);
        dq[1][0] = mfma16(tr_join(lo[1], hi[1]), ds0, dq[1][0]);
        dq[1][1] = mfma16(tr_join(lo[1], hi[1]), ds1, dq[1][1]);
        tr_wait<0>();
        dq[2][0] = mfma16(tr_join(lo[2], hi[2]), ds0, dq[2][0]);
        dq[2][1] = mfma16(tr_join(lo[2], hi[2]), ds1, dq[2][1]);
        dq[3][0] = mfma16(tr_join(lo[3], hi[3]), ds0, dq[3][0]);
        dq[3][1] = mfma16(tr_join(lo[3], hi[3]), ds1, dq[3][1]);
      }
    }
  }
  store_dq(false);
}

// ------------------------------------------------------------------------------------------ dK, dV on 16x16x32
// k_attn_bwd_dkv_mfma (the round-3 kernel: workgroup = 4 waves x 32 keys, key on the lane, loops over 64-query tiles) restated on
// v_mfma_f32_16x16x32 at the same output tile per wave.  For THIS kernel the keep-bit tensor needs no re-assembly: the 32 x 32 layout
// keeps the 32 query bits of one key in one dword (the round-3 kernel reads exactly that), and a lane of a 16 x 16 tile -- key 16 ki + c,
// queries 16 qi + 4 g + r -- tests bits 16 qi + 4 g + r of its key's dword.
//   * K / V fragments of the wave's two 16-key tiles stay in registers (B operands of S = Q K^T and dP = dO V^T);
//   * S / dP of a 32-query block = 2 x 2 tiles, two k-steps of 32 over dh: 8 + 8 MFMAs; the accumulators start from -lse[q] / -delta[q]
//     (one 16-byte LDS read per query tile: the lane's four queries are consecutive);
//   * P~ and dS as B operands of dV^T += dO^T P~ and dK^T += Q^T dS: k-index 8 g + j <-> query 16 (j >> 2) + 4 g + (j & 3) of the
//     block -- the four registers of query tile 0, then those of tile 1; the A operands take that order from two transposed reads each;
//   * ONE image per tile for Q and dO, read by rows (S, dP) AND transposed (dK, dV): the swizzle chunk ^ ((row >> 1) & 3) << 1 is
//     conflict-free for both patterns of the 16 x 16 x 32 shape (row reads: the lane groups of ds_read_b128 see rows {0-3, 12-15} at
//     chunk c and {4-11} at c ^ 1, i.e. eight different (row parity, position) slots per parity; transposed reads: the four same-parity
//     rows of a 32-lane half land in four different chunk pairs), so the images and the LDS-DMA are those of the round-3 kernel.
// DROP_NONE and DROP_BITS; selected by afm_attn_shape.reserved & 4096 (an A/B form, never the default).
__device__ __forceinline__ e16x8 frag_row16_d(const unsigned char* img, int R0, int ks, int lane) {      // row fragment of a tr16-swizzled image
  const int c16 = lane & 15;
  return *(const e16x8*)(img + (R0 + c16) * 128 + (((4 * ks + (lane >> 4)) ^ (((c16 >> 1) & 3) << 1)) << 4));
}

template <int DROP>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dkv_m16(AttnM a, const e16* __restrict__ Q,
                                                          const e16* __restrict__ K,
                                                          const e16* __restrict__ V,
                                                          const e16* __restrict__ dO,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ delta,
                                                          e16* __restrict__ dK, e16* __restrict__ dV) {
  static_assert(DROP == DROP_NONE || DROP == DROP_BITS, "the re-hash path stays with the round-3 kernel");
  constexpr int IMG = KT * DH * 2;
  constexpr int STAGE = 2 * IMG + 2 * KT * 4 + 4 * 256;   // Q, dO images; lse, -delta; two 128-byte keep-bit blocks per wave
  constexpr int DS = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, g = lane >> 4, c16 = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tk + 127) / 128);
  const int hd = blk_.hd, b = blk_.b;
  const int k0 = blk_.xb * 128 + w * 32;
  const e16* Qb = Q + (int64_t)b * a.Tq * a.ldq + hd * DH;
  const e16* Db = dO + (int64_t)b * a.Tq * a.ldo + hd * DH;
  int key[2];
  bool kmasked[2];
  e16x8 kf[2][2], vf[2][2];
  bool wave_all_masked = true;
#pragma unroll
  for (int ki = 0; ki < 2; ++ki) {
    key[ki] = k0 + 16 * ki + c16;
    const int kc = key[ki] < a.Tk ? key[ki] : a.Tk - 1;
    kmasked[ki] = key[ki] >= a.Tk || (a.key_pad && a.key_pad[(int64_t)b * a.Tk + kc]);
    wave_all_masked = wave_all_masked && __all(kmasked[ki]);
    const e16* kp = K + ((int64_t)b * a.Tk + kc) * a.ldk + hd * DH + 8 * g;
    const e16* vp = V + ((int64_t)b * a.Tk + kc) * a.ldv + hd * DH + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ki][ks] = ld8_once(kp + 32 * ks); vf[ki][ks] = ld8_once(vp + 32 * ks);
#pragma unroll
      for (int j = 0; j < 8; ++j) {      // K by scale * log2(e) (p = exp2(S') is one instruction), V by the dropout scale
        kf[ki][ks][j] = (e16)((float)kf[ki][ks][j] * a.scale_log2);
        if (DROP != DROP_NONE) vf[ki][ks][j] = (e16)((float)vf[ki][ks][j] * a.dd.scale16);
      }
    }
  }
  f32x4 dk[4][2], dv[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) { dk[dt][ki] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt][ki] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  int qbeg = 0;
  if (a.causal) qbeg = (blk_.xb * 128) / KT * KT;   // queries before the block's first key see none of it
  const int ntiles = (a.Tq - qbeg + KT - 1) / KT;
  const int64_t lbase = ((int64_t)b * a.H + hd) * a.Tq;
  unsigned long long* qmaskw = (unsigned long long*)(lds + DS * STAGE);
  int* const tl = (int*)(qmaskw + (a.Tq + KT - 1) / KT) + 1;
  if (a.qskip) build_mask_words(qmaskw, a.key_pad, b, a.Tq, (a.Tq + KT - 1) / KT, w, lane);
  auto store = [&](bool zeros) {
#pragma unroll
    for (int ki = 0; ki < 2; ++ki)
      if (key[ki] < a.Tk) {
        e16* dkp = dK + ((int64_t)b * a.Tk + key[ki]) * a.lddk + hd * DH + 4 * g;
        e16* dvp = dV + ((int64_t)b * a.Tk + key[ki]) * a.lddv + hd * DH + 4 * g;
        const bool z = zeros || kmasked[ki];      // a padded key took no part in any softmax: its dK / dV rows are zero
        const float sv = DROP != DROP_NONE ? a.dd.scale16 : 1.0f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          e16x4 x = {(e16)0.f, (e16)0.f, (e16)0.f, (e16)0.f}, y = x;
          if (!z) {
            x = (e16x4){(e16)(dk[dt][ki][0] * a.scale), (e16)(dk[dt][ki][1] * a.scale), (e16)(dk[dt][ki][2] * a.scale), (e16)(dk[dt][ki][3] * a.scale)};
            y = (e16x4){(e16)(dv[dt][ki][0] * sv), (e16)(dv[dt][ki][1] * sv), (e16)(dv[dt][ki][2] * sv), (e16)(dv[dt][ki][3] * sv)};
          }
          *(e16x4*)(dkp + 16 * dt) = x;
          *(e16x4*)(dvp + 16 * dt) = y;
        }
      }
  };
  if (__syncthreads_and(wave_all_masked)) {   // 128 padded keys: zeros, nothing to load
    store(true);
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
      dma_piece_tr16(st, Qb, a.ldq, row0, a.Tq, w + 4 * u, lane);
      dma_piece_tr16(st + IMG, Db, a.ldo, row0, a.Tq, w + 4 * u, lane);
    }
    if (w < 2) {   // lse / -delta of the tile's 64 queries: one 4-byte piece each
      int qq = row0 + lane;
      qq = qq < a.Tq ? qq : a.Tq - 1;
      const float* src = (w == 0 ? lse : delta) + lbase + qq;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + 2 * IMG + w * KT * 4), 4, 0, 0);
    }
    if (DROP == DROP_BITS) {   // keep-bit blocks (query block of lanes 0-31 / 32-63, this wave's key block): 2 x 32 dwords, as the round-3 kernel
      const uint32_t* src = (const uint32_t*)bits_block(a, b * a.H + hd, row0 >> 5, min(k0 >> 5, a.nk32 - 1)) + (lane >> 5) * (a.nk32 * 32) + (lane & 31);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + 2 * IMG + 2 * KT * 4 + w * 256), 4, 0, 0);
    }
  };
  // transposed reads (as the dQ form above): one lane address, d-tile dt = an XOR of dt << 5; rows as immediates
  unsigned tra0;
  {
    const int qq = (lane >> 2) & 3, p = lane & 3;
    const int s2 = 2 * (g & 1) + (qq >> 1);
    tra0 = (4 * g + qq) * 128 + ((2 * s2 + (p >> 1)) << 4) + ((p & 1) << 3);
  }
  issue(0);
  __builtin_assume(nlive >= 1);
  for (int j = 0; j < nlive; ++j) {
    const int qb = __builtin_amdgcn_readfirstlane(tl[j]) * KT;
    attn_wait_vmcnt<0>();          // this tile's pieces (the only ones in flight)
    __builtin_amdgcn_s_barrier();
    if (j + 1 < nlive) issue(j + 1);
    const unsigned char* Qrow = lds + (j % DS) * STAGE;
    const unsigned char* Drow = Qrow + IMG;
    const float* Ls = (const float*)(Qrow + 2 * IMG);   // lse (natural log units)
    const float* Ds = Ls + KT;                          // -delta
    const bool ragged = qb + KT > a.Tq;
    if ((a.causal && qb + KT - 1 < k0) || wave_all_masked || (a.qskip && qmaskw[qb / KT] == ~0ull)) continue;
    const unsigned qtr = (unsigned)(uintptr_t)Qrow, dtr = (unsigned)(uintptr_t)Drow;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      f32x4 s[2][2], dp[2][2];      // [query tile][key tile]
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {
        const f32x4 Lq = *(const f32x4*)(Ls + 32 * blk + 16 * qi + 4 * g) * -1.4426950408889634f;
        const f32x4 Dq = *(const f32x4*)(Ds + 32 * blk + 16 * qi + 4 * g);
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) { s[qi][ki] = Lq; dp[qi][ki] = Dq; }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
          const e16x8 qfr = frag_row16_d(Qrow, 32 * blk + 16 * qi, ks, lane);
          const e16x8 dfr = frag_row16_d(Drow, 32 * blk + 16 * qi, ks, lane);
#pragma unroll
          for (int ki = 0; ki < 2; ++ki) {
            s[qi][ki] = mfma16(qfr, kf[ki][ks], s[qi][ki]);       // S'[q][key] = S log2(e) / sqrt(dh) - lse[q]
            dp[qi][ki] = mfma16(dfr, vf[ki][ks], dp[qi][ki]);     // scale dP[q][key] - delta[q]
          }
        }
      if (a.causal || ragged) {   // rare: diagonal tiles of the decoder / the last, partly filled tile
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qq = qb + 32 * blk + 16 * qi + 4 * g + r;
#pragma unroll
            for (int ki = 0; ki < 2; ++ki) {
              const bool msk = (a.causal && key[ki] > qq) || qq >= a.Tq;
              s[qi][ki][r] = msk ? -INFINITY : s[qi][ki][r];
            }
          }
      }
      f32x4 pd[2][2];
#pragma unroll
      for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
#pragma unroll
          for (int r = 0; r < 4; ++r) pd[qi][ki][r] = fast_exp2(s[qi][ki][r]);
      if (DROP == DROP_BITS) {
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) {
          // this lane's key of tile ki is key 16 ki + c16 of the wave's 32-key block: its dword holds the 32 query bits of the block
          const uint32_t word = ((const uint32_t*)(Qrow + 2 * IMG + 2 * KT * 4 + w * 256))[32 * blk + bits_word_of_key(16 * ki + c16)] >> (4 * g);
#pragma unroll
          for (int qi = 0; qi < 2; ++qi) {
            const f32x4 nd = *(const f32x4*)(Ds + 32 * blk + 16 * qi + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool kp = (int)(word << (31 - (16 * qi + r))) < 0;
              s[qi][ki][r] = pd[qi][ki][r] * (kp ? dp[qi][ki][r] : nd[r]);      // dS = P (D dP - delta)
              pd[qi][ki][r] = kp ? pd[qi][ki][r] : 0.f;                           // dropped P for dV
            }
          }
        }
      } else {
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
          for (int ki = 0; ki < 2; ++ki)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[qi][ki][r] = pd[qi][ki][r] * dp[qi][ki][r];
      }
      {
        const e16x8 pf0 = cvt8_2x4(pd[0][0], pd[1][0]), pf1 = cvt8_2x4(pd[0][1], pd[1][1]);      // key tile 0 / 1: [query tile 0 regs | query tile 1 regs]
        const e16x8 sf0 = cvt8_2x4(s[0][0], s[1][0]), sf1 = cvt8_2x4(s[0][1], s[1][1]);
        s16x4 dlo[4], dhi[4], qlo[4], qhi[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const unsigned ad = dtr + (tra0 ^ (dt << 5));
          if (blk == 0) { AFM_TR_RD(dlo[dt], ad, 0); AFM_TR_RD(dhi[dt], ad, 2048); }
          else { AFM_TR_RD(dlo[dt], ad, 4096); AFM_TR_RD(dhi[dt], ad, 6144); }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const unsigned aq = qtr + (tra0 ^ (dt << 5));
          if (blk == 0) { AFM_TR_RD(qlo[dt], aq, 0); AFM_TR_RD(qhi[dt], aq, 2048); }
          else { AFM_TR_RD(qlo[dt], aq, 4096); AFM_TR_RD(qhi[dt], aq, 6144); }
        }
        tr_wait<8>();
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const e16x8 af = tr_join(dlo[dt], dhi[dt]);
          dv[dt][0] = mfma16(af, pf0, dv[dt][0]);
          dv[dt][1] = mfma16(af, pf1, dv[dt][1]);
        }
        tr_wait<0>();
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const e16x8 af = tr_join(qlo[dt], qhi[dt]);
          dk[dt][0] = mfma16(af, sf0, dk[dt][0]);
          dk[dt][1] = mfma16(af, sf1, dk[dt][1]);
        }
      }
    }
  }
  store(false);
}
