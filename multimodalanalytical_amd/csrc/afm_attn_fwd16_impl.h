// The single-pass flash-attention FORWARD on v_mfma_f32_16x16x32 (included by afm_attn_mfma_impl.h inside namespace AFM_E16_NS, after
// afm_attn_m16_impl.h whose helpers it uses).  Round 5, VERDICT r04 item 1: same geometry as k_attn_fwd_mfma -- workgroup = 4 waves x 32
// queries of one (batch, head), 64-key tiles through the same two-stage LDS-DMA ring, same arithmetic per score (scores in log2 units
// from pre-scaled Q fragments, lazy running maximum, exp2, fp32 row sums, dropout after the sum) -- restated on the other MFMA shape:
//   * S^T of a 64-key tile = 4 x 2 tiles of 16 keys x 16 queries, two k-steps of 32 over dh: 16 MFMAs (were 8); a lane owns TWO queries
//     (column lane & 15 of the two query tiles) and, per key tile, keys 4 (lane >> 4) .. + 3;
//   * P^T as the B operand of O^T += V^T P^T: k-index 8 g + j <-> key 16 (j >> 2) + 4 g + (j & 3) of a 32-key block, i.e. the four
//     registers of key tile 2 kb followed by those of key tile 2 kb + 1 -- no lane movement; the A operand V^T[d][key] takes the same key
//     order from two ds_read_b64_tr_b16 (rows 4 g .. + 3 and 16 + 4 g .. + 3) of the V image in the dQ kernel's transposed-read swizzle
//     (dma_piece_tr16);
//   * the keys of one query are spread over the four lane groups g, so the row maximum would cost two lane exchanges per query and tile;
//     the lazy-maximum rule makes them unnecessary: a tile only looks at its lane-LOCAL maxima, and only when some lane's exceeds 2^8 (or a
//     row is still unset) does the wave exchange and move m -- per query, identically in its four lanes, which is what the MFMA's sum over
//     the k-index needs;
//   * dropout keep bits (DROP_BITS: the forward writes the tensor): the tensor's words are lane masks of the 32 x 32 layout
//     (afm_attn_m16_impl.h header); the two ballots of a (key tile, register) pair -- one per query tile -- give words Ra = r + 8 ki' and
//     Rb = Ra + 4 of the block through four scalar 16-bit packs, stored through the scalar path as before.
// Selected by afm_attn_shape.reserved & 1024 (A / B form; & 2048: compiled for three workgroups per CU instead of four).

template <int KI2, int P>
__device__ __forceinline__ void fwd16_emit_pair(const DropDev& dd, const uint32_t (&base)[2], f32x4 (&s0)[2], unsigned long long* blk) {
  // keys j = 2 P, 2 P + 1 of key tile ki' = KI2 (of the 32-key block `blk`), both query tiles: s0[qi] is that tile's score register set
  const uint32_t h0 = afm_pair_mix(base[0] + (uint32_t)P * AFM_PAIR_STRIDE), h1 = afm_pair_mix(base[1] + (uint32_t)P * AFM_PAIR_STRIDE);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const bool k0 = (e ? (h0 >> 16) : (h0 & 0xFFFFu)) >= dd.thresh16, k1 = (e ? (h1 >> 16) : (h1 & 0xFFFFu)) >= dd.thresh16;
    const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
    s0[0][2 * P + e] = k0 ? s0[0][2 * P + e] : 0.f;
    s0[1][2 * P + e] = k1 ? s0[1][2 * P + e] : 0.f;
    const uint32_t m0l = (uint32_t)m0, m0h = (uint32_t)(m0 >> 32), m1l = (uint32_t)m1, m1h = (uint32_t)(m1 >> 32);
    uint32_t al, ah, bl, bh;
    asm("s_pack_ll_b32_b16 %0, %1, %2" : "=s"(al) : "s"(m0l), "s"(m1l));
    asm("s_pack_hh_b32_b16 %0, %1, %2" : "=s"(ah) : "s"(m0l), "s"(m1l));
    asm("s_pack_ll_b32_b16 %0, %1, %2" : "=s"(bl) : "s"(m0h), "s"(m1h));
    asm("s_pack_hh_b32_b16 %0, %1, %2" : "=s"(bh) : "s"(m0h), "s"(m1h));
    const unsigned long long wa = ((unsigned long long)ah << 32) | al, wb = ((unsigned long long)bh << 32) | bl;
    constexpr int RA = 8 * KI2 + 2 * P;
    if (e == 0) {
      asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(wa), "s"(blk), "n"(RA * 8) : "memory");
      asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(wb), "s"(blk), "n"((RA + 4) * 8) : "memory");
    } else {
      asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(wa), "s"(blk), "n"((RA + 1) * 8) : "memory");
      asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(wb), "s"(blk), "n"((RA + 5) * 8) : "memory");
    }
  }
}

template <int DROP, int OCC>
__global__ __launch_bounds__(256, OCC) void k_attn_fwd_m16(AttnM a, const e16* __restrict__ Q, const e16* __restrict__ K,
                                                           const e16* __restrict__ V, e16* __restrict__ O, float* __restrict__ lse) {
  constexpr int STAGE = 2 * KT * DH * 2;   // K row image + V image (transposed-read swizzle of the 16 x 16 x 32 shape)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned long long* maskw = (unsigned long long*)(lds + RS * STAGE);
  const int t = threadIdx.x, lane = t & 63, g = lane >> 4, c16 = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tq + 127) / 128);
  const int hd = blk_.hd, b = blk_.b;
  const int q0 = blk_.xb * 128 + w * 32;
  const e16* Kb = K + (int64_t)b * a.Tk * a.ldk + hd * DH;
  const e16* Vb = V + (int64_t)b * a.Tk * a.ldv + hd * DH;
  int kend = a.Tk;
  if (a.causal) kend = min(a.Tk, blk_.xb * 128 + 128);
  const int ntiles = (kend + KT - 1) / KT;
  int q[2], qc[2];
  e16x8 qf[2][2];
  uint32_t rowbase[2];
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    q[qi] = q0 + 16 * qi + c16;
    qc[qi] = q[qi] < a.Tq ? q[qi] : a.Tq - 1;
    const e16* qp = Q + ((int64_t)b * a.Tq + qc[qi]) * a.ldq + hd * DH + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const e16x8 x = ld8_once(qp + 32 * ks);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[qi][ks][j] = (e16)((float)x[j] * a.scale_log2);
    }
    rowbase[qi] = afm_row_hash(a.dd, (uint64_t)(b * a.H + hd) * a.Tq + qc[qi]);
  }
  build_mask_words(maskw, a.key_pad, b, a.Tk, ntiles, w, lane);
  int* const tl = (int*)(maskw + (a.Tk + KT - 1) / KT) + 1;
  __syncthreads();
  build_tile_list(tl, a.key_pad ? maskw : nullptr, 0, ntiles, w, lane);
  __syncthreads();   // plain loads above are retired here, before any LDS-DMA is in flight
  const int nlive = __builtin_amdgcn_readfirstlane(tl[-1]);
  auto issue = [&](int j) {
    unsigned char* st = lds + (j % RS) * STAGE;
    const int kt = tl[j];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      dma_piece<false>(st, Kb, a.ldk, kt * KT, a.Tk, w + 4 * u, lane);
      dma_piece_tr16(st + KT * DH * 2, Vb, a.ldv, kt * KT, a.Tk, w + 4 * u, lane);
    }
  };
  f32x4 o[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) o[dt][qi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
#pragma unroll
  for (int s = 0; s < RS - 1; ++s)
    if (s < nlive) issue(s);
  __builtin_assume(nlive >= 1);
  unsigned tra0;      // transposed reads of the V image: as the dQ kernel's K^T reads (afm_attn_m16_impl.h)
  {
    const int qq = (lane >> 2) & 3, p = lane & 3;
    const int s2 = 2 * (g & 1) + (qq >> 1);
    tra0 = (4 * g + qq) * 128 + ((2 * s2 + (p >> 1)) << 4) + ((p & 1) << 3);
  }
  for (int j = 0; j < nlive; ++j) {
    const int kt = __builtin_amdgcn_readfirstlane(tl[j]);
    const int kb = kt * KT;
    if (nlive - 1 - j >= RS - 2) attn_wait_vmcnt<4 * (RS - 2)>(); else attn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (j + RS - 1 < nlive) issue(j + RS - 1);
    if ((a.causal && kb > q0 + 31) || maskw[kt] == ~0ull) continue;
    const unsigned char* Kimg = lds + (j % RS) * STAGE;
    const unsigned vtr = (unsigned)(uintptr_t)(Kimg + KT * DH * 2);
    const unsigned long long mword = maskw[kt];
    KeepMasks km[2];
    if (DROP == DROP_READ) {
      const unsigned long long* bb = bits_block(a, b * a.H + hd, q0 >> 5, kb >> 5);
      keep_masks_issue(km[0], bb);
      keep_masks_issue(km[1], bb + 16);
    }
    f32x4 s[4][2];
    float init[2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) init[qi] = m[qi] == -INFINITY ? 0.f : -m[qi];
#pragma unroll
    for (int ki = 0; ki < 4; ++ki)
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) s[ki][qi] = (f32x4){init[qi], init[qi], init[qi], init[qi]};
#pragma unroll
    for (int ki = 0; ki < 4; ++ki)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const e16x8 kfr = frag_row16(Kimg, 16 * ki, ks, lane);
        s[ki][0] = mfma16(kfr, qf[0][ks], s[ki][0]);
        s[ki][1] = mfma16(kfr, qf[1][ks], s[ki][1]);
      }
    if (mword != 0ull || (a.causal && (kb + KT - 1 > q0))) {   // wave-uniform: the tile has masked keys
      const unsigned long long padg = mword >> (4 * g);          // bit 16 ki + r = this lane's key of register r, key tile ki
#pragma unroll
      for (int ki = 0; ki < 4; ++ki)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kc = 16 * ki + r;
          const bool pad = (padg >> kc) & 1ull;
#pragma unroll
          for (int qi = 0; qi < 2; ++qi) {
            bool msk = pad;
            if (a.causal) msk = msk || (kb + kc + 4 * g > q[qi]);
            s[ki][qi][r] = msk ? -INFINITY : s[ki][qi][r];
          }
        }
    }
    // lane-local maxima of S' = S - m (v_max3 from asm: see k_attn_fwd_mfma); the exchange only in the rare tile that moves m
    float mt[2];
    bool move = false;
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      float x = fmaxf(s[0][qi][0], s[0][qi][1]);
      x = max3_raw(x, s[0][qi][2], s[0][qi][3]);
#pragma unroll
      for (int ki = 1; ki < 4; ++ki) { x = max3_raw(x, s[ki][qi][0], s[ki][qi][1]); x = max3_raw(x, s[ki][qi][2], s[ki][qi][3]); }
      mt[qi] = x;
      move = move || (m[qi] == -INFINITY && x != -INFINITY) || x > 8.f;
    }
    if (__any(move)) {
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {
        float x = mt[qi];
        x = fmaxf(x, __shfl_xor(x, 16, 64));
        x = fmaxf(x, __shfl_xor(x, 32, 64));           // the query's maximum over the tile's 64 keys: equal in its four lanes
        const bool unset = m[qi] == -INFINITY;          // (m is per query, so `unset` is too)
        const float dlt = unset ? (x == -INFINITY ? 0.f : x) : fmaxf(x, 0.f);
        const float alpha = unset ? 1.f : fast_exp2(-dlt);
        m[qi] = (unset && x == -INFINITY) ? m[qi] : (unset ? dlt : m[qi] + dlt);
        l[qi] *= alpha;
        // o[dt][qi] holds O^T[d][query lane & 15]: the column's query is THIS lane's query qi
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[dt][qi][r] *= alpha;
#pragma unroll
        for (int ki = 0; ki < 4; ++ki)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[ki][qi][r] -= dlt;
      }
    }
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      float ls = 0.f;
#pragma unroll
      for (int ki = 0; ki < 4; ++ki)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fast_exp2(s[ki][qi][r]);
          s[ki][qi][r] = p;
          ls += p;
        }
      l[qi] += ls;
    }
    if (DROP == DROP_HASH) {
#pragma unroll
      for (int ki = 0; ki < 4; ++ki)
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) drop_select4(a.dd, rowbase[qi], kb + 16 * ki + 4 * g, s[ki][qi], 0.f);
    }
    if (DROP == DROP_BITS) {
      unsigned long long* bb = bits_block(a, b * a.H + hd, q0 >> 5, kb >> 5);
#pragma unroll
      for (int ki = 0; ki < 4; ++ki) {
        const uint32_t off = afm_pair_offset((uint32_t)(kb + 16 * ki + 4 * g) >> 1);
        const uint32_t base[2] = {rowbase[0] + off, rowbase[1] + off};
        unsigned long long* blk = bb + 16 * (ki >> 1);
        if (ki & 1) { fwd16_emit_pair<1, 0>(a.dd, base, s[ki], blk); fwd16_emit_pair<1, 1>(a.dd, base, s[ki], blk); }
        else { fwd16_emit_pair<0, 0>(a.dd, base, s[ki], blk); fwd16_emit_pair<0, 1>(a.dd, base, s[ki], blk); }
      }
    }
    if (DROP == DROP_READ) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        keep_masks_wait(km[blk]);
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
#pragma unroll
          for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              s[2 * blk + ki][qi][r] = __builtin_amdgcn_inverse_ballot_w64(keep_mask16x16(km[blk], ki, qi, r)) ? s[2 * blk + ki][qi][r] : 0.f;
      }
    }
    // O^T += V^T P^T, one 32-key block at a time: 8 transposed reads, then 8 MFMAs
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      s16x4 lo[4], hi[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const unsigned ad = vtr + (tra0 ^ (dt << 5));
        if (blk == 0) { AFM_TR_RD(lo[dt], ad, 0); AFM_TR_RD(hi[dt], ad, 2048); }
        else { AFM_TR_RD(lo[dt], ad, 4096); AFM_TR_RD(hi[dt], ad, 6144); }
      }
      const e16x8 p0 = cvt8_2x4(s[2 * blk][0], s[2 * blk + 1][0]), p1 = cvt8_2x4(s[2 * blk][1], s[2 * blk + 1][1]);
      tr_wait<4>();
      o[0][0] = mfma16(tr_join(lo[0], hi[0]), p0, o[0][0]);
      o[0][1] = mfma16(tr_join(lo[0], hi[0]), p1, o[0][1]);
      o[1][0] = mfma16(tr_join(lo[1], hi[1]), p0, o[1][0]);
      o[1][1] = mfma16(tr_join(lo[1], hi[1]), p1, o[1][1]);
      tr_wait<0>();
      o[2][0] = mfma16(tr_join(lo[2], hi[2]), p0, o[2][0]);
      o[2][1] = mfma16(tr_join(lo[2], hi[2]), p1, o[2][1]);
      o[3][0] = mfma16(tr_join(lo[3], hi[3]), p0, o[3][0]);
      o[3][1] = mfma16(tr_join(lo[3], hi[3]), p1, o[3][1]);
    }
  }
  if (DROP == DROP_BITS) bits_flush();
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    float ll = l[qi];
    ll += __shfl_xor(ll, 16, 64);
    ll += __shfl_xor(ll, 32, 64);
    const float inv = ll > 0.f ? a.dd.scale16 / ll : 0.f;
    if (q[qi] < a.Tq) {
      e16* op = O + ((int64_t)b * a.Tq + q[qi]) * a.ldo + hd * DH + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const e16x4 v = {(e16)(o[dt][qi][0] * inv), (e16)(o[dt][qi][1] * inv), (e16)(o[dt][qi][2] * inv), (e16)(o[dt][qi][3] * inv)};
        *(e16x4*)(op + 16 * dt) = v;
      }
      if (g == 0) lse[((int64_t)b * a.H + hd) * a.Tq + q[qi]] = ll > 0.f ? (m[qi] + __log2f(ll)) * 0.69314718055994531f : INFINITY;
    }
  }
}
