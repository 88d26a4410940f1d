// dK / dV of the flash-attention backward for SHORT query sequences (the decoder's cross-attention: T = 128 ... 192 queries against
// S = 1 024 memory keys), on v_mfma_f32_16x16x32 (round 5; included by afm_attn_mfma_impl.h inside the dtype namespace, after
// afm_attn_m16_impl.h whose helpers and lane maps it uses).
//
// The general kernels give a workgroup 128 keys and loop over 64-query tiles.  With two or three query tiles that loop is over before
// it has started: a workgroup's life is its prologue -- fragment loads, masks, tile list, ring fill, each a dependent round trip to L2 /
// HBM -- and the launch needs ceil(Tk / 128) * B * H of them (c2's cross-attention: 8 192 workgroups of ~20 us for ~5 us of products;
// measured 0.24 ... 0.33 ms per launch = 210 ... 290 TF/s where the same kernel reaches 840 over 1 024 queries).  Here the loops are turned
// inside out: ALL query tiles of one (batch, head) -- Q and dO images, lse, -delta: 33 ... 50 KB -- are brought into LDS once, and the
// workgroup then walks the head's key blocks; a key block's dK / dV are complete after the (two or three) resident tiles and are stored
// at once.  After the prologue the LDS is read-only, so the loop has NO barrier: the four waves drift apart freely, and a wave loads
// the K / V fragments, key mask and keep-bit dwords of its NEXT 32 keys from global memory into registers while it computes the
// current ones.  Per unit the arithmetic is k_attn_bwd_dkv_m16's (same lane maps, same LDS image swizzle, same order of products).
//   grid = nchunks * H * B, a chunk = `nb` consecutive 128-key blocks (the whole head when B * H >= 1 024)
//   LDS  = ntq * (Q image 8 KB + dO image 8 KB + lse 256 B + -delta 256 B),  ntq = ceil(Tq / 64) <= 3
// Conditions (dispatch): no causal mask, Tq <= 192 < 256 <= Tk, dropout through the keep-bit tensor or off.

template <int DROP, int NTQ>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dkv_sq(AttnM a, const e16* __restrict__ Q, const e16* __restrict__ K,
                                                          const e16* __restrict__ V, const e16* __restrict__ dO,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          e16* __restrict__ dK, e16* __restrict__ dV, int nb, int nchunks) {
  static_assert(DROP == DROP_NONE || DROP == DROP_BITS, "the re-hash path stays with the general kernels");
  constexpr int IMG = KT * DH * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, g = lane >> 4, c16 = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, nchunks);
  const int hd = blk_.hd, b = blk_.b;
  const int nkb = (a.Tk + 127) / 128;
  const int kb_lo = blk_.xb * nb, kb_hi = min(nkb, kb_lo + nb);
  const e16* Qb = Q + (int64_t)b * a.Tq * a.ldq + hd * DH;
  const e16* Db = dO + (int64_t)b * a.Tq * a.ldo + hd * DH;
  const int64_t lbase = ((int64_t)b * a.H + hd) * a.Tq;
  float* const Ls = (float*)(lds + NTQ * 2 * IMG);      // lse (natural log units), then -delta, NTQ * 64 each
  float* const Ds = Ls + NTQ * KT;
  // ---- prologue: every query tile of the head into LDS (rows past Tq are clamped copies, masked below)
#pragma unroll
  for (int j = 0; j < NTQ; ++j) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      dma_piece_tr16(lds + j * 2 * IMG, Qb, a.ldq, j * KT, a.Tq, w + 4 * u, lane);
      dma_piece_tr16(lds + j * 2 * IMG + IMG, Db, a.ldo, j * KT, a.Tq, w + 4 * u, lane);
    }
    if (w < 2) {
      int qq = j * KT + lane;
      qq = qq < a.Tq ? qq : a.Tq - 1;
      const float* src = (w == 0 ? lse : delta) + lbase + qq;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)((w == 0 ? Ls : Ds) + j * KT), 4, 0, 0);
    }
  }
  struct Raw { e16x8 kf[2][2], vf[2][2]; uint32_t bw[2][2 * NTQ]; bool kmasked[2]; };
  auto load = [&](int kb, Raw& r) __attribute__((always_inline)) {
    const int k0 = kb * 128 + w * 32;
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) {
      const int key = k0 + 16 * ki + c16;
      const int kc = key < a.Tk ? key : a.Tk - 1;
      r.kmasked[ki] = key >= a.Tk || (a.key_pad && a.key_pad[(int64_t)b * a.Tk + kc]);
      const e16* kp = K + ((int64_t)b * a.Tk + kc) * a.ldk + hd * DH + 8 * g;
      const e16* vp = V + ((int64_t)b * a.Tk + kc) * a.ldv + hd * DH + 8 * g;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { r.kf[ki][ks] = ld8_once(kp + 32 * ks); r.vf[ki][ks] = ld8_once(vp + 32 * ks); }
      if (DROP == DROP_BITS) {      // the key's dword of every 32-query block: bit q = keep(query 32 qb + q, key)
        const int kb32 = min(k0 >> 5, a.nk32 - 1);
        const uint32_t* bp = (const uint32_t*)(a.bits + (((int64_t)(b * a.H + hd) * a.nq32) * a.nk32 + kb32) * 16) + bits_word_of_key(16 * ki + c16);
#pragma unroll
        for (int qb = 0; qb < 2 * NTQ; ++qb) r.bw[ki][qb] = bp[(int64_t)qb * a.nk32 * 32];
      }
    }
  };
  Raw cur;
  load(kb_lo, cur);
  attn_wait_vmcnt<0>();
  __syncthreads();      // the images are complete; from here on the LDS is only read
  unsigned tra0;        // transposed reads: as the dQ form of afm_attn_m16_impl.h
  {
    const int qq = (lane >> 2) & 3, p = lane & 3;
    const int s2 = 2 * (g & 1) + (qq >> 1);
    tra0 = (4 * g + qq) * 128 + ((2 * s2 + (p >> 1)) << 4) + ((p & 1) << 3);
  }
  for (int kb = kb_lo; kb < kb_hi; ++kb) {
    Raw nxt = cur;
    if (kb + 1 < kb_hi) load(kb + 1, nxt);      // in flight under this block's products
    const int k0 = kb * 128 + w * 32;
    int key[2];
    e16x8 (&kf)[2][2] = cur.kf, (&vf)[2][2] = cur.vf;
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) {
      key[ki] = k0 + 16 * ki + c16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {      // K by scale * log2(e), V by the dropout scale (as in the general kernels), in place
          kf[ki][ks][j] = (e16)((float)kf[ki][ks][j] * a.scale_log2);
          if (DROP != DROP_NONE) vf[ki][ks][j] = (e16)((float)vf[ki][ks][j] * a.dd.scale16);
        }
    }
    const bool dead = __all(cur.kmasked[0] && cur.kmasked[1]);      // 32 padded keys: zero rows, no products
    f32x4 dk[4][2], dv[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int ki = 0; ki < 2; ++ki) { dk[dt][ki] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt][ki] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    if (!dead) {
#pragma unroll
      for (int j = 0; j < NTQ; ++j) {
        const unsigned char* Qrow = lds + j * 2 * IMG;
        const unsigned char* Drow = Qrow + IMG;
        const unsigned qtr = (unsigned)(uintptr_t)Qrow, dtr = (unsigned)(uintptr_t)Drow;
        const bool ragged = (j + 1) * KT > a.Tq;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          if (j * KT + 32 * blk >= a.Tq) continue;      // a 32-query half past the sequence (wave-uniform)
          f32x4 s[2][2], dp[2][2];      // [query tile][key tile]
#pragma unroll
          for (int qi = 0; qi < 2; ++qi) {
            const f32x4 Lq = *(const f32x4*)(Ls + j * KT + 32 * blk + 16 * qi + 4 * g) * -1.4426950408889634f;
            const f32x4 Dq = *(const f32x4*)(Ds + j * KT + 32 * blk + 16 * qi + 4 * g);
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
          if (ragged) {      // the last, partly filled tile
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const bool msk = j * KT + 32 * blk + 16 * qi + 4 * g + r >= a.Tq;
#pragma unroll
                for (int ki = 0; ki < 2; ++ki) s[qi][ki][r] = msk ? -INFINITY : s[qi][ki][r];
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
              const uint32_t word = cur.bw[ki][2 * j + blk] >> (4 * g);
#pragma unroll
              for (int qi = 0; qi < 2; ++qi) {
                const f32x4 nd = *(const f32x4*)(Ds + j * KT + 32 * blk + 16 * qi + 4 * g);
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
#pragma unroll
    for (int ki = 0; ki < 2; ++ki)
      if (key[ki] < a.Tk) {
        e16* dkp = dK + ((int64_t)b * a.Tk + key[ki]) * a.lddk + hd * DH + 4 * g;
        e16* dvp = dV + ((int64_t)b * a.Tk + key[ki]) * a.lddv + hd * DH + 4 * g;
        const bool z = cur.kmasked[ki];      // a padded key took no part in any softmax: its dK / dV rows are zero
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
    cur = nxt;
  }
}
