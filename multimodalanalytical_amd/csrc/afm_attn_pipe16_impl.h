// The software-pipelined dK / dV kernel (afm_attn_pipe_impl.h) on v_mfma_f32_16x16x32 (round 5; included by
// afm_attn_mfma_impl.h inside the dtype namespace, behind afm_attn_pipe_impl.h and afm_attn_m16_impl.h whose helpers it uses).
//
// Same ring, same tile loop, same unit pipeline -- A(u + 1) beside B(u) beside C(u - 1), a unit = 32 queries x the wave's 32 keys -- with
// every product as 2 x 2 tiles of 16 x 16 (a lane owns key 16 ki + c of key tile ki and queries 16 qi + 4 g + r of query tile qi):
//   A(u)  16 MFMAs: fragment f = (k-step ks, query tile qi, Q | dO) is one ds_read_b128 and feeds the two key tiles in slots 2 f, 2 f + 1
//   B(u)  16 scores per lane, ONE per slot: p = exp2(S'), keep bit 16 qi + r of the lane's key dword (>> 4 g), dS and P~ packed in pairs
//   C(u)  16 MFMAs: transposed fragment t = (d-tile dt, dO^T -> dV | Q^T -> dK) is two ds_read_b64_tr_b16 and feeds both key tiles;
//         the B operands are the packed words of unit u - 1: [query tile 0: r0 r1 | r2 r3 | query tile 1: r0 r1 | r2 r3]
// A group = sixteen slots { the reads of the NEXT fragment pair (every second slot); wait for this pair's; C-MFMA, A-MFMA; one score }.
// The keep-bit tensor keeps its 32 x 32 layout: a key's 32 query bits are one dword, which is what this kernel reads anyway.
// Q and dO images use the swizzle chunk ^ ((row >> 1) & 3) << 1 (dma_piece_tr16): conflict-free for the 16 x 16 x 32 row reads AND the
// transposed reads.  The accumulation order differs from the 32 x 32 x 16 kernels: results agree to rounding, not bit for bit.
// Taken for: no causal mask, Tq a multiple of 64, keep bits or no dropout, four waves.  THE DEFAULT dK/dV kernel there since round 5
// (0.652 ... 0.658 ms against the 32 x 32 x 16 pipeline's 0.690 ... 0.694 at the c2 encoder shape with keep bits, +0.9 % on the step);
// afm_attn_shape.reserved & 16384 keeps the 32 x 32 x 16 pipeline (A / B runs, its bit-identity test).

template <int DROP>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_dkv_pipe16(AttnM a, const e16* __restrict__ Q, const e16* __restrict__ K,
                                                             const e16* __restrict__ V, const e16* __restrict__ dO,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             e16* __restrict__ dK, e16* __restrict__ dV) {
  constexpr int NW = 4;
  constexpr int IMG = KT * DH * 2;                    // one dual-use image
  constexpr int STAGE = 2 * IMG, NS = 3;             // main ring: Q image, dO image
  constexpr int AUX0 = NS * STAGE, AUXSLOT = 2048 + NW * 1024;      // aux ring as in the 32 x 32 kernel: lse[4][64], -delta[4][64], keep bits [NW][4][64 dwords]
  constexpr int LS_OFF = 0, DS_OFF = 1024, KB_OFF = 2048;
  constexpr int KPB = 32 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, g = lane >> 4, c16 = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tk + KPB - 1) / KPB);
  const int hd = blk_.hd, b = blk_.b;
  const int k0 = blk_.xb * KPB + w * 32;
  const int64_t rq = attn_row0(a.q_off, b, a.Tq), rk = attn_row0(a.k_off, b, a.Tk);
  const int lim_k = attn_slot(a.k_off, b, a.Tk);      // key rows of this sample that are its own (packed: its slot)
  // query rows of this sample that are its own: the Q / dO tile loads clamp to the slot's last row beyond it (a 64-query tile may reach 32
  // rows past a slot; those queries are P = 0 -- lse +inf, delta 0 -- but 0 x whatever the rows behind the LAST slot hold must stay 0)
  const int lim_q = max(1, attn_slot(a.q_off, b, a.Tq));
  {
    int64_t tail0;
    if (attn_tail_block(a.k_off, a.B, b, blk_.xb, a.Tk, tail0)) {      // packed rows, a key block beyond the sample's slot: zeros to its block of the dead tail
      static_assert(KPB == 128, "the dead-tail bijection is stated in 128-row blocks");
      const int64_t fe = attn_fill_end(a.nofill, a.k_off, a.B);
      const e16x4 z = {(e16)0.f, (e16)0.f, (e16)0.f, (e16)0.f};
#pragma unroll
      for (int ki = 0; ki < 2; ++ki) {
        const int64_t trow = tail0 + w * 32 + 16 * ki + c16;
        if (trow >= fe) continue;
        e16* dkp = dK + trow * a.lddk + hd * DH + 4 * g;
        e16* dvp = dV + trow * a.lddv + hd * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { *(e16x4*)(dkp + 16 * dt) = z; *(e16x4*)(dvp + 16 * dt) = z; }
      }
      return;
    }
  }
  const e16* Qb = Q + rq * a.ldq + hd * DH;
  const e16* Db = dO + rq * a.ldo + hd * DH;
  e16x8 kf[2][2], vf[2][2];                           // [key tile][k-step]
  int key[2];
  bool kmasked[2];
  bool wave_all_masked = true;
#pragma unroll
  for (int ki = 0; ki < 2; ++ki) {
    key[ki] = k0 + 16 * ki + c16;
    const int kc = key[ki] < a.Tk ? key[ki] : a.Tk - 1;
    kmasked[ki] = key[ki] >= a.Tk || (a.key_pad && a.key_pad[(int64_t)b * a.Tk + kc]);
    wave_all_masked = wave_all_masked && __all(kmasked[ki]);
    const e16* kp = K + (rk + kc) * a.ldk + hd * DH + 8 * g;
    const e16* vp = V + (rk + kc) * a.ldv + hd * DH + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ki][ks] = ld8_once(kp + 32 * ks); vf[ki][ks] = ld8_once(vp + 32 * ks);
#pragma unroll
      for (int j = 0; j < 8; ++j) {      // K by scale * log2(e), V by the dropout scale (as in the 32 x 32 kernels)
        kf[ki][ks][j] = (e16)((float)kf[ki][ks][j] * a.scale_log2);
        if (DROP != DROP_NONE) vf[ki][ks][j] = (e16)((float)vf[ki][ks][j] * a.dd.scale16);
      }
    }
  }
  f32x4 dk[4][2], dv[4][2];                           // [d-tile][key tile]
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) { dk[dt][ki] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt][ki] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const int ntiles = a.Tq / KT;
  const int64_t lbase = ((int64_t)b * a.H + hd) * a.Tq;
  unsigned long long* qmaskw = (unsigned long long*)(lds + AUX0 + 2 * AUXSLOT);
  int* const tl = (int*)(qmaskw + ntiles) + 1;
  if (a.qskip) build_mask_words(qmaskw, a.key_pad, b, a.Tq, ntiles, w, lane);
  auto store_rows = [&](bool zeros) {
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) {
      // packed rows: keys beyond the slot (a partly used last block) are the next sample's rows: zeros go to their share of the dead tail
      const bool own = key[ki] < lim_k;
      const int64_t row = attn_out_row(a.k_off, a.B, b, a.Tk, rk, key[ki], lim_k);
      if (row >= 0 && (own || row < attn_fill_end(a.nofill, a.k_off, a.B))) {
        e16* dkp = dK + row * a.lddk + hd * DH + 4 * g;
        e16* dvp = dV + row * a.lddv + hd * DH + 4 * g;
        const bool z = zeros || kmasked[ki] || !own;          // a padded key took no part in any softmax: zero rows
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
    }
  };
  if (__syncthreads_and(wave_all_masked)) {
    store_rows(true);
    return;
  }
  build_tile_list(tl, a.qskip ? qmaskw : nullptr, 0, ntiles, w, lane);
  __syncthreads();   // K / V fragment loads retired before the LDS-DMA ring starts
  const int nlive = __builtin_amdgcn_readfirstlane(tl[-1]);
  auto issue = [&](int j, int stage_off) {
    unsigned char* st = lds + stage_off;
    const int row0 = tl[j] * KT;
#pragma unroll
    for (int u = 0; u < 8 / NW; ++u) {
      dma_piece_tr16(st, Qb, a.ldq, row0, lim_q, w + NW * u, lane);
      dma_piece_tr16(st + IMG, Db, a.ldo, row0, lim_q, w + NW * u, lane);
    }
  };
  auto auxo = [&](int j) { return AUX0 + ((j >> 2) & 1) * AUXSLOT + (j & 3) * 256; };
  auto issue_aux = [&](int gq) {      // listed tiles 4 gq .. 4 gq + 3 (clamped): lane l brings 16 bytes of tile 4 gq + (l >> 4)
    unsigned char* st = lds + AUX0 + (gq & 1) * AUXSLOT;
    int jj = 4 * gq + (lane >> 4);
    jj = jj < nlive ? jj : nlive - 1;
    const int tq = tl[jj];
    if (w < 2) {
      const float* src = (w == 0 ? lse : delta) + lbase + tq * KT + (lane & 15) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + w * 1024), 16, 0, 0);
    }
    if (DROP == DROP_BITS) {
      const int kb32 = min(k0 >> 5, a.nk32 - 1), qb32 = 2 * tq + ((lane >> 3) & 1);
      const unsigned long long* src = a.bits + (((int64_t)(b * a.H + hd) * a.nq32 + qb32) * a.nk32 + kb32) * 16 + (lane & 7) * 2;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + KB_OFF + w * 1024), 16, 0, 0);
    }
  };

  // ---- lane address registers (stage 0; moved from stage to stage by adding a wave-uniform byte difference)
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  unsigned aA[2];                                        // row fragments of the A target's tile: k-step ks of rows c16 [+ 16 qi, + 32 blk, image: immediates]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) aA[ks] = lds0 + c16 * 128 + (((4 * ks + g) ^ (((c16 >> 1) & 3) << 1)) << 4);
  unsigned xa[4];                                        // transposed fragments of the C unit's tile, d-tile dt
  {
    const int qq = (lane >> 2) & 3, p = lane & 3;
    const int s2 = 2 * (g & 1) + (qq >> 1);
    const unsigned tra0 = (4 * g + qq) * 128 + ((2 * s2 + (p >> 1)) << 4) + ((p & 1) << 3);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) xa[dt] = lds0 + (tra0 ^ (dt << 5));
  }
  unsigned aLA = lds0 + AUX0 + 16 * g;                   // lse / -delta of the A target's tile: queries 16 qi + 4 g .. + 3 (+ 32 blk)
  unsigned aLB = lds0 + AUX0 + 16 * g;                   // -delta of the B unit's tile
  unsigned aW = lds0 + AUX0 + KB_OFF + w * 1024 + 4 * bits_word_of_key(c16);   // keep dword of key c16 of the 32-key block (key tile 1: + 64 bytes)

  auto move2 = [&](unsigned (&arr)[2], int diff) { arr[0] += (unsigned)diff; arr[1] += (unsigned)diff; };
  auto move4 = [&](unsigned (&arr)[4], int diff) {
#pragma unroll
    for (int d = 0; d < 4; ++d) arr[d] += (unsigned)diff;
  };
  f32x4 s[2][2][2], dp[2][2][2];                        // [unit buffer][query tile][key tile]
  uint32_t pfw[2][2][4], dsw[2][2][4];                  // B's output per key tile: P~ and dS as packed pairs (words 2 qi + (r >> 1))
#pragma unroll
  for (int ki = 0; ki < 2; ++ki)
#pragma unroll
    for (int i = 0; i < 4; ++i) { pfw[1][ki][i] = 0u; dsw[1][ki][i] = 0u; }   // "C(-1, 1)" of the first group adds exact zeros
#pragma unroll
  for (int qi = 0; qi < 2; ++qi)
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) { s[1][qi][ki] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[1][qi][ki] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  u32x4_ abuf[2];                                        // fragment pair operands, two deep
  s16x4 clo[2], chi[2];
  f32x4 ndb[2];                                          // -delta of the B unit's queries, per query tile
  uint32_t word[2] = {0u, 0u};                           // keep dwords of the lane's two keys (>> 4 g)
  float hq = 0.f, hd_ = 0.f;                             // first score of a pair (P~ and dS), waiting for its partner

  // operands of fragment pair P (slots 2 P, 2 P + 1): the A fragment, the C fragment (two transposed reads); -delta of a query tile where one starts
  auto reads = [&](auto BB_, auto P_) __attribute__((always_inline)) {
    constexpr int BB = decltype(BB_)::value, BA = 1 - BB, BC = 1 - BB, P = decltype(P_)::value;
    (void)ndb; (void)aLB; (void)abuf; (void)clo; (void)chi; (void)xa; (void)aA; (void)BA; (void)BC;
    constexpr int ks = P >> 2, qi = (P >> 1) & 1, which = P & 1;                  // A fragment: k-step, query tile, Q (0) | dO (1)
    AFM_LDS_RD128(abuf[P & 1], aA[ks], which * IMG + BA * 4096 + qi * 2048);
    constexpr int cw = P & 1, dt = P >> 1;                                        // C fragment: dO^T -> dV (0) | Q^T -> dK (1), d-tile
    constexpr int cimg = cw ? 0 : IMG, olo = cimg + BC * 4096, ohi = olo + 2048;
    AFM_TR_RDN(clo[P & 1], xa[dt], olo);
    AFM_TR_RDN(chi[P & 1], xa[dt], ohi);
    if constexpr ((P == 0 || P == 4) && DROP != DROP_NONE) AFM_LDS_RD128(ndb[P >> 2], aLB, DS_OFF + (32 * BB + 16 * (P >> 2)) * 4);
  };
  auto group = [&](auto BB_) __attribute__((always_inline)) {
    constexpr int BB = decltype(BB_)::value, BA = 1 - BB, BC = 1 - BB;
    (void)ndb; (void)aLB; (void)aLA; (void)aW; (void)abuf; (void)clo; (void)chi; (void)xa; (void)aA; (void)word; (void)s; (void)dp; (void)pfw; (void)dsw; (void)dk; (void)dv; (void)kf; (void)vf; (void)g; (void)hq; (void)hd_;
    // ---- preamble: the A target's initial accumulators, the keep dwords, pair 0's operands
    {
      f32x4 si[2], di[2];
      AFM_LDS_RD128(si[0], aLA, LS_OFF + (32 * BA + 0) * 4);  AFM_LDS_RD128(si[1], aLA, LS_OFF + (32 * BA + 16) * 4);
      AFM_LDS_RD128(di[0], aLA, DS_OFF + (32 * BA + 0) * 4);  AFM_LDS_RD128(di[1], aLA, DS_OFF + (32 * BA + 16) * 4);
      if constexpr (DROP == DROP_BITS) {
        AFM_LDS_RD32(word[0], aW, 128 * BB);
        AFM_LDS_RD32(word[1], aW, 128 * BB + 64);
      }
      reads(BB_, std::integral_constant<int, 0>{});
      lgk_wait<0>();
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {
        const f32x4 sv = si[qi] * -1.4426950408889634f;
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) { s[BA][qi][ki] = sv; dp[BA][qi][ki] = di[qi]; }
      }
      if constexpr (DROP == DROP_BITS) { word[0] >>= 4 * g; word[1] >>= 4 * g; }
      __builtin_amdgcn_sched_barrier(0);
    }
    static_for<0, 16>([&](auto I_) __attribute__((always_inline)) {
      constexpr int i = decltype(I_)::value, P = i >> 1, ki = i & 1;
      (void)ndb; (void)abuf; (void)clo; (void)chi; (void)word; (void)s; (void)dp; (void)pfw; (void)dsw; (void)dk; (void)dv; (void)kf; (void)vf; (void)hq; (void)hd_;
      // the next pair's operands first (into the registers pair P - 1 has finished with), then wait for this pair's: LDS returns in order
      if constexpr ((i & 1) == 0) {
        if constexpr (P < 7) {
          reads(BB_, std::integral_constant<int, P + 1>{});
          lgk_wait<3 + (((P + 1) == 4 && DROP != DROP_NONE) ? 1 : 0)>();
        } else {
          lgk_wait<0>();
        }
      }
      {
        const e16x8 ca = tr_join(clo[P & 1], chi[P & 1]);
        const e16x8 fa = __builtin_bit_cast(e16x8, abuf[P & 1]);
        constexpr int cw = P & 1, dt = P >> 1;
        if constexpr (cw == 0) {
          const e16x8 pf = __builtin_bit_cast(e16x8, (u32x4_){pfw[BC][ki][0], pfw[BC][ki][1], pfw[BC][ki][2], pfw[BC][ki][3]});
          dv[dt][ki] = mfma16(ca, pf, dv[dt][ki]);
        } else {
          const e16x8 df = __builtin_bit_cast(e16x8, (u32x4_){dsw[BC][ki][0], dsw[BC][ki][1], dsw[BC][ki][2], dsw[BC][ki][3]});
          dk[dt][ki] = mfma16(ca, df, dk[dt][ki]);
        }
        constexpr int ks = P >> 2, qi = (P >> 1) & 1, which = P & 1;
        if constexpr (which) dp[BA][qi][ki] = mfma16(fa, vf[ki][ks], dp[BA][qi][ki]);
        else s[BA][qi][ki] = mfma16(fa, kf[ki][ks], s[BA][qi][ki]);
      }
      {   // one score of B: e = i -> query tile e >> 3, key tile (e >> 2) & 1, register e & 3
        constexpr int eq = i >> 3, ek = (i >> 2) & 1, er = i & 3;
        const float p0 = fast_exp2(s[BB][eq][ek][er]);
        float d0, q0;
        if constexpr (DROP == DROP_BITS) {
          uint32_t m0, e0, z0;
          asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m0) : "v"(word[ek]), "n"(16 * eq + er));
          asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(e0) : "v"(m0), "v"(dp[BB][eq][ek][er]), "v"(ndb[eq][er]));
          asm("v_and_b32 %0, %1, %2" : "=v"(z0) : "v"(m0), "v"(p0));
          d0 = p0 * __builtin_bit_cast(float, e0);      // dS = P (keep ? scale dP - delta : -delta)
          q0 = __builtin_bit_cast(float, z0);
        } else {
          d0 = p0 * dp[BB][eq][ek][er]; q0 = p0;
        }
        if constexpr ((er & 1) == 0) { hq = q0; hd_ = d0; }
        else {
          uint32_t pw = cvt_pk2(hq, q0), dw = cvt_pk2(hd_, d0);
          asm volatile("" : "+v"(pw), "+v"(dw));
          pfw[BB][ek][2 * eq + (er >> 1)] = pw; dsw[BB][ek][2 * eq + (er >> 1)] = dw;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  constexpr std::integral_constant<int, 0> B0{};
  constexpr std::integral_constant<int, 1> B1{};

  // ---- prologue: tiles 0 and 1 into fresh stages
  issue_aux(0);
  issue(0, 0);
  if (nlive > 1) issue(1, STAGE);
  attn_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  __builtin_assume(nlive >= 1);
  if (wave_all_masked) {     // 32 padded keys: the wave only keeps the ring and the barriers going (its outputs are zeros)
    int so = 0;
    for (int j = 0; j < nlive; ++j) {
      attn_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (j + 2 < nlive) {
        issue(j + 2, so == 0 ? (NS - 1) * STAGE : so - STAGE);
        if (((j + 2) & 3) == 0) issue_aux((j + 2) >> 2);
      }
      so = so + STAGE == NS * STAGE ? 0 : so + STAGE;
    }
  } else {
    {   // A(0, 0): the one product pair that overlaps nothing
      f32x4 si[2], di[2];
      AFM_LDS_RD128(si[0], aLA, LS_OFF + 0);  AFM_LDS_RD128(si[1], aLA, LS_OFF + 64);
      AFM_LDS_RD128(di[0], aLA, DS_OFF + 0);  AFM_LDS_RD128(di[1], aLA, DS_OFF + 64);
      u32x4_ fq[2][2], fd[2][2];                        // [k-step][query tile]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks == 0) {
          AFM_LDS_RD128(fq[0][0], aA[0], 0); AFM_LDS_RD128(fq[0][1], aA[0], 2048); AFM_LDS_RD128(fd[0][0], aA[0], IMG); AFM_LDS_RD128(fd[0][1], aA[0], IMG + 2048);
        } else {
          AFM_LDS_RD128(fq[1][0], aA[1], 0); AFM_LDS_RD128(fq[1][1], aA[1], 2048); AFM_LDS_RD128(fd[1][0], aA[1], IMG); AFM_LDS_RD128(fd[1][1], aA[1], IMG + 2048);
        }
      }
      lgk_wait<0>();
#pragma unroll
      for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) { s[0][qi][ki] = si[qi] * -1.4426950408889634f; dp[0][qi][ki] = di[qi]; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
          for (int ki = 0; ki < 2; ++ki) {
            s[0][qi][ki] = mfma16(__builtin_bit_cast(e16x8, fq[ks][qi]), kf[ki][ks], s[0][qi][ki]);
            dp[0][qi][ki] = mfma16(__builtin_bit_cast(e16x8, fd[ks][qi]), vf[ki][ks], dp[0][qi][ki]);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
    int sB = 0;                                            // stage byte offset of tile j
    for (int j = 0; j < nlive; ++j) {
      const int sN = sB + STAGE == NS * STAGE ? 0 : sB + STAGE;      // stage of tile j + 1
      const int sP = sB == 0 ? (NS - 1) * STAGE : sB - STAGE;        // stage of tile j - 1 (and of tile j + 2)
      group(B0);                                           // B(j, 0) beside A(j, 1) and C(j - 1, 1)
      const int dC = j == 0 ? 0 : sB - sP;                 // the C unit is tile j's from here on
      move4(xa, dC);
      attn_wait_vmcnt<0>();                                // tile j + 1 (the only one in flight)
      __builtin_amdgcn_s_barrier();                        // ... and every wave is past C(j - 1, 1), the last reader of tile j - 1's stage
      if (j + 2 < nlive) {
        issue(j + 2, sP);
        if (((j + 2) & 3) == 0) issue_aux((j + 2) >> 2);
      }
      const int dX = auxo(j + 1) - auxo(j);
      move2(aA, sN - sB);                                  // the A target is tile j + 1's first block
      aLA += (unsigned)dX;
      group(B1);                                           // B(j, 1) beside A(j + 1, 0) and C(j, 0)
      aLB += (unsigned)dX;
      aW += (unsigned)dX;
      sB = sN;
    }
    // ---- epilogue: C(last, 1); the transposed-read registers already point at the last tile
    {
      s16x4 lo[8], hi[8];
      static_for<0, 8>([&](auto P_) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value, cw = P & 1, dt = P >> 1;
        constexpr int cimg = cw ? 0 : IMG, olo = cimg + 4096, ohi = olo + 2048;
        (void)lo; (void)hi; (void)xa;
        AFM_TR_RDN(lo[P], xa[dt], olo);
        AFM_TR_RDN(hi[P], xa[dt], ohi);
      });
      lgk_wait<0>();
      static_for<0, 8>([&](auto P_) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value, cw = P & 1, dt = P >> 1;
        (void)pfw; (void)dsw; (void)dk; (void)dv; (void)lo; (void)hi;
        const e16x8 ca = tr_join(lo[P], hi[P]);
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) {
          if constexpr (cw == 0)
            dv[dt][ki] = mfma16(ca, __builtin_bit_cast(e16x8, (u32x4_){pfw[1][ki][0], pfw[1][ki][1], pfw[1][ki][2], pfw[1][ki][3]}), dv[dt][ki]);
          else
            dk[dt][ki] = mfma16(ca, __builtin_bit_cast(e16x8, (u32x4_){dsw[1][ki][0], dsw[1][ki][1], dsw[1][ki][2], dsw[1][ki][3]}), dk[dt][ki]);
        }
      });
    }
  }
  store_rows(false);
}
