// dK / dV kernel with the MFMAs and the softmax arithmetic interleaved inside ONE wave's instruction stream (round 4; included by
// afm_attn_mfma_impl.h inside the dtype namespace).
//
// Why: under the profiler the round-3 kernel (k_attn_bwd_dkv_mfma) shows the matrix pipe busy 0.33 of the time and the vector pipe
// 0.55, the two sums never overlapping (profiles/r04_attn_fp16_pmc.json): a wave runs 8 MFMAs back to back, then ~145 vector
// instructions, then 8 MFMAs, and the second wave of the SIMD drifts into the same phase.  MFMA / VALU overlap on CDNA4 comes from one
// wave's own stream (MI355X_MICROARCH.md: an MFMA holds the vector issue port for 8 of its 32 cycles; a handful of single-issue
// instructions per gap are free), so the tile loop is software-pipelined by hand over "units" of 32 queries x the wave's 32 keys:
//
//   A(u)  S' = Q K^T - lse, dP' = dO V^T - delta            8 MFMAs, row fragments of Q and dO (ds_read_b128)
//   B(u)  P = exp2(S'), dS = P (keep ? dP' : -delta), P~     ~115 vector instructions, 16 packed words out
//   C(u)  dV^T += dO^T P~, dK^T += Q^T dS                    8 MFMAs, transposed fragments of dO and Q (ds_read_b64_tr_b16)
//
// and every group of eight slots runs  B(u)  beside  A(u + 1)  and  C(u - 1):  slot i = { the reads of slot i + 1; wait for slot i's;
// C-MFMA i, A-MFMA i, two scores of B }.  Each accumulator still receives its products in the round-3 order: results are bit-identical
// to k_attn_bwd_dkv_mfma (tests/test_gpu_fp16.py).
//
// LDS reads are inline asm (afm_attn_tiles.h explains why for the transposed ones; the same holds for plain reads beside an LDS-DMA
// ring): the compiler's counters do not see them, LDS returns in order, and every consumer sits behind `s_waitcnt lgkmcnt(n)` with n =
// the number of reads issued after the ones it needs, plus a scheduling fence.
//
// Ring: three stages of {Q image, dO image} (lse, -delta and the keep bits travel four tiles at a time in a two-slot ring of their own).  A(j + 1, 0) runs in the second half of tile
// j, so tile j + 1 must have landed by the MIDDLE of tile j (vmcnt(0) + the tile's one barrier sit there) and tile j's stage is read
// until the first half of tile j + 1 (C(j, 1)); tile j + 2 is issued right behind that barrier into the stage tile j - 1 left at it.
//
// Workgroup = NW waves x 32 keys: with eight waves a tile's sixteen LDS-DMA pieces cost each wave two instructions instead of four
// (an LDS-DMA instruction holds the issuing wave ~60 cycles whatever its size: the round-3 kernel's six per tile were a quarter of its time).
//
// Taken for: no causal mask, Tq a multiple of 64, dropout through the keep-bit tensor or none (everything else: the round-3 kernel).

#define AFM_LDS_RD128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define AFM_LDS_RD32(dst, addr, OFF) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define AFM_TR_RDN(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))

template <int N> __device__ __forceinline__ void lgk_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ uint32_t cvt_pk2(float x, float y) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef e16 e16x2_ __attribute__((ext_vector_type(2)));
  asm("" : "+v"(x), "+v"(y));    // opaque inputs: one packed conversion whatever produced them (cvt8_pk, afm_attn_tiles.h)
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_){x, y}, e16x2_));
}
typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));

#ifndef AFM_DKV_PIPE_OCC
#define AFM_DKV_PIPE_OCC 2
#endif
// AFM_PIPE_ABL (template parameter ABL; an AFM_ATTN_ABLATIONS build instantiates the list in afm_attn_mfma_impl.h and
// afm_attn_shape.reserved bits 12-19 pick one; bits 8 and 9 are the live selectors of the 8-wave / 64-key forms): timing ablations with wrong results -- 1 no MFMAs, 2 no arithmetic, 4 no slot reads,
// 8 no barrier, 16 no preamble reads, 32 no LDS-DMA after the prologue, 64 no waits for the slot reads.
template <int DROP, int NW, int KB = 1, int AFM_PIPE_ABL = 0>
__global__ __launch_bounds__(64 * NW, (NW == 4 && KB == 1) ? 2 : 1) void k_attn_bwd_dkv_pipe(AttnM a, const e16* __restrict__ Q, const e16* __restrict__ K,
                                                                             const e16* __restrict__ V, const e16* __restrict__ dO,
                                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                                             e16* __restrict__ dK, e16* __restrict__ dV) {
  constexpr int IMG = KT * DH * 2;                    // one dual-use image
  constexpr int STAGE = 2 * IMG, NS = 3;             // main ring: Q image, dO image (stage bases are multiples of 128: the transposed-read addresses XOR below bit 7)
  // aux ring: two slots, each the small per-tile data of FOUR tiles (one 1-KiB LDS-DMA instruction per array instead of four 256-byte
  // ones: what a wave pays for an LDS-DMA instruction does not depend on its size) -- lse[4][64], -delta[4][64], keep bits [NW][4][64 dwords]
  constexpr int AUX0 = NS * STAGE, AUXSLOT = 2048 + NW * KB * 1024;
  constexpr int LS_OFF = 0, DS_OFF = 1024, KB_OFF = 2048;      // relative to a tile's 256-byte column of its aux slot
  constexpr int KPB = 32 * KB * NW;                  // keys per workgroup (KB 32-key blocks per wave)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, (a.Tk + KPB - 1) / KPB);
  const int hd = blk_.hd, b = blk_.b;
  const int k0 = blk_.xb * KPB + w * 32 * KB;             // the wave's first key; key block kb covers k0 + 32 kb ..
  const e16* Qb = Q + (int64_t)b * a.Tq * a.ldq + hd * DH;
  const e16* Db = dO + (int64_t)b * a.Tq * a.ldo + hd * DH;
  e16x8 kf[KB][4], vf[KB][4];
  bool kmasked[KB];
  bool wave_all_masked = true;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int key = k0 + 32 * kb + (lane & 31);
    const int kc = key < a.Tk ? key : a.Tk - 1;
    kmasked[kb] = key >= a.Tk || (a.key_pad && a.key_pad[(int64_t)b * a.Tk + kc]);
    wave_all_masked = wave_all_masked && __all(kmasked[kb]);
    const e16* kp = K + ((int64_t)b * a.Tk + kc) * a.ldk + hd * DH + 8 * h;
    const e16* vp = V + ((int64_t)b * a.Tk + kc) * a.ldv + hd * DH + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) { kf[kb][s] = ld8_once(kp + 16 * s); vf[kb][s] = ld8_once(vp + 16 * s); }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {      // as in the round-3 kernel: K by scale * log2(e), V by the dropout scale
        kf[kb][s][j] = (e16)((float)kf[kb][s][j] * a.scale_log2);
        if (DROP != DROP_NONE) vf[kb][s][j] = (e16)((float)vf[kb][s][j] * a.dd.scale16);
      }
  }
  f32x16 dk[KB][2], dv[KB][2];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[kb][0][i] = 0.f; dk[kb][1][i] = 0.f; dv[kb][0][i] = 0.f; dv[kb][1][i] = 0.f; }
  const int ntiles = a.Tq / KT;
  const int64_t lbase = ((int64_t)b * a.H + hd) * a.Tq;
  unsigned long long* qmaskw = (unsigned long long*)(lds + AUX0 + 2 * AUXSLOT);
  int* const tl = (int*)(qmaskw + ntiles) + 1;
  if (a.qskip) build_mask_words(qmaskw, a.key_pad, b, a.Tq, ntiles, w, lane);
  auto store_rows = [&](bool zeros) {       // dK (x 1/sqrt(dh)) and dV rows of the wave's keys
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int key = k0 + 32 * kb + (lane & 31);
      if (key < a.Tk) {
        e16* dkp = dK + ((int64_t)b * a.Tk + key) * a.lddk + hd * DH + 4 * h;
        e16* dvp = dV + ((int64_t)b * a.Tk + key) * a.lddv + hd * DH + 4 * h;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            e16x4 x = {(e16)0.f, (e16)0.f, (e16)0.f, (e16)0.f}, y = x;
            if (!zeros) {
              x = (e16x4){(e16)(dk[kb][db][4 * g4 + 0] * a.scale), (e16)(dk[kb][db][4 * g4 + 1] * a.scale),
                          (e16)(dk[kb][db][4 * g4 + 2] * a.scale), (e16)(dk[kb][db][4 * g4 + 3] * a.scale)};
              y = (e16x4){(e16)dv[kb][db][4 * g4 + 0], (e16)dv[kb][db][4 * g4 + 1], (e16)dv[kb][db][4 * g4 + 2], (e16)dv[kb][db][4 * g4 + 3]};
            }
            *(e16x4*)(dkp + 32 * db + 8 * g4) = x;
            *(e16x4*)(dvp + 32 * db + 8 * g4) = y;
          }
      }
    }
  };
  if (__syncthreads_and(wave_all_masked)) {   // every key of the workgroup is padding: zero rows, nothing to load
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
      dma_piece_dual(st, Qb, a.ldq, row0, a.Tq, w + NW * u, lane);
      dma_piece_dual(st + IMG, Db, a.ldo, row0, a.Tq, w + NW * u, lane);
    }
  };
  auto auxo = [&](int j) { return AUX0 + ((j >> 2) & 1) * AUXSLOT + (j & 3) * 256; };     // tile j's column in the aux ring
  auto issue_aux = [&](int g) {       // listed tiles 4g .. 4g + 3 (clamped): lane l brings 16 bytes of tile 4g + (l >> 4)
    unsigned char* st = lds + AUX0 + (g & 1) * AUXSLOT;
    int jj = 4 * g + (lane >> 4);
    jj = jj < nlive ? jj : nlive - 1;
    const int tq = tl[jj];
    if (w < 2) {
      const float* src = (w == 0 ? lse : delta) + lbase + tq * KT + (lane & 15) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + w * 1024), 16, 0, 0);
    }
    if (DROP == DROP_BITS) {          // the tile's two 32-query blocks of each of the wave's key blocks: 2 x 128 bytes, eight 16-byte chunks each
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int kb32 = min((k0 >> 5) + kb, a.nk32 - 1), qb32 = 2 * tq + ((lane >> 3) & 1);
        const unsigned long long* src = a.bits + (((int64_t)(b * a.H + hd) * a.nq32 + qb32) * a.nk32 + kb32) * 16 + (lane & 7) * 2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(st + KB_OFF + (w * KB + kb) * 1024), 16, 0, 0);
      }
    }
  };

  // ---- lane address registers (stage 0; moved from stage to stage by adding a wave-uniform byte difference)
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  unsigned aA[4];                                        // row fragments of the A target's tile: k-slice ks of rows (lane & 31) [+ 32 blk: immediate]
  {
    const int r = lane & 31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) aA[ks] = lds0 + r * 128 + (((2 * ks + h) ^ dual_f(r)) << 4);
  }
  unsigned xa[4], xb[4];                                 // transposed fragments of the C unit's tile
  {
    const unsigned t0 = tr_dual_t0(lane);
#pragma unroll
    for (int d = 0; d < 4; ++d) { xa[d] = lds0 + (t0 ^ (d << 4)); xb[d] = lds0 + (t0 ^ (d << 4) ^ 64); }
  }
  unsigned aLA = lds0 + AUX0 + 16 * h;                   // lse / -delta of the A target's tile (tile 0's aux column)
  unsigned aLB = lds0 + AUX0 + 16 * h;                   // -delta of the B unit's tile
  unsigned aW = lds0 + AUX0 + KB_OFF + w * KB * 1024 + 4 * bits_word_of_key(lane & 31);   // keep word of the B unit's tile (key block kb: + 1024 kb)

  auto move_stage = [&](unsigned (&arr)[4], int diff) {
#pragma unroll
    for (int d = 0; d < 4; ++d) arr[d] += (unsigned)diff;
  };
  f32x16 s[2][KB], dp[2][KB];
  uint32_t pfw[2][KB][8], dsw[2][KB][8];                // B's output: P~ and dS of a unit as packed pairs (word i = scores 2i, 2i + 1)
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { pfw[1][kb][i] = 0u; dsw[1][kb][i] = 0u; }   // "C(-1, 1)" of the first group adds exact zeros
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[1][kb][i] = 0.f; dp[1][kb][i] = 0.f; }
  }
  u32x4_ abuf[2];                                        // slot operands, two deep: slot i + 1's are in flight while slot i computes
  s16x4 clo[2], chi[2];
  f32x4 ndb[2];
  uint32_t word[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) word[kb] = 0u;

  // operands of slot I (the A fragment, the C fragment pair, -delta of four queries)
  auto reads = [&](auto BB_, auto I_) __attribute__((always_inline)) {
    constexpr int BB = decltype(BB_)::value, BA = 1 - BB, BC = 1 - BB, i = decltype(I_)::value, ks = i >> 1;
    (void)ndb; (void)aLB; (void)abuf; (void)clo; (void)chi; (void)xa; (void)xb; (void)aA; (void)ks; (void)BA; (void)BC;
    if constexpr (i & 1) AFM_LDS_RD128(abuf[i & 1], aA[ks], IMG + BA * 4096);
    else AFM_LDS_RD128(abuf[i & 1], aA[ks], BA * 4096);
    constexpr int img = ((i >> 1) & 1) ? 0 : IMG, SL = i >> 2, DLO = SL ? 3 : 0, DHI = SL ? 1 : 2;
    constexpr int olo = img + BC * 4096 + 16 * SL * 128, ohi = olo + 8 * 128;
    if constexpr (i & 1) { AFM_TR_RDN(clo[i & 1], xb[DLO], olo); AFM_TR_RDN(chi[i & 1], xb[DHI], ohi); }
    else { AFM_TR_RDN(clo[i & 1], xa[DLO], olo); AFM_TR_RDN(chi[i & 1], xa[DHI], ohi); }
    if constexpr ((i & 1) == 0 && DROP != DROP_NONE) AFM_LDS_RD128(ndb[(i >> 1) & 1], aLB, DS_OFF + (32 * BB + 4 * i) * 4);
  };
  // One group of eight slots: B on block BB of the current tile, A's target = the other block (of the same tile when BB = 0, of the next
  // tile when BB = 1), C = the unit B finished in the previous group.  Slot = { C-MFMA, A-MFMA | the next slot's reads | two scores }.
  auto group = [&](auto BB_) __attribute__((always_inline)) {
    constexpr int BB = decltype(BB_)::value, BA = 1 - BB, BC = 1 - BB;
    (void)ndb; (void)aLB; (void)aLA; (void)aW; (void)abuf; (void)clo; (void)chi; (void)xa; (void)xb; (void)aA; (void)word; (void)s; (void)dp; (void)pfw; (void)dsw; (void)dk; (void)dv; (void)kf; (void)vf; (void)h;   // (clang: names used only under `if constexpr` in a nested generic lambda are not captured implicitly)
    // ---- preamble: the A target's initial accumulators, the keep word, slot 0's operands
    {
      f32x4 si[4], di[4];
      if constexpr (AFM_PIPE_ABL & 16) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) { si[g4] = ndb[0]; di[g4] = ndb[0]; }
      } else {
      AFM_LDS_RD128(si[0], aLA, LS_OFF + (32 * BA + 0) * 4);  AFM_LDS_RD128(si[1], aLA, LS_OFF + (32 * BA + 8) * 4);
      AFM_LDS_RD128(si[2], aLA, LS_OFF + (32 * BA + 16) * 4); AFM_LDS_RD128(si[3], aLA, LS_OFF + (32 * BA + 24) * 4);
      AFM_LDS_RD128(di[0], aLA, DS_OFF + (32 * BA + 0) * 4);  AFM_LDS_RD128(di[1], aLA, DS_OFF + (32 * BA + 8) * 4);
      AFM_LDS_RD128(di[2], aLA, DS_OFF + (32 * BA + 16) * 4); AFM_LDS_RD128(di[3], aLA, DS_OFF + (32 * BA + 24) * 4);
      if constexpr (DROP == DROP_BITS) {
        AFM_LDS_RD32(word[0], aW, 128 * BB);
        if constexpr (KB == 2) AFM_LDS_RD32(word[KB - 1], aW, 1024 + 128 * BB);
      }
      }
      if constexpr (!(AFM_PIPE_ABL & 4)) reads(BB_, std::integral_constant<int, 0>{});
      lgk_wait<0>();
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float sv = si[g4][j] * -1.4426950408889634f;
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) { s[BA][kb][4 * g4 + j] = sv; dp[BA][kb][4 * g4 + j] = di[g4][j]; }
        }
      if constexpr (DROP == DROP_BITS) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) word[kb] >>= 4 * h;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    static_for<0, 8>([&](auto I_) __attribute__((always_inline)) {
      constexpr int i = decltype(I_)::value, ks = i >> 1;
      (void)ndb; (void)abuf; (void)clo; (void)chi; (void)word; (void)s; (void)dp; (void)pfw; (void)dsw; (void)dk; (void)dv; (void)kf; (void)vf; (void)ks;
      // the next slot's operands first (into the registers slot i - 1 has finished with), then wait for this slot's: LDS returns in order
      if constexpr (i < 7 && !(AFM_PIPE_ABL & 4)) {
        reads(BB_, std::integral_constant<int, i + 1>{});
        if constexpr (AFM_PIPE_ABL & 64) __builtin_amdgcn_sched_barrier(0);     // (ablation: operands used before they have landed)
        else lgk_wait<3 + ((((i + 1) & 1) == 0 && DROP != DROP_NONE) ? 1 : 0)>();
      } else {
        if constexpr (AFM_PIPE_ABL & 64) __builtin_amdgcn_sched_barrier(0);
        else lgk_wait<0>();
      }
      if constexpr (!(AFM_PIPE_ABL & 1)) {
        const e16x8 ca = tr_join(clo[i & 1], chi[i & 1]);
        const e16x8 fa = __builtin_bit_cast(e16x8, abuf[i & 1]);
        constexpr int w0 = (i >> 2) * 4;        // words 0-3: scores 0-7 (k-slice 0), words 4-7: k-slice 1
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {       // every fragment read from LDS serves all of the wave's key blocks
          if constexpr (((i >> 1) & 1) == 0) {
            const e16x8 pf = __builtin_bit_cast(e16x8, (u32x4_){pfw[BC][kb][w0], pfw[BC][kb][w0 + 1], pfw[BC][kb][w0 + 2], pfw[BC][kb][w0 + 3]});
            dv[kb][i & 1] = mfma32(ca, pf, dv[kb][i & 1]);
          } else {
            const e16x8 df = __builtin_bit_cast(e16x8, (u32x4_){dsw[BC][kb][w0], dsw[BC][kb][w0 + 1], dsw[BC][kb][w0 + 2], dsw[BC][kb][w0 + 3]});
            dk[kb][i & 1] = mfma32(ca, df, dk[kb][i & 1]);
          }
          if constexpr (i & 1) dp[BA][kb] = mfma32(fa, vf[kb][ks], dp[BA][kb]);
          else s[BA][kb] = mfma32(fa, kf[kb][ks], s[BA][kb]);
        }
      }
      const f32x4 nd = ndb[(i >> 1) & 1];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
      if constexpr (AFM_PIPE_ABL & 2) {
        pfw[BB][kb][i] = __builtin_bit_cast(uint32_t, s[BB][kb][2 * i]) ^ __builtin_bit_cast(uint32_t, nd[0]); dsw[BB][kb][i] = __builtin_bit_cast(uint32_t, dp[BB][kb][2 * i + 1]);
      } else {   // two scores of B per key block
        constexpr int r0 = 2 * i, r1 = 2 * i + 1;
        const float p0 = fast_exp2(s[BB][kb][r0]), p1 = fast_exp2(s[BB][kb][r1]);
        float d0, d1, q0, q1;
        if constexpr (DROP == DROP_BITS) {
          // keep bit -> all-ones / zero (v_bfe_i32), then two bit selects: no compare, no condition register
          uint32_t m0, m1, e0, e1, z0, z1;
          asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m0) : "v"(word[kb]), "n"(ACC_ROW(r0)));
          asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m1) : "v"(word[kb]), "n"(ACC_ROW(r1)));
          asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(e0) : "v"(m0), "v"(dp[BB][kb][r0]), "v"(nd[2 * (i & 1)]));
          asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(e1) : "v"(m1), "v"(dp[BB][kb][r1]), "v"(nd[2 * (i & 1) + 1]));
          asm("v_and_b32 %0, %1, %2" : "=v"(z0) : "v"(m0), "v"(p0));
          asm("v_and_b32 %0, %1, %2" : "=v"(z1) : "v"(m1), "v"(p1));
          d0 = p0 * __builtin_bit_cast(float, e0); d1 = p1 * __builtin_bit_cast(float, e1);      // dS = P (keep ? scale dP - delta : -delta)
          q0 = __builtin_bit_cast(float, z0); q1 = __builtin_bit_cast(float, z1);
        } else {
          d0 = p0 * dp[BB][kb][r0]; d1 = p1 * dp[BB][kb][r1]; q0 = p0; q1 = p1;
        }
        uint32_t pw = cvt_pk2(q0, q1), dw = cvt_pk2(d0, d1);
        asm volatile("" : "+v"(pw), "+v"(dw));      // packed HERE (the compiler would otherwise carry the fp32 pairs into the next group)
        pfw[BB][kb][i] = pw; dsw[BB][kb][i] = dw;
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
  if (wave_all_masked) {     // 32 padded keys: the wave only keeps the ring and the barriers going (its outputs are zeros, below)
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
      f32x4 si[4], di[4];
      AFM_LDS_RD128(si[0], aLA, LS_OFF + 0);  AFM_LDS_RD128(si[1], aLA, LS_OFF + 32);
      AFM_LDS_RD128(si[2], aLA, LS_OFF + 64); AFM_LDS_RD128(si[3], aLA, LS_OFF + 96);
      AFM_LDS_RD128(di[0], aLA, DS_OFF + 0);  AFM_LDS_RD128(di[1], aLA, DS_OFF + 32);
      AFM_LDS_RD128(di[2], aLA, DS_OFF + 64); AFM_LDS_RD128(di[3], aLA, DS_OFF + 96);
      u32x4_ fq[4], fd[4];
      AFM_LDS_RD128(fq[0], aA[0], 0); AFM_LDS_RD128(fd[0], aA[0], IMG); AFM_LDS_RD128(fq[1], aA[1], 0); AFM_LDS_RD128(fd[1], aA[1], IMG);
      AFM_LDS_RD128(fq[2], aA[2], 0); AFM_LDS_RD128(fd[2], aA[2], IMG); AFM_LDS_RD128(fq[3], aA[3], 0); AFM_LDS_RD128(fd[3], aA[3], IMG);
      lgk_wait<0>();
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) { s[0][kb][4 * g4 + j] = si[g4][j] * -1.4426950408889634f; dp[0][kb][4 * g4 + j] = di[g4][j]; }
        }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          s[0][kb] = mfma32(__builtin_bit_cast(e16x8, fq[ks]), kf[kb][ks], s[0][kb]);
          dp[0][kb] = mfma32(__builtin_bit_cast(e16x8, fd[ks]), vf[kb][ks], dp[0][kb]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    // One tile = two groups with the tile's barrier between them.  The loop body is the same for every tile: the first group of tile
    // 0 runs its "C(-1, 1)" on tile 0's own fragments against zero operands (exact zeros added), the second group of the last tile
    // its "A(n, 0)" on whatever the next stage holds (nobody reads the result).
    int sB = 0;                                            // stage byte offset of tile j
    for (int j = 0; j < nlive; ++j) {
      const int sN = sB + STAGE == NS * STAGE ? 0 : sB + STAGE;      // stage of tile j + 1
      const int sP = sB == 0 ? (NS - 1) * STAGE : sB - STAGE;        // stage of tile j - 1 (and of tile j + 2)
      group(B0);                                           // B(j, 0) beside A(j, 1) and C(j - 1, 1)
      const int dC = j == 0 ? 0 : sB - sP;                 // the C unit is tile j's from here on
      move_stage(xa, dC); move_stage(xb, dC);
      attn_wait_vmcnt<0>();                                // tile j + 1 (the only one in flight)
      if (!(AFM_PIPE_ABL & 8)) __builtin_amdgcn_s_barrier();   // ... and every wave is past C(j - 1, 1), the last reader of tile j - 1's stage
      if (!(AFM_PIPE_ABL & 32) && j + 2 < nlive) {
        issue(j + 2, sP);
        if (((j + 2) & 3) == 0) issue_aux((j + 2) >> 2);   // needed from the second half of tile j + 1 on (A(j + 2, 0)'s initial values)
      }
      const int dX = auxo(j + 1) - auxo(j);
      move_stage(aA, sN - sB);                             // the A target is tile j + 1's first block
      aLA += (unsigned)dX;
      group(B1);                                           // B(j, 1) beside A(j + 1, 0) and C(j, 0)
      aLB += (unsigned)dX;
      aW += (unsigned)dX;
      sB = sN;
    }
    // ---- epilogue: C(last, 1); the transposed-read registers already point at the last tile
    {
      s16x4 lo[8], hi[8];
      static_for<0, 8>([&](auto I_) __attribute__((always_inline)) {
        constexpr int i = decltype(I_)::value;
        constexpr int img = ((i >> 1) & 1) ? 0 : IMG, SL = i >> 2, DLO = SL ? 3 : 0, DHI = SL ? 1 : 2;
        constexpr int olo = img + 4096 + 16 * SL * 128, ohi = olo + 8 * 128;
        (void)lo; (void)hi; (void)xa; (void)xb;
        if constexpr (i & 1) { AFM_TR_RDN(lo[i], xb[DLO], olo); AFM_TR_RDN(hi[i], xb[DHI], ohi); }
        else { AFM_TR_RDN(lo[i], xa[DLO], olo); AFM_TR_RDN(hi[i], xa[DHI], ohi); }
      });
      lgk_wait<0>();
      static_for<0, 8>([&](auto I_) __attribute__((always_inline)) {
        constexpr int i = decltype(I_)::value, w0 = (i >> 2) * 4;
        (void)pfw; (void)dsw; (void)dk; (void)dv;
        const e16x8 ca = tr_join(lo[i], hi[i]);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          if constexpr (((i >> 1) & 1) == 0)
            dv[kb][i & 1] = mfma32(ca, __builtin_bit_cast(e16x8, (u32x4_){pfw[1][kb][w0], pfw[1][kb][w0 + 1], pfw[1][kb][w0 + 2], pfw[1][kb][w0 + 3]}), dv[kb][i & 1]);
          else
            dk[kb][i & 1] = mfma32(ca, __builtin_bit_cast(e16x8, (u32x4_){dsw[1][kb][w0], dsw[1][kb][w0 + 1], dsw[1][kb][w0 + 2], dsw[1][kb][w0 + 3]}), dk[kb][i & 1]);
        }
      });
    }
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kmasked[kb]) {   // a padded key took no part in any softmax: its dK / dV rows are zero
#pragma unroll
      for (int i = 0; i < 16; ++i) { dk[kb][0][i] = 0.f; dk[kb][1][i] = 0.f; dv[kb][0][i] = 0.f; dv[kb][1][i] = 0.f; }
    }
    if (DROP != DROP_NONE) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { dv[kb][0][i] *= a.dd.scale16; dv[kb][1][i] *= a.dd.scale16; }
    }
  }
  store_rows(false);
}
