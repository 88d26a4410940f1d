// Fused short-query attention backward for gfx950 (included by afm_attn_mfma_impl.h inside namespace AFM_E16_NS).  Round 6.
//
// The decoder's cross-attention has Tq = 128 queries against Tk = 1 024 keys per (batch, head).  The two general backward kernels are
// furthest from any roof there (profiles/r06_c2_fp16_kernel_stats_by_grid.csv: dQ 105 us, dK/dV 209 us per layer at 52 / 69 GFLOP): the
// dK/dV kernel gives a workgroup 128 keys and TWO query tiles, so a workgroup's life is its prologue, and both kernels recompute S and dP
// (7 products of Tq x Tk x dh where the algorithm has 5) and run the softmax chain twice.
//
// Here ONE workgroup owns a (batch, head) with ALL its queries resident: 4 waves x 32 queries, the query on the lane as in the dQ kernel.
// Per 64-key tile:
//   phase A (the dQ kernel's tile body):  S^T = K Q^T, dP^T = V dO^T  ->  P, dS^T = P (D dP - delta)  ->  dQ^T += K^T dS^T;
//            the wave also writes its 32 rows of P_drop = D P and of dS, as 16-bit values, into two [128 q][64 key] LDS tiles laid out for
//            transposed reads (swizzle fsq_f below)
//   barrier
//   phase B: wave w takes the 32 x 32 block (d-block w >> 1, key-block w & 1) of dV^T = dO^T P_drop and of dK^T = Q^T dS over all 128
//            queries: 8 slices of 16 queries -- dO^T / Q^T fragments read once per kernel (13 of the 16 live in registers, 3 wait in LDS),
//            P / dS by ds_read_b64_tr_b16 from the tiles of phase A -- and hands on those dK / dV rows: every query of the head is in the
//            workgroup, so a key tile's dK / dV is complete after its own iteration (no atomics, no accumulation across tiles).  The rows
//            turn through 4 KB of the just-consumed ring stage (16 bytes per lane, four lanes to a row's 64 bytes) and are STORED at the top
//            of the next iteration, behind its wait for the ring (stores count in vmcnt: see the loop).
// 5 products, one softmax chain, K / V / Q / dO / keep bits read once.  Key tiles of nothing but padding never enter the loop; their dK / dV
// rows are written as zeros behind it (packed key rows: through attn_out_row, honouring the no-fill limit).
// The dO^T / Q^T operands of phase B do not depend on the key tile: the wave reads its 2 x 8 fragments from the images ONCE, before the loop.
// LDS: ring 2 x 16 KB (K as ONE dual-use image for the row and the transposed reads, V row image) + P, dS tiles 32 KB (the Q / dO images
// of the prologue lie in the same 32 KB: read out before the first tile's P is written) + 12 KB of parked fragments + masks = 76.3 KB: two
// workgroups per CU, 256 registers each.  (First form, 113 KB and one workgroup per CU: 279 us at the c2 shape against 356 for the two
// kernels; this form 198 us -- DESIGN.md 4.0r6 item 10b has every step.)  Dropout through the keep-bit tensor or off; Tq <= 128; a causal
// mask where Tq == Tk (the decoder's self-attention: its one or two key tiles are walked by every wave, masked above the diagonal);
// dense query rows (q_off unsupported), packed key rows supported.  No atomics: two runs give the same bits
// (tools/experiments/fsq_fuzz.py checks exactly that over random shapes).
//
// The P / dS tiles are [128 q][64 keys] e16, 128-byte rows, the 8-byte unit u of row r at unit  u ^ f(r),  f(r) = 8 ((r >> 1) & 1) | ((r >> 2) & 7):
// phase A's lane (query r, half h) stores 8 bytes per register group -- 32 consecutive rows of one half-wave land on 32 different 8-byte
// bank pairs -- and phase B's transposed reads (4 rows x 64 bytes per half-wave, f constant but for the (r >> 1) & 1 bit) stay conflict-free.
__device__ __forceinline__ int fsq_f(int row) { return (((row >> 1) & 1) << 6) | (((row >> 2) & 7) << 3); }
// one A / B fragment (16-row slice at byte offset OFF of a transposed-read image), complete when the statement ends
template <int OFF> __device__ __forceinline__ e16x8 fsq_frag_sync(unsigned addr) {
  s16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(lo), "=&v"(hi) : "v"(addr), "n"(OFF), "n"(OFF + 1024) : "memory");
  return tr_join(lo, hi);
}
struct FsqPS { s16x4 plo, phi, slo, shi; };
// 16-query slice SL (0 .. 7) of the P and dS tiles (dS 16 KB behind P): x[2 (SL & 1) + hi] are the lane's four address forms
template <int SL> __device__ __forceinline__ FsqPS fsq_ps_issue(const unsigned (&x)[4]) {
  FsqPS r;
  r.plo = tr_rd<SL * 2048>(x[2 * (SL & 1)]);
  r.phi = tr_rd<SL * 2048 + 1024>(x[2 * (SL & 1) + 1]);
  r.slo = tr_rd<16384 + SL * 2048>(x[2 * (SL & 1)]);
  r.shi = tr_rd<16384 + SL * 2048 + 1024>(x[2 * (SL & 1) + 1]);
  return r;
}
template <int SL>
__device__ __forceinline__ void fsq_phase_b(const unsigned (&x)[4], const e16x8 (&dT)[5], const e16x8* park, const e16x8 (&qT)[8], const FsqPS cur, f32x16& dv, f32x16& dk) {
  e16x8 dfrag;
  if constexpr (SL >= 5) dfrag = park[(SL - 5) * 256]; else dfrag = dT[SL];      // (slices 5-7 of dO^T wait in LDS: see the kernel)
  if constexpr (SL < 7) {
    const FsqPS nxt = fsq_ps_issue<SL + 1>(x);
    tr_wait<4>();
    dv = mfma32(dfrag, tr_join(cur.plo, cur.phi), dv);
    dk = mfma32(qT[SL], tr_join(cur.slo, cur.shi), dk);
    fsq_phase_b<SL + 1>(x, dT, park, qT, nxt, dv, dk);
  } else {
    tr_wait<0>();
    dv = mfma32(dfrag, tr_join(cur.plo, cur.phi), dv);
    dk = mfma32(qT[SL], tr_join(cur.slo, cur.shi), dk);
  }
}
// ABL (AFM_ATTN_ABLATIONS builds, timing only -- results wrong by construction): 1 no dK / dV stores, 2 no phase B (reads and products),
// 4 no P / dS stores to LDS, 8 no dQ product (transposed K reads and MFMAs), 16 no exp2 / dropout / mask work on the scores
template <int DROP, int ABL = 0>
__global__ __launch_bounds__(256, 2) void k_attn_bwd_fsq(AttnM a, const e16* __restrict__ Q, const e16* __restrict__ K,
                                                          const e16* __restrict__ V, const e16* __restrict__ O,
                                                          const e16* __restrict__ dO, const float* __restrict__ lse,
                                                          float* __restrict__ delta, e16* __restrict__ dQ, e16* __restrict__ dK,
                                                          e16* __restrict__ dV) {
  static_assert(DROP == DROP_NONE || DROP == DROP_BITS, "keep-bit tensor or no dropout");
  constexpr int IMG = KT * DH * 2;            // one [64][64] e16 image
  constexpr int STAGE = 2 * IMG;              // K dual-use image, V row image
  constexpr int PIMG = RS * STAGE, QIMG = PIMG, DOIMG = PIMG + 2 * IMG, PARK = PIMG + 4 * IMG, MASK = PARK + 3 * 4096;      // (dS tile: PIMG + 2 IMG; Q / dO images: prologue only)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned long long* maskw = (unsigned long long*)(lds + MASK);
  const int t = threadIdx.x, lane = t & 63, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const AttnBlock blk_ = attn_block(a.H, a.B, 1);
  const int hd = blk_.hd, b = blk_.b;
  const int q0 = w * 32;
  const int q = q0 + (lane & 31);
  const int qc = q < a.Tq ? q : a.Tq - 1;
  // (the packed offsets come out of memory: told to be wave-uniform, the tile bases below stay in scalar registers.  As lane values they
  // were spilled, and the reload of a spilled base in front of every LDS-DMA piece came with `s_waitcnt vmcnt(0)`: the ring's pieces
  // went out one full memory latency apart)
  const int64_t rq = (int64_t)b * a.Tq, rk = (int64_t)__builtin_amdgcn_readfirstlane((int)attn_row0(a.k_off, b, a.Tk));
  const int lim_k = __builtin_amdgcn_readfirstlane(attn_slot(a.k_off, b, a.Tk));
  const e16* Kb = K + rk * a.ldk + hd * DH;
  const e16* Vb = V + rk * a.ldv + hd * DH;
  const e16* Qb = Q + rq * a.ldq + hd * DH;
  const e16* Db = dO + rq * a.ldo + hd * DH;
  // Q / dO fragments (B operands of S^T = K Q^T and dP^T = V dO^T) and delta_i = dO_i . O_i, as in the dQ kernel
  e16x8 qf[4], dof[4];
  float dl = 0.f;
  {
    const e16* qp = Qb + (int64_t)qc * a.ldq + 8 * h;
    const e16* dop = Db + (int64_t)qc * a.ldo + 8 * h;
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
  if (q < a.Tq && h == 0) delta[lrow] = -dl;
  const float L = lse[lrow];
  const float nL2 = (L == INFINITY || q >= a.Tq) ? -INFINITY : -L * 1.4426950408889634f;      // (rows past Tq take no part: their scores start, and stay, at -inf)
  // dO stays as loaded (the dQ kernel multiplies it by the dropout scale and rounds again: 2^-11 of |dP| left in dS, visible where the
  // softmax is one-hot and dS cancels to nothing).  Here the chain starts from -delta / scale and the scale rides on P:
  // dS = (scale P) (keep ? dP - delta / scale : -delta / scale)
  const float ndl = DROP != DROP_NONE ? -dl / a.dd.scale16 : -dl;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[s][j] = (e16)((float)qf[s][j] * a.scale_log2);
  f32x16 dq[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }

  const int ntiles = (a.Tk + KT - 1) / KT;
  build_mask_words(maskw, a.key_pad, b, a.Tk, ntiles, w, lane);
  int* const tl = (int*)(maskw + ntiles) + 1;      // key tiles with at least one real key
  __syncthreads();
  build_tile_list(tl, a.key_pad ? maskw : nullptr, 0, ntiles, w, lane);
  __syncthreads();   // plain loads / the delta store retired before the first LDS-DMA piece
  const int nlive = __builtin_amdgcn_readfirstlane(tl[-1]);
  // Every workgroup starts its walk over the key tiles somewhere else.  The resident workgroups advance in step, and with the walk starting
  // at tile 0 everywhere they all read and write the same 128 KB window of their sample's rows at the same time -- samples lie a power of
  // two apart (S rows of 2 KB), so that window is the same few memory channels for all of them.
  const int rot = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 5u + (blockIdx.x >> 3)) % (unsigned)nlive));
  // the whole head's Q and dO as transposed-read images (two images of 64 query rows each): A operands of phase B
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      dma_piece_s<1>(lds + QIMG + m * IMG, Qb, a.ldq, 64 * m, a.Tq, w + 4 * u, lane);
      dma_piece_s<1>(lds + DOIMG + m * IMG, Db, a.ldo, 64 * m, a.Tq, w + 4 * u, lane);
    }
  auto issue = [&](int j) {
    unsigned char* st = lds + (j % RS) * STAGE;
    const int kt = tl[j + rot < nlive ? j + rot : j + rot - nlive];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      dma_piece_s<2>(st, Kb, a.ldk, kt * KT, a.Tk, w + 4 * u, lane);
      dma_piece_s<0>(st + IMG, Vb, a.ldv, kt * KT, a.Tk, w + 4 * u, lane);
    }
  };
#pragma unroll
  for (int s = 0; s < RS - 1; ++s)
    if (s < nlive) issue(s);
  __builtin_assume(nlive >= 1);
  // the lane's row of the P / dS tiles; register group g4 of block blk = keys 32 blk + 8 g4 + 4 h .. + 3 = bytes 64 blk + 16 g4 + 8 h .. + 7
  const int prow = 32 * w + (lane & 31);
  unsigned char* const Pw = lds + PIMG + prow * 128;
  const int pcol = (8 * h) ^ fsq_f(prow);
  // phase B: the lane's transposed-read addresses into the P tile (slice 0; the slice, the +8 rows and the dS tile are immediates)
  unsigned xps[4];
  {
    const int g = lane >> 4, qq = (lane >> 2) & 3, p4 = lane & 3;
    const unsigned rb = (unsigned)(uintptr_t)(lds + PIMG) + (4 * h + qq) * 128;
    const int cp = (64 * (w & 1) + 32 * (g & 1) + 8 * p4) ^ (64 * (qq >> 1)) ^ (8 * h);
#pragma unroll
    for (int i = 0; i < 4; ++i) xps[i] = rb + (cp ^ (16 * i));
  }
  // dO^T / Q^T fragments of this wave's d-block over the 128 queries
  e16x8 dT[5], qT[8];
  attn_wait_vmcnt<4>();          // the image pieces are older than the first stage's four
  __builtin_amdgcn_s_barrier();
  // (read AND waited for inside one asm statement each: the compiler does not know these reads are asynchronous, and a fragment it decides
  // to spill would go to scratch before it has arrived -- seen with the first form of this prologue)
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    unsigned x0, x1;
    tr_lane_addr(lds + DOIMG + m * IMG, lane, x0, x1);
    const unsigned aD = (w >> 1) ? x1 : x0;
    tr_lane_addr(lds + QIMG + m * IMG, lane, x0, x1);
    const unsigned aQ = (w >> 1) ? x1 : x0;
    e16x8* const park = (e16x8*)(lds + PARK) + t;      // (slices 5-7: straight to their place in LDS, below)
    if (m == 0) { dT[0] = fsq_frag_sync<0>(aD); dT[1] = fsq_frag_sync<2048>(aD); dT[2] = fsq_frag_sync<4096>(aD); dT[3] = fsq_frag_sync<6144>(aD); }
    else { dT[4] = fsq_frag_sync<0>(aD); park[0] = fsq_frag_sync<2048>(aD); park[256] = fsq_frag_sync<4096>(aD); park[512] = fsq_frag_sync<6144>(aD); }
    qT[4 * m] = fsq_frag_sync<0>(aQ); qT[4 * m + 1] = fsq_frag_sync<2048>(aQ); qT[4 * m + 2] = fsq_frag_sync<4096>(aQ); qT[4 * m + 3] = fsq_frag_sync<6144>(aQ);
  }
  // Three of the sixteen fragments wait in LDS (16 bytes per lane each, 12 KB) and come back at the start of every phase B: phase A is 12
  // registers over the 256 a wave has at two workgroups per CU, and what the compiler spilled instead were these same fragments -- to
  // scratch, three dependent reloads per tile, each behind an `s_waitcnt vmcnt(0)`.
  const e16x8* const park = (const e16x8*)(lds + PARK) + t;
  const unsigned kt0 = tr_dual_t0(lane);
  const int64_t fill_end = attn_fill_end(a.nofill, a.k_off, a.B);
  const int kblk = w & 1, dblk = w >> 1;
  // dK / dV rows of a tile leave at the TOP of the next iteration, right behind its wait for the ring: on this target stores count in
  // vmcnt with the loads, so the ring's `vmcnt(0)` would otherwise sit on the write acknowledgements of stores issued just before it
  // (measured: phase B with its stores was half of the kernel).  Issued there, they have a whole tile to complete.
  uint4 pdk[2], pdv[2];
  int64_t prow_[2] = {-1, -1};
  auto flush = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (prow_[i] >= 0) {
        *(uint4*)(dK + prow_[i] * a.lddk + hd * DH + 32 * dblk + 8 * (lane & 3)) = pdk[i];
        *(uint4*)(dV + prow_[i] * a.lddv + hd * DH + 32 * dblk + 8 * (lane & 3)) = pdv[i];
      }
  };
  for (int j = 0; j < nlive; ++j) {
    const int kt = __builtin_amdgcn_readfirstlane(tl[j + rot < nlive ? j + rot : j + rot - nlive]);
    const int kb = kt * KT;
    if (nlive - 1 - j >= RS - 2) attn_wait_vmcnt<4 * (RS - 2)>(); else attn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();     // tile j has landed; every wave is done with the P / dS tiles of tile j - 1
    if (j > 0) flush();
    if (j + RS - 1 < nlive) issue(j + RS - 1);
    if (maskw[kt] == ~0ull) continue;   // (workgroup-uniform: the list's fallback tile of an all-padded head; zero rows written behind the loop)
    const unsigned char* Krow = lds + (j % RS) * STAGE;
    const unsigned char* Vrow = Krow + IMG;
    const unsigned long long mword = maskw[kt];
    const unsigned long long pad = mword >> (4 * h);
    unsigned xa[4], xb[4];      // transposed-read address registers of this stage's K image
#pragma unroll
    for (int dd_ = 0; dd_ < 4; ++dd_) { xa[dd_] = (unsigned)(uintptr_t)Krow + (kt0 ^ (dd_ << 4)); xb[dd_] = (unsigned)(uintptr_t)Krow + (kt0 ^ (dd_ << 4) ^ 64); }
    KeepMasks km[2];
    if (DROP == DROP_BITS) {
      const unsigned long long* kbp = bits_block(a, b * a.H + hd, q0 >> 5, 2 * kt);
      keep_masks_issue(km[0], kbp);
      keep_masks_issue(km[1], kbp + 16);
    }
    // ---- phase A
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = nL2; dp[i] = ndl; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = mfma32(frag_row_dual(Krow, 32 * blk, ks, lane), qf[ks], s);
        dp = mfma32(frag_row(Vrow, 32 * blk, ks, lane), dof[ks], dp);
      }
      if (DROP == DROP_BITS && !(ABL & 16)) drop_select_masks(dp, km[blk], ndl);
      if ((mword != 0ull || (a.causal && kb + KT - 1 > q0)) && !(ABL & 16)) {   // wave-uniform: tile has masked keys (padding / above the diagonal)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ko = 32 * blk + ACC_ROW(r);
          bool msk = (pad >> ko) & 1ull;
          if (a.causal) msk = msk || (kb + ko + 4 * h > q);
          s[r] = msk ? -INFINITY : s[r];
        }
      }
      f32x16 pd;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = (ABL & 16) ? s[r] : fast_exp2(s[r]);   // masked keys and rows past Tq: exp2(-inf) = 0
        pd[r] = DROP != DROP_NONE ? p * a.dd.scale16 : p;
        s[r] = pd[r] * dp[r];   // dS^T = P (D dP - delta)
      }
      if (DROP == DROP_BITS && !(ABL & 16)) drop_select_masks(pd, km[blk], 0.f);      // P_drop = D P
      if (!(ABL & 4))
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        unsigned char* const pp = Pw + (pcol ^ (64 * blk + 16 * g4));
        *(e16x4*)pp = (e16x4){(e16)pd[4 * g4], (e16)pd[4 * g4 + 1], (e16)pd[4 * g4 + 2], (e16)pd[4 * g4 + 3]};
        *(e16x4*)(pp + 2 * IMG) = (e16x4){(e16)s[4 * g4], (e16)s[4 * g4 + 1], (e16)s[4 * g4 + 2], (e16)s[4 * g4 + 3]};
      }
      {
        const e16x8 ds0 = cvt8(s, 0), ds1 = cvt8(s, 1);
#define AFM_FSQ_DQ(B32)                                                                                   \
        {                                                                                                 \
          const TrQuad k0q = tr_quad_dual<B32, 0>(xa, xb), k1q = tr_quad_dual<B32, 1>(xa, xb);            \
          tr_wait<4>();                                                                                   \
          dq[0] = mfma32(tr_join(k0q.lo0, k0q.hi0), ds0, dq[0]);                                          \
          dq[1] = mfma32(tr_join(k0q.lo1, k0q.hi1), ds0, dq[1]);                                          \
          tr_wait<0>();                                                                                   \
          dq[0] = mfma32(tr_join(k1q.lo0, k1q.hi0), ds1, dq[0]);                                          \
          dq[1] = mfma32(tr_join(k1q.lo1, k1q.hi1), ds1, dq[1]);                                          \
        }
        if (ABL & 8) { dq[0][0] += (float)ds0[0] + (float)ds1[0]; } else
        if (blk == 0) AFM_FSQ_DQ(0) else AFM_FSQ_DQ(32 * 128)
#undef AFM_FSQ_DQ
      }

    }
    __syncthreads();      // the head's P / dS tiles of this key tile are complete
    // ---- phase B: dV^T (d-block dblk, key-block kblk) = dO^T P_drop, dK^T = Q^T dS over the 128 queries
    f32x16 dv, dk;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dv[i] = 0.f; dk[i] = 0.f; }
    if (!(ABL & 2)) fsq_phase_b<0>(xps, dT, park, qT, fsq_ps_issue<0>(xps), dv, dk);
    else { dv[0] = (float)dT[0][0] + (float)dT[4][3]; dk[0] = (float)qT[0][1] + (float)qT[7][2]; }
    if (!(ABL & 1) || j == 0) {
      // The accumulators hold dK^T / dV^T with the key on the lane: stored from there, a lane would write 8 bytes into each of four 64-byte
      // row pieces (measured: the stores were over half of the kernel).  They turn through LDS instead -- this wave's 4 KB of the ring
      // stage phase A has just finished with (the next DMA into it is issued behind the next barrier) -- and leave as 16 bytes per lane,
      // four lanes to a row's 64 bytes.  [32 keys][64 bytes], 16-byte piece c of key k at piece c ^ ((k >> 2) & 3).
      unsigned char* const tb = lds + (j % RS) * STAGE + w * 4096;
      {
        const int kl = lane & 31;
        unsigned char* const wp = tb + kl * 64 + 8 * h;
        const int sw = (kl >> 2) & 3;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          *(e16x4*)(wp + ((g4 ^ sw) << 4)) = (e16x4){(e16)(dk[4 * g4] * a.scale), (e16)(dk[4 * g4 + 1] * a.scale), (e16)(dk[4 * g4 + 2] * a.scale), (e16)(dk[4 * g4 + 3] * a.scale)};
          *(e16x4*)(wp + 2048 + ((g4 ^ sw) << 4)) = (e16x4){(e16)dv[4 * g4], (e16)dv[4 * g4 + 1], (e16)dv[4 * g4 + 2], (e16)dv[4 * g4 + 3]};
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kl = (lane >> 2) + 16 * i, c = lane & 3;
        const int key = kb + 32 * kblk + kl;
        const bool own = key < lim_k;
        const int64_t row = attn_out_row(a.k_off, a.B, b, a.Tk, rk, key, lim_k);
        const unsigned char* rp = tb + kl * 64 + ((c ^ ((kl >> 2) & 3)) << 4);
        uint4 x = *(const uint4*)rp, y = *(const uint4*)(rp + 2048);
        if (!own) { x = make_uint4(0u, 0u, 0u, 0u); y = x; }
        pdk[i] = x; pdv[i] = y;
        prow_[i] = (row >= 0 && (own || row < fill_end)) ? row : -1;
      }
    }
  }
  flush();
  // dQ rows (dense query rows)
  if (q < a.Tq) {
    e16* dqp = dQ + (rq + q) * a.lddq + hd * DH + 4 * h;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        e16x4 v = {(e16)(dq[db][4 * g4 + 0] * a.scale), (e16)(dq[db][4 * g4 + 1] * a.scale),
                    (e16)(dq[db][4 * g4 + 2] * a.scale), (e16)(dq[db][4 * g4 + 3] * a.scale)};
        *(e16x4*)(dqp + 32 * db + 8 * g4) = v;
      }
  }
  // key tiles of nothing but padding were never loaded: their dK / dV rows are zeros (wave w: keys 16 w .. + 15 of the tile, lane >> 4 = the
  // 16-column quarter of the head's 64)
  {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int kt = 0; kt < ntiles; ++kt) {
      if (maskw[kt] != ~0ull) continue;
      const int key = kt * KT + 16 * w + (lane & 15);
      const int64_t row = attn_out_row(a.k_off, a.B, b, a.Tk, rk, key, lim_k);
      if (row >= 0 && (key < lim_k || row < fill_end)) {
        e16* dkp = dK + row * a.lddk + hd * DH + 16 * (lane >> 4);
        e16* dvp = dV + row * a.lddv + hd * DH + 16 * (lane >> 4);
        *(uint4*)dkp = z; *(uint4*)(dkp + 8) = z;
        *(uint4*)dvp = z; *(uint4*)(dvp + 8) = z;
      }
    }
  }
}
