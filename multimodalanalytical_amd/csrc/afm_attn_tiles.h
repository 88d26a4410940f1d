// LDS tile images, transposed-read helpers, the LDS-DMA piece and the dropout block of the MFMA flash-attention
// kernels (afm_attn_mfma.hip: single e16 pass; afm_attn_x3.hip: split-pair operands, three passes per product).
#pragma once
#include <type_traits>
#include "afm_common.h"

#define DH 64
#define KT 64   // keys (or queries, in the dK/dV kernel) per LDS tile

struct AttnM {
  int B, H, Tq, Tk;
  int ldq, ldk, ldv, ldo;
  int lddq, lddk, lddv;
  int causal;
  float scale_log2;  // scale * log2(e)
  float scale;
  const uint8_t* key_pad;
  DropDev dd;
  unsigned long long* bits;   // keep-bit tensor (include/afm_hip.h: afm_attn_shape.drop_bits), null = re-hash
  int nq32, nk32;
  int qskip;   // backward, self-attention: query rows at padded positions carry zero dO (afm_attn_shape.reserved & 64): skipped, exactly
  const int32_t* q_off;   // afm_attn_shape.q_off / k_off (B + 1 entries, nullable): PACKED rows -- sample b's query / key rows start at row
  const int32_t* k_off;   // off[b] of Q, O, dO, dQ / K, V, dK, dV, its slot is off[b + 1] - off[b] rows (a multiple of 32)
  int nofill;             // forward (afm_attn_shape.reserved bit 17): blocks without rows of their own write nothing (the caller's O is finite there)
};

// rows of sample b's slot on a packed side (a multiple of 32: a wave's rows never straddle samples), else T
__device__ __forceinline__ int attn_slot(const int32_t* off, int b, int T) { return off ? off[b + 1] - off[b] : T; }
// first row of the dead tail that belongs to sample b's rows beyond its slot: in-sample row r >= slot <-> tail row attn_tail0 + (r - slot)
__device__ __forceinline__ int64_t attn_tail0(const int32_t* off, int B, int b, int T) { return (int64_t)off[B] + ((int64_t)b * T - off[b]); }
// Without fills (AttnM.nofill): tail rows at or beyond this row are not written.  The end of the slots (off[B]) is a multiple of 32, the
// kernels that work on 64-row blocks (LayerNorm, the weight gradients' k-steps) treat the block around it as live and READ its upper half:
// those <= 32 tail rows always get their zeros.
__device__ __forceinline__ int64_t attn_fill_end(int nofill, const int32_t* off, int B) {
  return (nofill && off) ? (((int64_t)off[B] + 63) & ~(int64_t)63) : ((int64_t)1 << 62);
}
// the row an output of in-sample row r goes to: the sample's own row, or (packed, r beyond the slot) its row of the dead tail; -1: none (r >= T unpacked)
__device__ __forceinline__ int64_t attn_out_row(const int32_t* off, int B, int b, int T, int64_t row0, int r, int lim) {
  if (r < lim) return row0 + r;
  return (off && r < T) ? attn_tail0(off, B, b, T) + (r - lim) : -1;
}
// first row of sample b on the query / key side
__device__ __forceinline__ int64_t attn_row0(const int32_t* off, int b, int T) { return off ? (int64_t)off[b] : (int64_t)b * T; }
// Packed rows: 128-row block `blk128` of sample b lies beyond the sample's slot.  Its workgroup has nothing to compute; it writes ZEROS to
// the matching rows of the dead tail [off[B], B * T) instead -- the in-sample rows beyond the slots and the rows of the tail are equally
// many (off[B] + sum_b (T - slot_b) = B * T), in-sample row r of sample b <-> tail row attn_tail0(b) + (r - slot_b) is the bijection -- so
// every row of the output holds a finite value whoever loads it.  A block that is only PARTLY beyond the slot (slots are multiples of 32
// rows, blocks 128) runs as usual; its waves whose 32 rows lie beyond the slot must not store there (those are the next sample's rows):
// they zero their share of the tail through the same bijection (the kernels' final stores).
__device__ __forceinline__ bool attn_tail_block(const int32_t* off, int B, int b, int blk128, int T, int64_t& row0) {
  if (!off) return false;
  const int slot = off[b + 1] - off[b];
  if (blk128 * 128 < slot) return false;
  row0 = (int64_t)off[B] + ((int64_t)b * T - off[b]) + (blk128 * 128 - slot);
  return true;
}

typedef __attribute__((ext_vector_type(4))) short s16x4;

// 16-byte load of an operand this kernel reads ONCE per workgroup (the query-side rows of the forward / dQ kernels, the key-side rows
// of the dK/dV kernel).  AFM_ATTN_NT=1 makes these loads nontemporal -- measured in round 5 and LEFT OFF: one
// box, alternating processes, c2 encoder shape, fp16: forward 0.431 / 0.437 -> 0.440 / 0.449 ms, dQ 0.535 / 0.533 -> 0.566 / 0.570,
// dK/dV 0.682 / 0.683 -> 0.710 / 0.713; step 3 192 -> 3 134 samples/s with the GEMM-side loads (which DO pay) in both.  The rows a
// kernel reads "once" are re-read by the next kernel of the layer (Q and dO by the dK/dV kernel right behind the dQ kernel): marked
// evict-first they leave the memory-side cache too.
#ifndef AFM_ATTN_NT
#define AFM_ATTN_NT 0
#endif
__device__ __forceinline__ e16x8 ld8_once(const e16* p) {
  typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
#if AFM_ATTN_NT
  return __builtin_bit_cast(e16x8, __builtin_nontemporal_load((const u32x4_nt*)p));
#else
  return *(const e16x8*)p;
#endif
}

__device__ __forceinline__ f32x16 mfma32(e16x8 a, e16x8 b, f32x16 c) {
  return mfma32_raw(a, b, c);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// LDS images of a [64 rows][64 cols] e16 tile (128-byte rows, 8 chunks of 16 bytes):
//   "row image": read by rows with ds_read_b128 (lane = row, 32 rows x one chunk per half-wave)
//   "tr image" : read transposed with ds_read_b64_tr_b16 (4 rows x 16 columns per 16 lanes)
__device__ __forceinline__ int img_row(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int img_tr(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4); }

// accumulator register r of a 32x32 tile <-> row (r&3) + 8*(r>>2) + 4*(lane>>5)
#define ACC_ROW(r) (((r) & 3) + 8 * ((r) >> 2))

// A-operand fragment "X^T slice": element j of lane-half h <-> tile row R0 + 4h + (j&3) + 8*(j>>2),
// column C0 + (lane&31); two transposed reads.  R0 = first row of the 16-row k-slice.
//
// The reads are inline asm, not the ds_read_tr builtin: behind the builtin hipcc puts
// `s_waitcnt vmcnt(0)` in front of the first transposed read of every tile, which drains the LDS-DMA
// ring (the prefetch of the next tile) in the middle of the tile.  The asm reads are invisible to the
// compiler's counters, so the consumer waits by hand: tr_wait<N>() = s_waitcnt lgkmcnt(N) + scheduling
// fence (cdna_hip_programming.md 5.7 form iii).  LDS returns in order, so lgkmcnt(N) with N younger
// reads outstanding is enough for the older ones (a compiler-issued LGKM op in between only makes the
// wait stricter).  One quad = the two 32-column halves (db = 0, 1) of a 16-row slice.
struct TrQuad { s16x4 lo0, hi0, lo1, hi1; };
#define AFM_TR_RD(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
// per-lane byte addresses (db = 0 / 1) of slice R0 = 0 inside a tr image; slice R0 adds R0*128 (immediate)
__device__ __forceinline__ void tr_lane_addr(const unsigned char* img, int lane, unsigned& a0, unsigned& a1) {
  const int g = lane >> 4, qq = (lane >> 2) & 3, p = lane & 3;
  const int row = 4 * (g >> 1) + qq, bsw = (qq >> 1) & 1;        // ((R0 + row) >> 1) & 1 for R0 % 16 == 0
  const int c = 2 * (g & 1) + (p >> 1), sub = (p & 1) << 3;
  const unsigned base = (unsigned)(uintptr_t)img + row * 128 + sub;
  a0 = base + ((4 * bsw + c) << 4);
  a1 = base + ((4 * (1 ^ bsw) + c) << 4);
}
__device__ __forceinline__ TrQuad tr_issue(unsigned a0, unsigned a1, int R0) {
  TrQuad q;
  switch (R0) {   // R0 is a constant after unrolling; the offsets must be literals for the asm
    case 0:  AFM_TR_RD(q.lo0, a0, 0);    AFM_TR_RD(q.hi0, a0, 1024); AFM_TR_RD(q.lo1, a1, 0);    AFM_TR_RD(q.hi1, a1, 1024); break;
    case 16: AFM_TR_RD(q.lo0, a0, 2048); AFM_TR_RD(q.hi0, a0, 3072); AFM_TR_RD(q.lo1, a1, 2048); AFM_TR_RD(q.hi1, a1, 3072); break;
    case 32: AFM_TR_RD(q.lo0, a0, 4096); AFM_TR_RD(q.hi0, a0, 5120); AFM_TR_RD(q.lo1, a1, 4096); AFM_TR_RD(q.hi1, a1, 5120); break;
    default: AFM_TR_RD(q.lo0, a0, 6144); AFM_TR_RD(q.hi0, a0, 7168); AFM_TR_RD(q.lo1, a1, 6144); AFM_TR_RD(q.hi1, a1, 7168); break;
  }
  return q;
}
template <int N> __device__ __forceinline__ void tr_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ e16x8 tr_join(s16x4 lo, s16x4 hi) {
  const e16x4 l = __builtin_bit_cast(e16x4, lo), h = __builtin_bit_cast(e16x4, hi);
  return (e16x8){l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
}
// A-operand fragment by rows: lane holds tile[R0 + (lane&31)][16*s + 8*(lane>>5) .. +7]
__device__ __forceinline__ e16x8 frag_row(const unsigned char* img, int R0, int s, int lane) {
  return *(const e16x8*)(img + img_row(R0 + (lane & 31), 2 * s + (lane >> 5)));
}
__device__ __forceinline__ e16x8 cvt8(const f32x16& x, int s) {
  return (e16x8){(e16)x[8 * s + 0], (e16)x[8 * s + 1], (e16)x[8 * s + 2], (e16)x[8 * s + 3],
                  (e16)x[8 * s + 4], (e16)x[8 * s + 5], (e16)x[8 * s + 6], (e16)x[8 * s + 7]};
}
// The same as four PACKED conversions (v_cvt_pk_*: half an instruction per score) whatever precedes them.  The empty asm makes the
// inputs opaque: where a dropout select precedes the conversion hipcc otherwise converts each value alone, selects on the 16-bit
// result and re-packs (one instruction per score more).  Used by the dK/dV kernels (-2.5 %) and, since the two-level dropout hash freed
// the registers it needs at four waves per SIMD, by the forward.
__device__ __forceinline__ e16x8 cvt8_pk(const f32x16& x, int s) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef e16 e16x2_ __attribute__((ext_vector_type(2)));
  e16x8 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = x[8 * s + 2 * i], b = x[8 * s + 2 * i + 1];
    asm("" : "+v"(a), "+v"(b));
    const e16x2_ h = __builtin_convertvector((f32x2_){a, b}, e16x2_);
    o[2 * i] = h[0]; o[2 * i + 1] = h[1];
  }
  return o;
}

// stage a [64][64] e16 tile of a (rows x ld) matrix: thread t -> rows t>>3 and 32 + t>>3, chunk t&7
struct Stage2 { uint4 v[2]; };
__device__ __forceinline__ Stage2 stage_load(const e16* base, int ld, int row0, int nrows, int t) {
  Stage2 s;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r = row0 + (t >> 3) + 32 * i;
    r = r < nrows ? r : nrows - 1;  // clamped rows are masked out by the caller
    s.v[i] = *(const uint4*)(base + (int64_t)r * ld + (t & 7) * 8);
  }
  return s;
}
template <bool TR>
__device__ __forceinline__ void stage_store(unsigned char* img, const Stage2& s, int t) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (t >> 3) + 32 * i;
    *(uint4*)(img + (TR ? img_tr(r, t & 7) : img_row(r, t & 7))) = s.v[i];
  }
}

// dropout on a 32x32 score block held transposed (rows = keys in registers, query on the lane): the lane's ROW HASH (afm_row_hash of
// its query's score-matrix row, made once per kernel) plus the key pair's stride offset goes through the four-instruction pair mix
// (afm_common.h, two-level stream of round 4); keys (ACC_ROW(r), +1) take the low / high 16 bits.
__device__ __forceinline__ uint32_t hash_pair32(uint32_t rowhash_plus_pair) { return afm_pair_mix(rowhash_plus_pair); }
// rowhash + stride * (first pair of the lane's keys in the block): key0 a multiple of 32, 4 h even
__device__ __forceinline__ uint32_t pair_base(uint32_t rowhash, int key0, int h) { return rowhash + afm_pair_offset((uint32_t)(key0 + 4 * h) >> 1); }
#define AFM_PAIR_OFF(R) ((uint32_t)(ACC_ROW(R) >> 1) * AFM_PAIR_STRIDE)      // compile-time constant per register pair
__device__ __forceinline__ void drop_block(const DropDev& dd, uint32_t rowhash, int key0, int h, f32x16& x) {
  const uint32_t base = pair_base(rowhash, key0, h);
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const uint32_t hsh = hash_pair32(base + AFM_PAIR_OFF(r));
    x[r] = (hsh & 0xFFFFu) >= dd.thresh16 ? x[r] : 0.f;
    x[r + 1] = (hsh >> 16) >= dd.thresh16 ? x[r + 1] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------ LDS-DMA tile ring
// K / V / Q / dO tiles go HBM -> LDS with global_load_lds_dwordx4 (1-KiB pieces = 8 tile rows), the
// image swizzles applied on the SOURCE chunk index, into a ring of RS stages with RS-1 tiles in
// flight (counted vmcnt + raw s_barrier, as in the GEMM ring): the tile loads no longer sit on the
// critical path of each iteration.
template <int N> __device__ __forceinline__ void attn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Two stages since round 3 (three before): with three, the dQ kernel's ring (3 images x 8 KB x RS) left room for only two
// workgroups per CU although its registers allow three (0.59 -> 0.51 ms at the c2 shape), and the forward can run four (below).
#ifndef RS
#define RS 2
#endif
// one piece: rows [8*pi, 8*pi+8) of the 64-row tile starting at global row `row0` of `base`
template <bool TR>
__device__ __forceinline__ void dma_piece(unsigned char* img, const e16* base, int ld, int row0, int nrows,
                                          int pi, int lane) {
  const int r = 8 * pi + (lane >> 3), slot = lane & 7;
  const int chunk = TR ? (slot ^ (((r >> 1) & 1) << 2)) : (slot ^ ((r >> 1) & 7));
  int gr = row0 + r;
  gr = gr < nrows ? gr : nrows - 1;   // clamped rows are masked out by the caller
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (int64_t)gr * ld + chunk * 8),
                                   (__attribute__((address_space(3))) void*)(img + pi * 1024), 16, 0, 0);
}
// key-mask words (bit i of word t = key 64 t + i is padded or past Tk) for one batch row, in LDS
__device__ __forceinline__ void build_mask_words(unsigned long long* maskw, const uint8_t* key_pad, int b, int Tk,
                                                 int nwords, int w, int lane) {
  for (int wd = w; wd < nwords; wd += 4) {
    const int kk = wd * 64 + lane;
    const bool msk = kk >= Tk || (key_pad && key_pad[(int64_t)b * Tk + kk]);
    const unsigned long long word = __ballot(msk);
    if (lane == 0) maskw[wd] = word;
  }
}

// Compacted list of the tiles [t_lo, t_hi) a workgroup really has to visit: those whose mask word is not all ones (`maskw` null:
// all of them).  Skipping a padded tile inside the loop still paid its LDS-DMA and its barrier; walking the list does not load it
// at all.  Wave 0 writes list[0 .. n) and the count to list[-1]; never empty (an all-padded range keeps its first tile, which the
// kernels' own masks then turn into zeros).  The caller puts a workgroup barrier before (masks built) and after.
__device__ __forceinline__ void build_tile_list(int* list, const unsigned long long* maskw, int t_lo, int t_hi, int w, int lane) {
  if (w != 0) return;
  int n = 0;
  for (int t0 = t_lo; t0 < t_hi; t0 += 64) {
    const int tt = t0 + lane;
    const bool live = tt < t_hi && (!maskw || maskw[tt] != ~0ull);
    const unsigned long long bal = __ballot(live);
    if (live) list[n + __popcll(bal & ((1ull << lane) - 1ull))] = tt;
    n += __popcll(bal);
  }
  if (lane == 0) {
    if (n == 0) { list[0] = t_lo; n = 1; }
    list[-1] = n;
  }
}

// ------------------------------------------------------------------------------------------ XCD-aware block map
// 1-D grid of nblk * B * H workgroups (nblk = 128-row blocks of one (batch, head)).  Workgroups are dealt round-robin
// over the 8 XCDs (block L -> XCD L % 8), so with the natural order the blocks of one (batch, head) land on 8 different
// L2s and each of them streams that head's K / V (or Q / dO) from HBM: measured 4.8 GB of L2 fills for 1.1 GB of
// tensors, L2 hit rate 0.28 (profiles/r02_attn_*).  Here the blocks of one (batch, head) share an XCD: they run side by
// side on its 32 CUs and the second to eighth block find the tiles in that XCD's L2.  Speed only, never correctness.
struct AttnBlock { int xb, hd, b; };
__device__ __forceinline__ AttnBlock attn_block(int H, int B, int nblk) {
  const int L = blockIdx.x, nbh = H * B;
  int bh, xb;
  if ((nbh & 7) == 0) {
    const int xcd = L & 7, idx = L >> 3;
    bh = (idx / nblk) * 8 + xcd;
    xb = idx % nblk;
  } else {
    xb = L % nblk;
    bh = L / nblk;
  }
  return {xb, bh % H, bh / H};
}

// ------------------------------------------------------------------------------------------ dual-use image
// ONE LDS image of a [rows][64] e16 tile that serves both the row reads (ds_read_b128: the 32x32x16 row operand) and
// the transposed reads (ds_read_b64_tr_b16), conflict-free for both (tools/lds_swizzle_check.py): chunk c of row r sits
// at chunk  c ^ f(r),  f(r) = ((r>>1)&1) << 2  |  ((r>>2)&3) ^ 3*((r>>4)&1).
// Bit 2 separates rows r, r+2 of a transposed 4-row block into the two 64-byte halves; bits 1..0 spread the eight
// same-parity rows of a ds_read_b128 lane group over eight chunk positions.  Backward kernels that used a row image AND
// a transposed image of the same tile (K in the dQ kernel, Q and dO in the dK/dV kernel) stage it once: half the LDS
// and half the LDS-DMA pieces.
__device__ __forceinline__ int dual_f(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 3) ^ (3 * ((row >> 4) & 1))); }
__device__ __forceinline__ void dma_piece_dual(unsigned char* img, const e16* base, int ld, int row0, int nrows, int pi, int lane) {
  const int r = 8 * pi + (lane >> 3), slot = lane & 7;
  const int chunk = slot ^ dual_f(r);
  int gr = row0 + r;
  gr = gr < nrows ? gr : nrows - 1;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (int64_t)gr * ld + chunk * 8),
                                   (__attribute__((address_space(3))) void*)(img + pi * 1024), 16, 0, 0);
}
// row fragment k-slice s of rows R0 .. R0+31 (R0 a multiple of 32): f(R0 + r) = f(r)
__device__ __forceinline__ e16x8 frag_row_dual(const unsigned char* img, int R0, int s, int lane) {
  const int r = lane & 31;
  return *(const e16x8*)(img + (R0 + r) * 128 + (((2 * s + (lane >> 5)) ^ dual_f(r)) << 4));
}
// transposed reads: lane address of the 8-row block at row 0, 32-column half 0; the block at row R (multiple of 8), half
// db is  R*128 + (T0 ^ (delta(R) << 4) ^ (db << 6))  with delta(R) = ((R>>2)&3) ^ 3*((R>>4)&1)  = 0, 2, 3, 1 for R = 0, 8, 16, 24
__device__ __forceinline__ unsigned tr_dual_t0(int lane) {
  const int g = lane >> 4, qq = (lane >> 2) & 3, p = lane & 3;
  const int rl = 4 * (g >> 1) + qq;
  const int c = 2 * (g & 1) + (p >> 1);
  const int fl = (((rl >> 1) & 1) << 2) | ((rl >> 2) & 3);
  return (unsigned)(rl * 128 + ((c ^ fl) << 4) + ((p & 1) << 3));
}
template <int OFF> __device__ __forceinline__ s16x4 tr_rd(unsigned addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// the A-operand quad (both 32-column halves) of the 16-row slice SL (0 / 1) of a 32-row dual-use image at byte offset IMG
// from the address registers' base: xa[d] = base + (T0 ^ (d << 4)), xb[d] = base + (T0 ^ (d << 4) ^ 64), d = 0..3
template <int IMG, int SL>
__device__ __forceinline__ TrQuad tr_quad_dual(const unsigned (&xa)[4], const unsigned (&xb)[4]) {
  TrQuad q;
  constexpr int DLO = SL == 0 ? 0 : 3, DHI = SL == 0 ? 2 : 1;      // delta of rows 16 SL and 16 SL + 8
  q.lo0 = tr_rd<IMG + (16 * SL) * 128>(xa[DLO]);
  q.hi0 = tr_rd<IMG + (16 * SL + 8) * 128>(xa[DHI]);
  q.lo1 = tr_rd<IMG + (16 * SL) * 128>(xb[DLO]);
  q.hi1 = tr_rd<IMG + (16 * SL + 8) * 128>(xb[DHI]);
  return q;
}

// dma_piece / dma_piece_dual with the address as (wave-uniform base) + (32-bit lane offset) (round 6): the base stays in scalar registers and
// the load takes its saddr form.  With 64-bit lane addresses the compiler hoists base + chunk per piece out of the tile loop (two registers
// per piece) and, where registers are short, spills them -- and ANY scratch reload inside a ring loop comes with `s_waitcnt vmcnt(0)`, i.e.
// waits for every piece in flight (tools/isa_ring_drain_check.py lists them).  `base` must be wave-uniform for the compiler to see (a row
// offset read from memory goes through readfirstlane first); the offset must stay below 4 GB (rows of ONE sample).
template <int KIND>      // 0 row image, 1 transposed-read image, 2 dual-use image, 3 the 16 x 16 x 32 kernels' transposed-read image (dma_piece_tr16)
__device__ __forceinline__ void dma_piece_s(unsigned char* img, const e16* base, int ld, int row0, int nrows, int pi, int lane) {
  const int r = 8 * pi + (lane >> 3), slot = lane & 7;
  const int chunk = KIND == 3 ? (slot ^ (((r >> 1) & 3) << 1)) : KIND == 2 ? (slot ^ dual_f(r)) : KIND == 1 ? (slot ^ (((r >> 1) & 1) << 2)) : (slot ^ ((r >> 1) & 7));
  int gr = row0 + r;
  gr = gr < nrows ? gr : nrows - 1;   // clamped rows are masked out by the caller
  const uint32_t off = ((uint32_t)gr * (uint32_t)ld + (uint32_t)chunk * 8u) * 2u;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)base + off),
                                   (__attribute__((address_space(3))) void*)(img + pi * 1024), 16, 0, 0);
}
// ------------------------------------------------------------------------------------------ keep-bit tensor
// DROP template values of the MFMA attention kernels: 0 no dropout, 1 hash per score pair, 2 keep-bit tensor (forward:
// hash + emit the lane masks; backward: read them).
enum { DROP_NONE = 0, DROP_HASH = 1, DROP_BITS = 2, DROP_READ = 3 };   // (READ: forward only -- the tensor was filled beforehand, afm_attn_drop_bits_fill)
__device__ __forceinline__ unsigned long long* bits_block(const AttnM& a, int bh, int qb32, int kb32) {
  // every index is wave-uniform; readfirstlane tells the compiler so (the masks then travel through SGPRs: s_load / s_store)
  unsigned long long* p = a.bits + (((int64_t)bh * a.nq32 + qb32) * a.nk32 + kb32) * 16;
  const uint64_t u = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return (unsigned long long*)(((uint64_t)hi << 32) | lo);
}
// forward: dropout of a 32x32 transposed score block (as drop_block) + the 16 lane masks to `blk` (wave-uniform pointer)
// through the scalar store path: no vector instruction, no VGPR
template <int R>
__device__ __forceinline__ void drop_emit_rows(const DropDev& dd, uint32_t base, f32x16& x, unsigned long long* blk) {
  if constexpr (R < 16) {
    const uint32_t hsh = hash_pair32(base + AFM_PAIR_OFF(R));
    const bool k0 = (hsh & 0xFFFFu) >= dd.thresh16, k1 = (hsh >> 16) >= dd.thresh16;
    const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
    x[R] = k0 ? x[R] : 0.f;
    x[R + 1] = k1 ? x[R + 1] : 0.f;
    asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m0), "s"(blk), "n"(R * 8) : "memory");
    asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m1), "s"(blk), "n"(R * 8 + 8) : "memory");
    drop_emit_rows<R + 2>(dd, base, x, blk);
  }
}
__device__ __forceinline__ void drop_block_emit(const DropDev& dd, uint32_t rowhash, int key0, int h, f32x16& x,
                                                unsigned long long* blk) {
  drop_emit_rows<0>(dd, pair_base(rowhash, key0, h), x, blk);
}
// the same 16 lane masks without a score block to apply them to (k_attn_bits_fill)
template <int R>
__device__ __forceinline__ void bits_emit_rows(const DropDev& dd, uint32_t base, unsigned long long* blk) {
  if constexpr (R < 16) {
    const uint32_t hsh = hash_pair32(base + AFM_PAIR_OFF(R));
    const unsigned long long m0 = __ballot((hsh & 0xFFFFu) >= dd.thresh16), m1 = __ballot((hsh >> 16) >= dd.thresh16);
    asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m0), "s"(blk), "n"(R * 8) : "memory");
    asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(m1), "s"(blk), "n"(R * 8 + 8) : "memory");
    bits_emit_rows<R + 2>(dd, base, blk);
  }
}
__device__ __forceinline__ void bits_flush() {   // before the kernel ends: write the scalar cache back
  asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
// backward, query-on-lane layout (dQ kernel): x[r] = keep ? x[r] * scale : 0 with the block's 16 masks as SGPR pairs.
// The loads are explicit (the compiler only picks the scalar path when it can prove nobody writes the memory): two
// s_load_dwordx16 issued at the top of the key tile (keep_masks_issue), waited for by hand right before use
// (keep_masks_wait, which ties the 32 SGPRs through "+s" so nothing consumes them earlier), ~1 us of MFMA work in between.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
struct KeepMasks { u32x16 a, b; };
__device__ __forceinline__ void keep_masks_issue(KeepMasks& m, const unsigned long long* blk) {
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(m.a), "=&s"(m.b) : "s"(blk) : "memory");
}
__device__ __forceinline__ void keep_masks_wait(KeepMasks& m) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(m.a), "+s"(m.b));
}
template <int R>
__device__ __forceinline__ void keep_apply_rows(f32x16& x, const KeepMasks& m, float scale) {
  // The multiply is ordinary C++ ON PURPOSE: x comes straight out of an MFMA chain and hipcc inserts the MFMA -> VALU wait
  // states only for instructions it emits itself; an asm v_cndmask reading the accumulator directly executed too early and
  // saw stale registers (found by bit-comparing against the re-hash path).
  if constexpr (R < 16) {
    const u32x16& v = R < 8 ? m.a : m.b;
    const unsigned long long mk = ((unsigned long long)v[2 * (R & 7) + 1] << 32) | v[2 * (R & 7)];
    const float t = x[R] * scale;
    float y;
    asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(y) : "v"(t), "s"(mk));
    x[R] = y;
    keep_apply_rows<R + 1>(x, m, scale);
  }
}
__device__ __forceinline__ void drop_apply_masks(f32x16& x, KeepMasks& m, float scale) {
  keep_masks_wait(m);
  keep_apply_rows<0>(x, m, scale);
}
// Select forms for the single-pass backward kernels (round 3): the accumulator already holds  scale * dP - delta  (V / dO
// fragments pre-multiplied by the dropout scale, -delta as the chain's initial value), so dropout is ONE select per score,
// x = keep ? x : alt.  __builtin_amdgcn_inverse_ballot_w64 turns the 64-bit SGPR mask into the select's condition without an
// instruction, and the v_cndmask is compiler-emitted (it gets the MFMA -> VALU wait states an asm statement would not).
template <int R>
__device__ __forceinline__ void keep_select_rows(f32x16& x, const KeepMasks& m, float alt) {
  if constexpr (R < 16) {
    const u32x16& v = R < 8 ? m.a : m.b;
    const unsigned long long mk = ((unsigned long long)v[2 * (R & 7) + 1] << 32) | v[2 * (R & 7)];
    x[R] = __builtin_amdgcn_inverse_ballot_w64(mk) ? x[R] : alt;
    keep_select_rows<R + 1>(x, m, alt);
  }
}
__device__ __forceinline__ void drop_select_masks(f32x16& x, KeepMasks& m, float alt) {
  keep_masks_wait(m);
  keep_select_rows<0>(x, m, alt);
}
__device__ __forceinline__ void drop_block_select(const DropDev& dd, uint32_t rowhash, int key0, int h, f32x16& x, float alt) {
  const uint32_t base = pair_base(rowhash, key0, h);
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const uint32_t hsh = hash_pair32(base + AFM_PAIR_OFF(r));
    x[r] = (hsh & 0xFFFFu) >= dd.thresh16 ? x[r] : alt;
    x[r + 1] = (hsh >> 16) >= dd.thresh16 ? x[r + 1] : alt;
  }
}
// backward, key-on-lane layout (dK/dV kernel): dword of the block that holds this lane's key (bit q = keep of the tile's query q)
__device__ __forceinline__ int bits_word_of_key(int j) { return 2 * ((j & 3) + 4 * (j >> 3)) + ((j >> 2) & 1); }
