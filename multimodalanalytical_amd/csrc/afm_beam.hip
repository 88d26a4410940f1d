// Beam-search bookkeeping on the device (reference modeling/wrapper.py:443-451 -> transformers
// GenerationMixin beam search with num_beams = num_return_sequences = k, length_penalty 1, early_stopping False,
// forced EOS at max_length).  One workgroup per sample and step: log-softmax of the k rows, top 2k of the k*V
// accumulated scores in descending order, then HF's sequential rules over those candidates (EOS inside the top k
// closes a hypothesis scored sum_logprobs / generated_len, the first k non-EOS candidates continue, the sample is done
// when it holds k hypotheses and its worst kept score beats the step's best score / generated_len), the new running
// sequences and the source row of each (for the KV-cache reorder).  The host reads back ONE int (open samples) when it
// wants to stop early.  Tiny problem sizes: latency-bound by design, no HBM traffic to speak of.
#include "afm_common.h"
#include <math.h>

struct BeamArgs {
  int B, k, V, ldl, cur_len, max_length, eos, pad, stop_rule, Lmax;
  const float* logits;
  const int64_t* seq_in;
  int64_t* seq_out;
  float* beam_scores;
  int32_t* beam_idx;
  int64_t* hyp_seq;
  float* hyp_score;
  int32_t* hyp_len;
  int32_t* hyp_count;
  int32_t* done;
  int32_t* n_open;
};

#define BEAM_MAXK 64

__device__ __forceinline__ bool cand_before(float va, int ia, float vb, int ib) {   // (value desc, index asc) order
  return va > vb || (va == vb && ia < ib);
}

__global__ __launch_bounds__(256) void k_beam_step(BeamArgs a) {
  __shared__ float lse[BEAM_MAXK];
  __shared__ float red_v[4];
  __shared__ int red_i[4];
  __shared__ float cs[2 * BEAM_MAXK];     // sorted candidate scores
  __shared__ int ci[2 * BEAM_MAXK];       // flat index beam * V + token
  __shared__ int nxt_src[BEAM_MAXK], nxt_tok[BEAM_MAXK], add_src[BEAM_MAXK], add_slot[BEAM_MAXK];
  __shared__ float nxt_sc[BEAM_MAXK];
  __shared__ int n_add;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int k = a.k, V = a.V;
  const int row0 = b * k;
  if (a.done[b]) {   // HF: a finished sample keeps emitting pad from its first row with score 0
    for (int j = 0; j < k; ++j) {
      for (int p = t; p < a.cur_len; p += 256) a.seq_out[(int64_t)(row0 + j) * a.Lmax + p] = a.seq_in[(int64_t)row0 * a.Lmax + p];
      if (t == 0) {
        a.seq_out[(int64_t)(row0 + j) * a.Lmax + a.cur_len] = a.pad;
        a.beam_scores[row0 + j] = 0.f;
        a.beam_idx[row0 + j] = row0;
      }
    }
    return;
  }
  const bool forced = a.cur_len == a.max_length - 1;    // ForcedEOSTokenLogitsProcessor
  // 1. log-sum-exp per running row (one wave per row, strided)
  for (int r = w; r < k; r += 4) {
    const float* lr = a.logits + (int64_t)(row0 + r) * a.ldl;
    float m = -INFINITY;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, lr[v]);
    m = wave_max(m);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(lr[v] - m);
    s = wave_sum(s);
    if (lane == 0) lse[r] = m + logf(s);
  }
  __syncthreads();
  auto score = [&](int idx) -> float {
    const int r = idx / V, v = idx - r * V;
    const float bs = a.beam_scores[row0 + r];
    if (forced) return v == a.eos ? bs : -INFINITY;
    return a.logits[(int64_t)(row0 + r) * a.ldl + v] - lse[r] + bs;
  };
  // 2. the 2k best candidates in descending order: 2k rounds of a block-wide argmax below the previous pick
  float pv = INFINITY;
  int pi = -1;
  const int total = k * V, want = min(2 * k, total);
  for (int round = 0; round < want; ++round) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int idx = t; idx < total; idx += 256) {
      const float v = score(idx);
      if (!(v == v)) continue;
      const bool after_prev = pi < 0 || cand_before(pv, pi, v, idx);
      if (after_prev && cand_before(v, idx, bv, bi)) { bv = v; bi = idx; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (cand_before(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { red_v[w] = bv; red_i[w] = bi; }
    __syncthreads();
    bv = red_v[0]; bi = red_i[0];
    for (int x = 1; x < 4; ++x)
      if (cand_before(red_v[x], red_i[x], bv, bi)) { bv = red_v[x]; bi = red_i[x]; }
    if (t == 0) { cs[round] = bv; ci[round] = bi; }
    pv = bv; pi = bi;
    __syncthreads();
  }
  // 3. HF's sequential rules (BeamSearchScorer.process / BeamHypotheses.add / is_done)
  if (t == 0) {
    int count = a.hyp_count[b], j = 0, nadd = 0;
    float* hs = a.hyp_score + (int64_t)b * k;
    const float glen = (float)a.cur_len;       // tokens generated once the candidate token is appended
    for (int rank = 0; rank < want && j < k; ++rank) {
      const int idx = ci[rank];
      if (idx == 0x7fffffff) break;            // fewer than 2k finite candidates
      const int beam = idx / V, tok = idx - beam * V;
      const float s = cs[rank];
      if (tok == a.eos) {
        if (rank >= k) continue;
        const float sc = s / glen;
        int slot = -1;
        if (count < k) slot = count++;
        else {
          int wi = 0;
          for (int x = 1; x < k; ++x) if (hs[x] < hs[wi]) wi = x;
          if (sc > hs[wi]) slot = wi;
        }
        if (slot >= 0) {
          hs[slot] = sc;
          a.hyp_len[(int64_t)b * k + slot] = a.cur_len;
          // a later add of this step may overwrite the same slot: keep only the last writer per slot
          for (int x = 0; x < nadd; ++x) if (add_slot[x] == slot) add_slot[x] = -1;
          add_src[nadd] = row0 + beam; add_slot[nadd] = slot; ++nadd;
        }
      } else {
        nxt_sc[j] = s; nxt_tok[j] = tok; nxt_src[j] = row0 + beam; ++j;
      }
    }
    for (; j < k; ++j) { nxt_sc[j] = -1e9f; nxt_tok[j] = a.pad; nxt_src[j] = row0; }   // cannot happen with V >= 2k + 1
    a.hyp_count[b] = count;
    n_add = nadd;
    bool is_done = false;
    if (count >= k) {
      float worst = hs[0];
      for (int x = 1; x < k; ++x) worst = fminf(worst, hs[x]);
      const float best = a.stop_rule == 0 ? cs[0] : nxt_sc[0];       // 4.48.3: best of all 2k candidates; 5.x: best running beam
      is_done = worst >= best / glen;
    }
    if (is_done) a.done[b] = 1; else atomicAdd(a.n_open, 1);
  }
  __syncthreads();
  // 4. copies: closed hypotheses (the prefix WITHOUT the eos), new running rows
  for (int x = 0; x < n_add; ++x) {
    if (add_slot[x] < 0) continue;
    int64_t* dst = a.hyp_seq + ((int64_t)b * k + add_slot[x]) * a.Lmax;
    const int64_t* src = a.seq_in + (int64_t)add_src[x] * a.Lmax;
    for (int p = t; p < a.cur_len; p += 256) dst[p] = src[p];
  }
  for (int j = 0; j < k; ++j) {
    int64_t* dst = a.seq_out + (int64_t)(row0 + j) * a.Lmax;
    const int64_t* src = a.seq_in + (int64_t)nxt_src[j] * a.Lmax;
    for (int p = t; p < a.cur_len; p += 256) dst[p] = src[p];
    if (t == 0) {
      dst[a.cur_len] = nxt_tok[j];
      a.beam_scores[row0 + j] = nxt_sc[j];
      a.beam_idx[row0 + j] = nxt_src[j];
    }
  }
}

// Close the search (BeamSearchScorer.finalize): the open beams of unfinished samples become hypotheses, the k best per
// sample are written best first as  tokens, eos (if shorter than max_length), pad ...
__global__ __launch_bounds__(64) void k_beam_finalize(BeamArgs a, int64_t* out, float* out_scores, int32_t* out_len) {
  const int b = blockIdx.x, lane = threadIdx.x, k = a.k;
  __shared__ int order[BEAM_MAXK];
  float* hs = a.hyp_score + (int64_t)b * k;
  int32_t* hl = a.hyp_len + (int64_t)b * k;
  if (lane == 0) {
    int count = a.hyp_count[b];
    if (!a.done[b]) {
      const float glen = (float)(a.cur_len - 1);
      for (int j = 0; j < k; ++j) {
        const float sc = a.beam_scores[b * k + j] / glen;
        int slot = -1;
        if (count < k) slot = count++;
        else {
          int wi = 0;
          for (int x = 1; x < k; ++x) if (hs[x] < hs[wi]) wi = x;
          if (sc > hs[wi]) slot = wi;
        }
        if (slot >= 0) {
          hs[slot] = sc; hl[slot] = a.cur_len;
          int64_t* dst = a.hyp_seq + ((int64_t)b * k + slot) * a.Lmax;
          const int64_t* src = a.seq_in + (int64_t)(b * k + j) * a.Lmax;
          for (int p = 0; p < a.cur_len; ++p) dst[p] = src[p];
        }
      }
      a.hyp_count[b] = count;
    }
    for (int i = 0; i < k; ++i) order[i] = i;         // descending by score (later additions first among equals, as HF's pop())
    for (int i = 1; i < k; ++i) {
      const int x = order[i];
      int j = i - 1;
      while (j >= 0 && (hs[order[j]] < hs[x] || (hs[order[j]] == hs[x] && order[j] < x))) { order[j + 1] = order[j]; --j; }
      order[j + 1] = x;
    }
  }
  __syncthreads();
  for (int i = 0; i < k; ++i) {
    const int slot = order[i];
    const int len = hl[slot];
    const int64_t* src = a.hyp_seq + ((int64_t)b * k + slot) * a.Lmax;
    int64_t* dst = out + (int64_t)(b * k + i) * a.max_length;
    for (int p = lane; p < a.max_length; p += 64) dst[p] = p < len ? src[p] : (p == len ? (int64_t)a.eos : (int64_t)a.pad);
    if (lane == 0) { out_scores[b * k + i] = hs[slot]; out_len[b * k + i] = min(len + 1, a.max_length); }
  }
}

static int beam_args(const afm_beam_desc* d, BeamArgs& a) {
  if (!d || d->B <= 0 || d->k <= 0 || d->k > BEAM_MAXK || d->V <= 1 || d->ldl < d->V || d->cur_len < 1 ||
      d->max_length < d->cur_len || d->Lmax < d->max_length || !d->seq_in || !d->beam_scores || !d->hyp_seq ||
      !d->hyp_score || !d->hyp_len || !d->hyp_count || !d->done)
    return AFM_ERR_ARG;
  a.B = d->B; a.k = d->k; a.V = d->V; a.ldl = d->ldl; a.cur_len = d->cur_len; a.max_length = d->max_length;
  a.eos = d->eos; a.pad = d->pad; a.stop_rule = d->stop_rule; a.Lmax = d->Lmax;
  a.logits = d->logits; a.seq_in = d->seq_in; a.seq_out = d->seq_out; a.beam_scores = d->beam_scores; a.beam_idx = d->beam_idx;
  a.hyp_seq = d->hyp_seq; a.hyp_score = d->hyp_score; a.hyp_len = d->hyp_len; a.hyp_count = d->hyp_count; a.done = d->done;
  a.n_open = d->n_open;
  return AFM_OK;
}

extern "C" int afm_beam_step(const afm_beam_desc* d, void* stream) {
  BeamArgs a;
  const int r = beam_args(d, a);
  if (r != AFM_OK) return r;
  if (!d->logits || !d->seq_out || !d->beam_idx || !d->n_open || d->cur_len >= d->max_length) return AFM_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(d->n_open, 0, sizeof(int32_t), st) != hipSuccess) return AFM_ERR_LAUNCH;
  AFM_LAUNCH(k_beam_step, dim3(d->B), dim3(256), 0, st, a);
  return AFM_OK;
}

extern "C" int afm_beam_finalize(const afm_beam_desc* d, int64_t* out, float* out_scores, int32_t* out_len, void* stream) {
  BeamArgs a;
  const int r = beam_args(d, a);
  if (r != AFM_OK) return r;
  if (!out || !out_scores || !out_len) return AFM_ERR_ARG;
  AFM_LAUNCH(k_beam_finalize, dim3(d->B), dim3(64), 0, (hipStream_t)stream, a, out, out_scores, out_len);
  return AFM_OK;
}

// Rows [0, t_used) of every (Tmax x width) cache block: dst[r] = src[beam_idx[r]]  (KV-cache reorder between steps).
__global__ __launch_bounds__(256) void k_cache_reorder(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                       const int32_t* __restrict__ beam_idx, int64_t block_u4, int64_t used_u4) {
  const int r = blockIdx.y;
  const uint4* s = src + (int64_t)beam_idx[r] * block_u4;
  uint4* d = dst + (int64_t)r * block_u4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < used_u4; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}
extern "C" int afm_cache_reorder(const void* src, void* dst, const int32_t* beam_idx, int32_t rows, int64_t block_bytes,
                                 int64_t used_bytes, void* stream) {
  if (!src || !dst || !beam_idx || rows <= 0 || block_bytes <= 0 || used_bytes < 0 || used_bytes > block_bytes ||
      (block_bytes & 15) || (used_bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15))
    return AFM_ERR_ARG;
  if (used_bytes == 0) return AFM_OK;
  int gx = (int)((used_bytes / 16 + 255) / 256);
  if (gx > 64) gx = 64;
  AFM_LAUNCH(k_cache_reorder, dim3(gx, rows), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, beam_idx,
             block_bytes / 16, used_bytes / 16);
  return AFM_OK;
}
