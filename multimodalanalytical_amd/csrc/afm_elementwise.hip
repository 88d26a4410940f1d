// HBM-bound elementwise / gather / reduction kernels of the spectra->SMILES path (gfx950).
// Each kernel moves every byte once; rows are walked with 64-lane waves on contiguous data.
#include <algorithm>
#include "afm_common.h"

static thread_local const char* g_last_algo = "none";
extern "C" void afm_set_last_algo(const char* name) { g_last_algo = name; }
extern "C" const char* afm_last_algo(void) { return g_last_algo; }
static thread_local int g_last_hint = 0;
extern "C" void afm_note_hint(int state) { g_last_hint = state; }
extern "C" int afm_last_hint(void) { return g_last_hint; }
extern "C" int afm_abi_version(void) { return AFM_ABI_VERSION; }
extern "C" int afm_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(afm_dropout);
    case 1: return (int)sizeof(afm_gemm_desc);
    case 2: return (int)sizeof(afm_ln_shape);
    case 3: return (int)sizeof(afm_attn_shape);
    case 4: return (int)sizeof(afm_patch_desc);
    case 5: return (int)sizeof(afm_beam_desc);
    case 6: return (int)sizeof(afm_cast_item);
    default: return -1;
  }
}
extern "C" const char* afm_error_string(int code) {
  switch (code) {
    case AFM_OK: return "ok";
    case AFM_ERR_ARG: return "invalid argument";
    case AFM_ERR_UNSUPPORTED: return "unsupported shape/dtype for the requested algorithm";
    case AFM_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown error";
  }
}

static inline int grid_for(int64_t work, int block, int cap = 256 * 8) {
  int64_t g = (work + block - 1) / block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

// ---------------------------------------------------------------- embedding rows
__global__ void k_gather_rows(const int64_t* __restrict__ ids, const float* __restrict__ scale,
                              const float* __restrict__ table, float* __restrict__ out, int64_t n,
                              int d, int V) {
  // one wave per row, lanes stride the row
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const float s = scale ? scale[r] : 1.0f;
    const float* src = table + id * (int64_t)d;
    float* dst = out + r * (int64_t)d;
    for (int j = lane; j < d; j += 64) dst[j] = src[j] * s;
  }
}
extern "C" int afm_gather_rows(const int64_t* ids, const float* scale, const float* table, float* out,
                               int64_t n, int32_t d, int32_t V, void* stream) {
  if (!ids || !table || !out || n < 0 || d <= 0 || V <= 0) return AFM_ERR_ARG;
  if (n == 0) return AFM_OK;
  AFM_LAUNCH(k_gather_rows, dim3(grid_for(n * 64, 256)), dim3(256), 0, (hipStream_t)stream,
                     ids, scale, table, out, n, d, V);
  return AFM_OK;
}

__global__ void k_scatter_add_rows(const int64_t* __restrict__ ids, const float* __restrict__ scale,
                                   const float* __restrict__ dout, float* __restrict__ dtable,
                                   int64_t n, int d, int V, int64_t pad) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    int64_t id = ids[r];
    if (id == pad) continue;
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const float s = scale ? scale[r] : 1.0f;
    const float* src = dout + r * (int64_t)d;
    float* dst = dtable + id * (int64_t)d;
    for (int j = lane; j < d; j += 64) atomicAdd(dst + j, src[j] * s);
  }
}
extern "C" int afm_scatter_add_rows(const int64_t* ids, const float* scale, const float* dout,
                                    float* dtable, int64_t n, int32_t d, int32_t V,
                                    int64_t padding_idx, void* stream) {
  if (!ids || !dout || !dtable || n < 0 || d <= 0 || V <= 0) return AFM_ERR_ARG;
  if (n == 0) return AFM_OK;
  AFM_LAUNCH(k_scatter_add_rows, dim3(grid_for(n * 64, 256)), dim3(256), 0,
                     (hipStream_t)stream, ids, scale, dout, dtable, n, d, V, padding_idx);
  return AFM_OK;
}

// ---------------------------------------------------------------- embedding rows without the per-modality LayerNorm
// multimodal_norm = False (modeling/utils.py:165-168 skipped): the modality's rows go straight into their slice of the
// concatenated sequence, positional rows added: the LayerNorm kernel's layout fusion without the normalisation.
//   gather == 0:  y[out_row(r), :] = x[r, :] + (pos ? pos[out_off + r % seg_len, :] : 0)
//   gather == 1:  y[r, :] = x[out_row(r), :]                         (its backward: the slice of the stream gradient)
__global__ void k_place_rows(const float* __restrict__ x, const float* __restrict__ pos, float* __restrict__ y, int64_t rows, int d,
                             int64_t seg_len, int64_t seg_stride, int64_t off, int gather) {
  const int64_t total = rows * (int64_t)d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / d;
    const int c = (int)(i - r * d);
    const int64_t orow = seg_len ? (r / seg_len) * seg_stride + off + (r % seg_len) : r;
    if (gather) y[i] = x[orow * d + c];
    else y[orow * d + c] = x[i] + (pos ? pos[(off + (seg_len ? r % seg_len : r)) * d + c] : 0.f);
  }
}
extern "C" int afm_place_rows(const float* x, const float* pos, float* y, int64_t rows, int32_t d, int64_t seg_len,
                              int64_t out_seg_stride, int64_t out_off, int32_t gather, void* stream) {
  if (!x || !y || rows < 0 || d <= 0 || seg_len < 0) return AFM_ERR_ARG;
  if (rows == 0) return AFM_OK;
  AFM_LAUNCH(k_place_rows, dim3(grid_for(rows * d, 256)), dim3(256), 0, (hipStream_t)stream, x, pos, y, rows, d, seg_len,
             out_seg_stride, out_off, gather);
  return AFM_OK;
}

// ---------------------------------------------------------------- padded positions out of the forward pass (include/afm_hip.h, ABI 6)
// live positions per sample
#define AFM_SLOT_ROWS 32      // a packed sample's slot = its live length rounded up to this (a wave of the attention kernels owns 32 rows)
__global__ __launch_bounds__(256) void k_count_live(const uint8_t* __restrict__ key_pad, int S, int32_t* __restrict__ n_live,
                                                    uint8_t* __restrict__ clear, int64_t nclear) {
  __shared__ int part[4];
  // (packed mode: the block flags are MARKED by the samples that own rows in them -- blocks straddle samples -- so they start cleared)
  if (clear) for (int64_t i = blockIdx.x + (int64_t)gridDim.x * threadIdx.x; i < nclear; i += (int64_t)gridDim.x * 256) clear[i] = 0;
  const uint8_t* kp = key_pad + (int64_t)blockIdx.x * S;
  int cnt = 0;
  for (int s = threadIdx.x; s < S; s += 256) cnt += kp[s] == 0;
  cnt = (int)wave_sum((float)cnt);      // (<= 4096 / 4 per wave: exact in fp32)
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) n_live[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// One workgroup per sample: thread t owns the positions [t * chunk, (t + 1) * chunk); live counts -> block scan -> the stable partition.
// mode 2 (packed): the sample's slot starts at off = sum over earlier samples of ceil32(live); its padded positions fill the slot's
// last rows first, the rest go to the batch's dead tail behind all slots, in order.
__global__ __launch_bounds__(256) void k_compact_plan(const uint8_t* __restrict__ key_pad, int B, int S, int tile_rows, int mode,
                                                      int32_t* __restrict__ dest, int32_t* __restrict__ seq_off, uint8_t* __restrict__ pad_out,
                                                      uint8_t* __restrict__ live64, uint8_t* __restrict__ live_tile,
                                                      const int32_t* __restrict__ n_live) {
  __shared__ int scan[256];
  __shared__ int blk_any[64];
  __shared__ int red[2][4];
  const int b = blockIdx.x, t = threadIdx.x;
  const uint8_t* kp = key_pad + (int64_t)b * S;
  // rows in front of this sample's slot and rows in use over the batch
  int64_t off = (int64_t)b * S, total = (int64_t)B * S;
  if (mode == 2) {
    int before = 0, all = 0;
    for (int i = t; i < B; i += 256) {
      const int slot = (n_live[i] + AFM_SLOT_ROWS - 1) & ~(AFM_SLOT_ROWS - 1);
      all += slot;
      if (i < b) before += slot;
    }
    before = (int)wave_sum((float)before); all = (int)wave_sum((float)all);      // (sums below 2^24: exact)
    if ((t & 63) == 0) { red[0][t >> 6] = before; red[1][t >> 6] = all; }
    __syncthreads();
    off = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    total = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
  const int chunk = (S + 255) / 256;
  const int s0 = min(S, t * chunk), s1 = min(S, s0 + chunk);
  if (t < 64) blk_any[t] = 0;
  int cnt = 0;
  for (int s = s0; s < s1; ++s) cnt += kp[s] == 0;
  scan[t] = cnt;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {      // Hillis-Steele inclusive scan (256 entries)
    const int v = t >= o ? scan[t - o] : 0;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  const int L = scan[255];
  const int slot = mode == 2 ? (L + AFM_SLOT_ROWS - 1) & ~(AFM_SLOT_ROWS - 1) : S;
  const int64_t tail0 = total + ((int64_t)b * S - off);      // packed: where this sample's share of the dead tail starts
  int before = scan[t] - cnt;                    // live positions in front of this thread's range
  for (int s = s0; s < s1; ++s) {
    const bool live = kp[s] == 0;
    if (dest) {
      int64_t row;
      if (mode == 0) row = off + s;
      else if (live) row = off + before;
      else {
        const int r = s - before;                // rank among the sample's padded positions
        row = r < slot - L ? off + L + r : tail0 + (r - (slot - L));
      }
      dest[(int64_t)b * S + s] = (int32_t)row;
    }
    if (mode == 0 && live) blk_any[s >> 6] = 1;   // (benign race: every writer stores 1)
    before += live;
  }
  if (seq_off) {
    if (t == 0) seq_off[b] = (int32_t)off;
    if (t == 1 && b == B - 1) seq_off[B] = (int32_t)(mode == 2 ? total : (int64_t)B * S);
  }
  __syncthreads();
  if (pad_out) for (int p = t; p < S; p += 256) pad_out[(int64_t)b * S + p] = mode ? (uint8_t)(p >= L) : kp[p];
  const int nb = S >> 6, per = tile_rows >> 6;
  if (mode != 2) {
    for (int i = t; i < nb; i += 256) {
      bool l64, lt;
      if (mode == 1) { l64 = 64 * i < L; lt = (i / per) * tile_rows < L; }
      else {
        l64 = blk_any[i] != 0; lt = false;
        for (int j = (i / per) * per; j < (i / per + 1) * per; ++j) lt = lt || blk_any[j] != 0;
      }
      if (live64) live64[(int64_t)b * nb + i] = l64;
      if (live_tile) live_tile[(int64_t)b * nb + i] = lt;
    }
  } else {
    // packed: blocks are counted over the whole matrix.  This sample flags the blocks of its own slot; every sample takes its share
    // (index mod B) of the dead tail's blocks and of the widened flags.
    const int64_t nblk = (int64_t)B * nb;
    if (live64 && L > 0)      // (cleared by k_count_live) every 64-row block that holds one of this sample's live rows [off, off + L)
      for (int64_t i = (off >> 6) + t; i <= ((off + L - 1) >> 6); i += 256) live64[i] = 1;
    if (live_tile)
      for (int64_t i = b + (int64_t)B * t; i < nblk; i += (int64_t)B * 256) live_tile[i] = (i / per) * tile_rows < total;
  }
}
extern "C" int afm_compact_plan(const uint8_t* key_pad, int32_t B, int32_t S, int32_t tile_rows, int32_t compact, int32_t* dest, int32_t* seq_off,
                                uint8_t* pad_out, uint8_t* live64, uint8_t* live_tile, int32_t* n_live, void* stream) {
  if (!key_pad || !n_live || B <= 0 || S <= 0 || S > 4096 || tile_rows <= 0 || (tile_rows & 63) || (S % tile_rows) || compact < 0 || compact > 2) return AFM_ERR_ARG;
  if (compact && pad_out == key_pad) return AFM_ERR_ARG;      // the kernel re-reads the mask after writing the new one
  if ((int64_t)B * S > (1ll << 24)) return AFM_ERR_ARG;      // (row counts are summed in fp32 lanes: exact below 2^24)
  if (compact == 2 && (S & 127)) return AFM_ERR_ARG;          // (the attention kernels' 128-row blocks inside S-row budgets)
  AFM_LAUNCH(k_count_live, dim3(B), dim3(256), 0, (hipStream_t)stream, key_pad, S, n_live, compact == 2 ? live64 : (uint8_t*)nullptr,
             (int64_t)B * (S >> 6));
  AFM_LAUNCH(k_compact_plan, dim3(B), dim3(256), 0, (hipStream_t)stream, key_pad, B, S, tile_rows, compact, dest, seq_off, pad_out, live64, live_tile,
             (const int32_t*)n_live);
  return AFM_OK;
}

// one wave per row, 16 bytes per lane
__global__ __launch_bounds__(256) void k_permute_rows(const float* __restrict__ x, float* __restrict__ y, const int32_t* __restrict__ map,
                                                      int64_t rows, int S, int d, int gather) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t m = map[r];
    const float* src = x + (gather ? m : r) * (int64_t)d;
    float* dst = y + (gather ? r : m) * (int64_t)d;
    if ((d & 3) == 0) for (int c = lane * 4; c < d; c += 256) *(float4*)(dst + c) = *(const float4*)(src + c);
    else for (int c = lane; c < d; c += 64) dst[c] = src[c];
  }
}
extern "C" int afm_permute_rows(const float* x, float* y, const int32_t* map, int32_t B, int32_t S, int32_t d, int32_t gather, void* stream) {
  if (!x || !y || !map || x == y || B <= 0 || S <= 0 || d <= 0) return AFM_ERR_ARG;
  const int64_t rows = (int64_t)B * S;
  AFM_LAUNCH(k_permute_rows, dim3((int)std::min<int64_t>((rows + 3) / 4, 2048)), dim3(256), 0, (hipStream_t)stream, x, y, map, rows, S, d, gather);
  return AFM_OK;
}

// ---------------------------------------------------------------- GLU / GELU
template <typename T, bool RELU>
__global__ void k_glu_fwd(const T* __restrict__ u, const T* __restrict__ v, T* __restrict__ g,
                          int64_t rows, int f, int ldu, int ldv, int ldg, DropDev dd) {
  const int64_t total = rows * (int64_t)f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / f;
    const int c = (int)(i - r * f);
    const float uu = ld_rc(u, r, c, ldu);
    float x = RELU ? fmaxf(uu, 0.f) : afm_gelu(uu);
    if (v) x *= ld_rc(v, r, c, ldv);
    st_rc(g, r, c, ldg, afm_drop(dd, (uint64_t)i, x));
  }
}
extern "C" int afm_glu_fwd(const void* u, const void* v, void* g, int64_t rows, int32_t f,
                           int32_t ldu, int32_t ldv, int32_t ldg, int32_t dtype, int32_t act,
                           const afm_dropout* drop, void* stream) {
  if (!u || !g || rows < 0 || f <= 0 || (act != AFM_ACT_GELU && act != AFM_ACT_RELU)) return AFM_ERR_ARG;
  if (rows == 0) return AFM_OK;
  const DropDev dd = afm_make_drop(drop);
  const int grid = grid_for(rows * f, 256);
  if (act == AFM_ACT_RELU) {
    AFM_DT_SWITCH(dtype, T, AFM_LAUNCH((k_glu_fwd<T, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)u,
                                       (const T*)v, (T*)g, rows, f, ldu, ldv, ldg, dd));
  } else {
    AFM_DT_SWITCH(dtype, T, AFM_LAUNCH((k_glu_fwd<T, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)u,
                                       (const T*)v, (T*)g, rows, f, ldu, ldv, ldg, dd));
  }
  return AFM_OK;
}

template <typename T, bool RELU>
__global__ void k_glu_bwd(const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ dg,
                          T* __restrict__ du, T* __restrict__ dv, int64_t rows, int f, int ldu,
                          int ldv, int lddg, int lddu, int lddv, DropDev dd) {
  const int64_t total = rows * (int64_t)f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / f;
    const int c = (int)(i - r * f);
    const float g = afm_drop(dd, (uint64_t)i, ld_rc(dg, r, c, lddg));
    const float uu = ld_rc(u, r, c, ldu);
    float gu = RELU ? (uu > 0.f ? g : 0.f) : g * afm_gelu_grad(uu);      // (torch: relu'(0) = 0)
    if (v) {
      gu *= ld_rc(v, r, c, ldv);
      st_rc(dv, r, c, lddv, g * (RELU ? fmaxf(uu, 0.f) : afm_gelu(uu)));
    }
    st_rc(du, r, c, lddu, gu);
  }
}
extern "C" int afm_glu_bwd(const void* u, const void* v, const void* dg, void* du, void* dv,
                           int64_t rows, int32_t f, int32_t ldu, int32_t ldv, int32_t lddg,
                           int32_t lddu, int32_t lddv, int32_t dtype, int32_t act, const afm_dropout* drop,
                           void* stream) {
  if (!u || !dg || !du || rows < 0 || f <= 0 || (v && !dv) || (act != AFM_ACT_GELU && act != AFM_ACT_RELU)) return AFM_ERR_ARG;
  if (rows == 0) return AFM_OK;
  const DropDev dd = afm_make_drop(drop);
  const int grid = grid_for(rows * f, 256);
  if (act == AFM_ACT_RELU) {
    AFM_DT_SWITCH(dtype, T, AFM_LAUNCH((k_glu_bwd<T, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)u,
                                       (const T*)v, (const T*)dg, (T*)du, (T*)dv, rows, f, ldu, ldv, lddg, lddu, lddv, dd));
  } else {
    AFM_DT_SWITCH(dtype, T, AFM_LAUNCH((k_glu_bwd<T, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)u,
                                       (const T*)v, (const T*)dg, (T*)du, (T*)dv, rows, f, ldu, ldv, lddg, lddu, lddv, dd));
  }
  return AFM_OK;
}

// ---------------------------------------------------------------- dropout + cast
template <typename T>
__global__ void k_dropout_cast(const float* __restrict__ x, T* __restrict__ y, int64_t rows, int n,
                               int ldx, int ldy, DropDev dd) {
  const int64_t total = rows * (int64_t)n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / n;
    const int c = (int)(i - r * n);
    st_rc(y, r, c, ldy, afm_drop(dd, (uint64_t)i, x[r * ldx + c]));
  }
}
extern "C" int afm_dropout_cast(const float* x, void* y, int64_t rows, int32_t n, int32_t ldx,
                                int32_t ldy, int32_t y_dtype, const afm_dropout* drop,
                                void* stream) {
  if (!x || !y || rows < 0 || n <= 0) return AFM_ERR_ARG;
  if (rows == 0) return AFM_OK;
  const DropDev dd = afm_make_drop(drop);
  const int grid = grid_for(rows * n, 256);
  AFM_DT_SWITCH(y_dtype, T, AFM_LAUNCH(k_dropout_cast<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (T*)y,
                                       rows, n, ldx, ldy, dd));
  return AFM_OK;
}

// ---------------------------------------------------------------- dtype conversion
template <typename TS, typename TD>
__global__ void k_convert(const TS* __restrict__ x, TD* __restrict__ y, int64_t rows, int n, int lds, int ldd) {
  const int64_t total = rows * (int64_t)n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / n;
    const int c = (int)(i - r * n);
    st_rc(y, r, c, ldd, ld_rc(x, r, c, lds));
  }
}
extern "C" int afm_convert(const void* src, int32_t src_dtype, int32_t lds, void* dst, int32_t dst_dtype,
                           int32_t ldd, int64_t rows, int32_t n, void* stream) {
  if (!src || !dst || rows < 0 || n <= 0) return AFM_ERR_ARG;
  if (rows == 0) return AFM_OK;
  const int grid = grid_for(rows * n, 256);
  AFM_DT_SWITCH(src_dtype, TS, AFM_DT_SWITCH(dst_dtype, TD,
      AFM_LAUNCH((k_convert<TS, TD>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const TS*)src, (TD*)dst,
                 rows, n, lds, ldd)));
  return AFM_OK;
}

__global__ void k_relu_bwd(const float* __restrict__ dy, const float* __restrict__ act, float* __restrict__ dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = act[i] > 0.f ? dy[i] : 0.f;
}
extern "C" int afm_relu_bwd(const float* dy, const float* act, float* dx, int64_t n, void* stream) {
  if (!dy || !act || !dx || n < 0) return AFM_ERR_ARG;
  if (n == 0) return AFM_OK;
  AFM_LAUNCH(k_relu_bwd, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, act, dx, n);
  return AFM_OK;
}

// ---------------------------------------------------------------- column sums (bias gradients)
// grid.x tiles the columns (64 per block), grid.y splits the rows; each block reduces its row
// range for 64 columns (4 waves x 64 lanes: lane = column, wave = row phase), then one atomic
// per column.  Lanes of a wave read 64 consecutive elements of a row: coalesced.
template <typename T>
__global__ void k_colsum(const T* __restrict__ x, float* __restrict__ out, int64_t rows, int n,
                         int ld) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int64_t chunk = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * chunk;
  const int64_t r1 = r0 + chunk < rows ? r0 + chunk : rows;
  float acc = 0.f;
  if (c < n)
    for (int64_t r = r0 + w; r < r1; r += 4) acc += ld_rc(x, r, c, ld);
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0 && c < n) atomicAdd(out + c, part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
}
extern "C" int afm_colsum(const void* x, float* out, int64_t rows, int32_t n, int32_t ld,
                          int32_t dtype, int32_t accumulate, void* stream) {
  if (!x || !out || rows < 0 || n <= 0) return AFM_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate)
    if (hipMemsetAsync(out, 0, sizeof(float) * n, st) != hipSuccess) return AFM_ERR_LAUNCH;
  if (rows == 0) return AFM_OK;
  const int gx = (n + 63) / 64;
  int gy = (int)((rows + 255) / 256);
  const int cap = (2048 + gx - 1) / gx;
  if (gy > cap) gy = cap;
  if (gy < 1) gy = 1;
  AFM_DT_SWITCH(dtype, T, AFM_LAUNCH(k_colsum<T>, dim3(gx, gy), dim3(256), 0, st, (const T*)x, out, rows, n, ld));
  return AFM_OK;
}

__global__ void k_add_inplace(float* __restrict__ y, const float* __restrict__ x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    y[i] += x[i];
}
extern "C" int afm_add_inplace(float* y, const float* x, int64_t n, void* stream) {
  if (!y || !x || n < 0) return AFM_ERR_ARG;
  if (n == 0) return AFM_OK;
  AFM_LAUNCH(k_add_inplace, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, y, x, n);
  return AFM_OK;
}

__global__ void k_batch_sum(const float* __restrict__ x, float* __restrict__ out, int B, int64_t S,
                            int d, int accumulate) {
  const int64_t total = S * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    float acc = accumulate ? out[i] : 0.f;
    for (int b = 0; b < B; ++b) acc += x[(int64_t)b * total + i];
    out[i] = acc;
  }
}
extern "C" int afm_batch_sum(const float* x, float* out, int32_t B, int64_t S, int32_t d,
                             int32_t accumulate, void* stream) {
  if (!x || !out || B <= 0 || S < 0 || d <= 0) return AFM_ERR_ARG;
  if (S == 0) return AFM_OK;
  AFM_LAUNCH(k_batch_sum, dim3(grid_for(S * d, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     out, B, S, d, accumulate);
  return AFM_OK;
}

// ---------------------------------------------------------------- fp32 -> bf16 (+ transpose)
// 64x64 tiles through LDS so both the row-major copy and the transposed copy are written with
// consecutive lanes on consecutive addresses.
__global__ void k_cast_bf16(const float* __restrict__ src, bf16* __restrict__ dst,
                            bf16* __restrict__ dst_t, int rows, int cols) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 row phases
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = src[(int64_t)r * cols + c];
      if (dst) dst[(int64_t)r * cols + c] = (bf16)v;
    }
    tile[i][tx] = v;
  }
  if (!dst_t) return;
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;  // transposed: row index of dst_t is the column
    if (r < rows && c < cols) dst_t[(int64_t)c * rows + r] = (bf16)tile[tx][i];
  }
}
extern "C" int afm_cast_bf16(const float* src, void* dst, void* dst_t, int32_t rows, int32_t cols,
                             void* stream) {
  if (!src || (!dst && !dst_t) || rows <= 0 || cols <= 0) return AFM_ERR_ARG;
  AFM_LAUNCH(k_cast_bf16, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0,
                     (hipStream_t)stream, src, (bf16*)dst, (bf16*)dst_t, rows, cols);
  return AFM_OK;
}

// fp32 -> split bf16 pair (+ transpose): dst rows are [hi(cols) | lo(cols)], dst_t rows [hi(rows) | lo(rows)]
__global__ void k_cast_x2(const float* __restrict__ src, bf16* __restrict__ dst,
                          bf16* __restrict__ dst_t, int rows, int cols) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = src[(int64_t)r * cols + c];
      if (dst) {
        bf16 hi, lo;
        afm_split(v, hi, lo);
        dst[(int64_t)r * 2 * cols + c] = hi;
        dst[(int64_t)r * 2 * cols + cols + c] = lo;
      }
    }
    tile[i][tx] = v;
  }
  if (!dst_t) return;
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) {
      bf16 hi, lo;
      afm_split(tile[tx][i], hi, lo);
      dst_t[(int64_t)c * 2 * rows + r] = hi;
      dst_t[(int64_t)c * 2 * rows + rows + r] = lo;
    }
  }
}
extern "C" int afm_cast_x2(const float* src, void* dst, void* dst_t, int32_t rows, int32_t cols, void* stream) {
  if (!src || (!dst && !dst_t) || rows <= 0 || cols <= 0) return AFM_ERR_ARG;
  AFM_LAUNCH(k_cast_x2, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0,
                     (hipStream_t)stream, src, (bf16*)dst, (bf16*)dst_t, rows, cols);
  return AFM_OK;
}

// ---------------------------------------------------------------- weight shadows with the gated-FFN interleave
__device__ __forceinline__ int glu_interleave(int r, int f) {   // row r of [W1 ; Wg] -> row of the interleaved matrix
  const int j = r < f ? r : r - f;
  return ((j >> 2) << 3) + (j & 3) + (r < f ? 0 : 4);
}
// MODE 0: bf16, 1: split bf16 pair, 2: fp16 (one plane)
template <int MODE>
__global__ void k_cast_weights(const float* __restrict__ src, bf16* __restrict__ dst, bf16* __restrict__ dst_t, int rows,
                               int cols, int glu_f) {
  constexpr bool X2T = MODE == 1;
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int ldd = X2T ? 2 * cols : cols, ldt = X2T ? 2 * rows : rows;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = src[(int64_t)r * cols + c];
      if (dst) {
        const int rd = glu_f ? glu_interleave(r, glu_f) : r;
        if (MODE == 2) ((f16*)dst)[(int64_t)rd * ldd + c] = (f16)v;
        else {
          bf16 hi, lo;
          afm_split(v, hi, lo);
          dst[(int64_t)rd * ldd + c] = hi;
          if (X2T) dst[(int64_t)rd * ldd + cols + c] = lo;
        }
      }
    }
    tile[i][tx] = v;
  }
  if (!dst_t) return;
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) {
      const int rd = glu_f ? glu_interleave(r, glu_f) : r;
      if (MODE == 2) ((f16*)dst_t)[(int64_t)c * ldt + rd] = (f16)tile[tx][i];
      else {
        bf16 hi, lo;
        afm_split(tile[tx][i], hi, lo);
        dst_t[(int64_t)c * ldt + rd] = hi;
        if (X2T) dst_t[(int64_t)c * ldt + rows + rd] = lo;
      }
    }
  }
}
// Every weight shadow of an optimiser step in ONE launch (afm_cast_weights_batch): block b finds its item by the running tile count.
template <int MODE>
__global__ void k_cast_weights_batch(const afm_cast_item* __restrict__ items, int n) {
  int lo = 0, hi = n - 1;                     // last item with tile0 <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const afm_cast_item it = items[lo];
  const int rows = it.rows, cols = it.cols, glu_f = it.glu_rows;
  const int tpr = (cols + 63) / 64, lt = (int)blockIdx.x - it.tile0;
  const float* __restrict__ src = it.src;
  bf16* __restrict__ dst = (bf16*)it.dst;
  bf16* __restrict__ dst_t = (bf16*)it.dst_t;
  constexpr bool X2T = MODE == 1;
  __shared__ float tile[64][65];
  const int r0 = (lt / tpr) * 64, c0 = (lt % tpr) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int ldd = X2T ? 2 * cols : cols, ldt = X2T ? 2 * rows : rows;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = src[(int64_t)r * cols + c];
      if (dst) {
        const int rd = glu_f ? glu_interleave(r, glu_f) : r;
        if (MODE == 2) ((f16*)dst)[(int64_t)rd * ldd + c] = (f16)v;
        else {
          bf16 hi_, lo_;
          afm_split(v, hi_, lo_);
          dst[(int64_t)rd * ldd + c] = hi_;
          if (X2T) dst[(int64_t)rd * ldd + cols + c] = lo_;
        }
      }
    }
    tile[i][tx] = v;
  }
  if (!dst_t) return;
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) {
      const int rd = glu_f ? glu_interleave(r, glu_f) : r;
      if (MODE == 2) ((f16*)dst_t)[(int64_t)c * ldt + rd] = (f16)tile[tx][i];
      else {
        bf16 hi_, lo_;
        afm_split(tile[tx][i], hi_, lo_);
        dst_t[(int64_t)c * ldt + rd] = hi_;
        if (X2T) dst_t[(int64_t)c * ldt + rows + rd] = lo_;
      }
    }
  }
}
extern "C" int afm_cast_weights_batch(const afm_cast_item* items, int32_t n, int32_t tiles, int32_t dtype, void* stream) {
  if (!items || n <= 0 || tiles <= 0) return AFM_ERR_ARG;
  if (dtype == AFM_BF16) AFM_LAUNCH(k_cast_weights_batch<0>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, items, n);
  else if (dtype == AFM_BF16X2) AFM_LAUNCH(k_cast_weights_batch<1>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, items, n);
  else if (dtype == AFM_F16) AFM_LAUNCH(k_cast_weights_batch<2>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, items, n);
  else return AFM_ERR_ARG;
  return AFM_OK;
}
extern "C" int afm_cast_weights(const float* src, void* dst, void* dst_t, int32_t rows, int32_t cols, int32_t dtype,
                                int32_t glu_rows, void* stream) {
  if (!src || (!dst && !dst_t) || rows <= 0 || cols <= 0 || glu_rows < 0) return AFM_ERR_ARG;
  if (glu_rows && (rows != 2 * glu_rows || (glu_rows & 3))) return AFM_ERR_ARG;
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (dtype == AFM_BF16) AFM_LAUNCH(k_cast_weights<0>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, (bf16*)dst_t, rows, cols, glu_rows);
  else if (dtype == AFM_BF16X2) AFM_LAUNCH(k_cast_weights<1>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, (bf16*)dst_t, rows, cols, glu_rows);
  else if (dtype == AFM_F16) AFM_LAUNCH(k_cast_weights<2>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, (bf16*)dst_t, rows, cols, glu_rows);
  else return AFM_ERR_ARG;
  return AFM_OK;
}
