/*
 * afm_hip.h -- C ABI of libafm_hip.so, the MI355X (gfx950 / CDNA4) kernel library behind
 * the spectra->SMILES training path of `analytical_fm` (rxn4chemistry/MultimodalAnalytical).
 *
 * The reference has no native boundary: its hot path is Python calling torch.nn modules.
 * Each entry point below therefore replaces one torch operator family at the place the
 * reference invokes it; the reference call site is cited on every declaration
 * (paths relative to the reference's src/analytical_fm/, "torch:" = the torch wheel).
 * INTEGRATION.md shows the ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in _host;
 *   - caller allocates everything; no entry point allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and the call
 *     returns immediately (safe under hipGraph stream capture);
 *   - return value: AFM_OK (0) or a negative AFM_ERR_* code; nothing is launched on error;
 *   - matrices are row-major; `ld*` are row strides in ELEMENTS;
 *   - dtype codes: AFM_F32 = 0 (float), AFM_BF16 = 1 (bfloat16, round-to-nearest-even),
 *     AFM_BF16X2 = 2: a SPLIT bf16 pair per element, value = hi + lo with hi = bf16(v) and
 *     lo = bf16(v - hi) (16 significant bits).  The two bf16 planes of a row sit side by side: a tensor
 *     with row stride ld (elements) keeps hi(r, c) at base[r*ld + c] and lo(r, c) at base[r*ld + ld/2 + c],
 *     so a contiguous (rows x n) tensor occupies rows x 2n bf16 and any column slice keeps the parent's
 *     ld.  The hi plane alone is a valid AFM_BF16 tensor of the same ld.  GEMM / attention on split
 *     operands run three bf16 MFMAs per product (hi*hi + hi*lo + lo*hi, fp32 accumulate): the
 *     "bf16x3" precision mode, fp32-grade results (the <= 1e-3 logits / exact-argmax bar of the parity
 *     tests) on the bf16 matrix cores.
 *     AFM_F16 = 3 (IEEE half, round-to-nearest-even): the operand format of the "fp16" precision mode, the reference's own
 *     GPU arithmetic (trainer/trainer.py:69, Lightning "16-mixed": fp16 autocast + GradScaler).  One v_mfma_*_f16 pass per
 *     product, fp32 accumulate; 11 significant bits keep the logits within the 1e-3 bar (measured 4e-4 .. 7e-4 at the BASELINE
 *     shapes), the 5-bit exponent is what the loss scaler of afm_adam_step / afm_scaler_update exists for.
 */
#ifndef AFM_HIP_H
#define AFM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (a binding compares afm_abi_version() AND afm_struct_size(i) with its own):
 *   4  afm_gemm_desc.k_live, afm_ln_shape.row_live (padded-row hints)
 *   5  exports added since 4: afm_cast_weights_batch, afm_comm_count, afm_struct_size(6) = afm_cast_item; afm_layernorm_bwd's
 *      workspace contract (afm_layernorm_bwd_ws); BOTH dropout streams redefined (two-level hashing: one full mixer per score-matrix
 *      row / per block of 64 elements, a pair mix per element pair) -- a library older than 5 produces different masks for the same
 *      (seed, site, index), so checkpoints' dropout positions and tests/dropmask.py belong to version >= 5.
 *   6  padded rows in the FORWARD pass of a training step: afm_gemm_desc.reserved2 bit 1 (k_live in the forward sense), afm_ln_shape.row_live
 *      read by afm_layernorm_fwd, afm_ln_shape.row_map (compaction of the live positions at the embedder), afm_attn_fwd honours
 *      reserved bit 6, afm_attn_shape.q_off / k_off (packed rows); exports added: afm_compact_plan, afm_permute_rows.  The ablation selector of AFM_ATTN_ABLATIONS builds moved to
 *      afm_attn_shape.reserved bits 20-27 (it overlapped the live selectors 16384 / 32768 / 65536). */
#define AFM_ABI_VERSION 6

enum { AFM_OK = 0, AFM_ERR_ARG = -1, AFM_ERR_UNSUPPORTED = -2, AFM_ERR_LAUNCH = -3 };
enum { AFM_F32 = 0, AFM_BF16 = 1, AFM_BF16X2 = 2, AFM_F16 = 3 };
enum { AFM_ACT_NONE = 0, AFM_ACT_RELU = 1, AFM_ACT_GELU = 2, AFM_ACT_GELU_BWD = 3,
       AFM_ACT_GELU_SAVE_GRAD = 4, AFM_ACT_MUL_SAVED = 5, AFM_ACT_GLU = 6, AFM_ACT_GLU_SAVE = 7, AFM_ACT_GLU_BWD = 8 };
enum { AFM_ALGO_AUTO = 0, AFM_ALGO_GENERIC = 1, AFM_ALGO_MFMA = 2 };

int afm_abi_version(void);
/* sizeof() of the ABI structs as compiled into the library (0 afm_dropout, 1 afm_gemm_desc, 2 afm_ln_shape,
 * 3 afm_attn_shape, 4 afm_patch_desc, 5 afm_beam_desc; -1 otherwise): a binding checks its own struct layouts against these,
 * so a stale library cannot be driven with newer descriptors. */
int afm_struct_size(int which);
const char* afm_error_string(int code);
/* Name of the kernel family the last afm_gemm / afm_attn_* call on this thread dispatched to
 * ("generic", "mfma_nt", "mfma_tn", ...): lets tests assert the fast path really ran. */
const char* afm_last_algo(void);
/* Padded-row hint of the last afm_gemm / afm_gemm_group / afm_layernorm_bwd call on this thread: 0 none given, 1 the kernels took it (tile /
 * k-step lists, row flags), -1 given but IGNORED (a kernel form without lists, an ineligible shape).  A caller that wants to stop filling
 * dead rows in its backward (reserved2 bit 3 / afm_ln_shape.flags bit 0 / afm_attn_shape.reserved bit 17 there) first runs a step WITH the
 * fills and checks that every hinted call answers 1: only then does nothing ever load those rows. */
int afm_last_hint(void);

/* ------------------------------------------------------------------------------------------
 * Dropout stream.  Every dropout site of the reference (torch `dropout`/`bernoulli_`,
 * custom_modeling.py:122-130,169-177) is a counter-based mask: keep(i) is a pure function
 * of (seed, site, element index i), so backward recomputes it and no mask tensor exists.
 * p == 0 disables.  Kept values are scaled by 1/(1-p).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  float p;
  uint32_t site;
  uint64_t seed;
} afm_dropout;

/* ------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:   T = op(A) . op(B) + bias ; T = act(T) ; T = dropout(T) ;
 *                             C = T + (residual ? residual : 0) + (accumulate ? C : 0)
 * op(A) is M x K:  transA == 0 -> A[m*lda + k],  transA == 1 -> A[k*lda + m]
 * op(B) is K x N:  transB == 0 -> B[k*ldb + n],  transB == 1 -> B[n*ldb + k]
 * Replaces aten::linear / addmm / mm and their backward:
 *   forward  y = x W^T + b   (torch:nn/functional.py linear; call sites modeling/utils.py:120-134,
 *            torch:nn/functional.py:5785 _in_projection_packed, torch:nn/modules/transformer.py:980-982,
 *            custom_modeling.py:145-149,486)                       transA=0, transB=1
 *   dgrad    dx = dy W                                              transA=0, transB=0
 *   wgrad    dW += dy^T x  (accumulate=1: gradient accumulation)    transA=1, transB=0
 * `pre_act` (optional, dtype/ld of C) receives T before the activation (kept for GELU backward).
 * `a_colsum` (wgrad form only) accumulates the column sums of A = dy: the bias gradient.
 * act == AFM_ACT_GELU_BWD fuses the GELU backward into a dgrad GEMM: T = dropout(T) * gelu'(U) with U read
 * from `pre_act` (the saved pre-activation, an INPUT in this mode): du = dropout'(dy W2) * gelu'(u).
 * act == AFM_ACT_GELU_SAVE_GRAD is the forward of the same pair with the backward factor precomputed:
 * C = dropout(gelu(T)) and pre_act <- keep * scale * gelu'(T) (same keep bits), so the dgrad needs neither
 * erf nor the dropout hash: act == AFM_ACT_MUL_SAVED computes C = T * pre_act (pre_act an INPUT, no bias /
 * dropout).  dropout'(dy W2) * gelu'(u) = (dy W2) * [keep * scale * gelu'(u)]: identical values.
 * Gated FFN (custom_modeling.py:137-152,184-199: W2 (gelu(W1 h) * (Wg h))), fused into the projections.  The up-projection
 * runs as ONE GEMM over the (2f x d) matrix [W1 ; Wg] whose rows are INTERLEAVED in groups of four (rows 8i..8i+3 = W1 rows
 * 4i..4i+3, rows 8i+4..8i+7 = Wg rows 4i..4i+3: afm_cast_weights with glu_rows = f), so one lane of the epilogue holds u and v of
 * the same hidden units.  bias / a_colsum / the wgrad's C stay in the reference's [W1 ; Wg] order: glu_rows = f in the descriptor
 * makes the kernels translate.   act == AFM_ACT_GLU:       N = 2f, C is M x f:  C = dropout(gelu(u) * v)
 *                                 act == AFM_ACT_GLU_SAVE:  also pre_act (M x 2f, interleaved) <- keep*scale*[gelu'(u) v | gelu(u)]
 *                                 act == AFM_ACT_GLU_BWD:   N = f (the dgrad dg = dy W2), C is M x 2f interleaved:
 *                                                           C = [dg * saved_a | dg * saved_b] = [du | dv], pre_act an INPUT
 * (dropout element index = row-major in the M x f tensor g, as afm_glu_fwd).  MFMA kernels, whole tiles only; other shapes
 * return AFM_ERR_UNSUPPORTED and the caller keeps the unfused afm_glu_fwd / afm_glu_bwd path.
 * bf16 / fp16 operands take the MFMA path (v_mfma_f32_16x16x32_bf16 | _f16 / 32x32x16, fp32 accumulate) when
 * shape/alignment allow; everything else takes the exact-fp32 FMA path.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t M, N, K;
  int32_t transA, transB;
  int32_t lda, ldb, ldc;
  int32_t a_dtype, b_dtype, c_dtype;
  const void* A;
  const void* B;
  void* C;
  const float* bias;      /* N, fp32, nullable */
  const void* residual;   /* M x N, dtype/ld of C, nullable */
  void* pre_act;          /* M x N, dtype/ld of C, nullable */
  float* a_colsum;        /* transA only, nullable: a_colsum[m] += sum_k A[k][m]  (the bias gradient
                             sum_rows dy, produced by the wgrad pass that already streams dy) */
  int32_t act;            /* AFM_ACT_* */
  int32_t accumulate;     /* C += ... */
  int32_t algo;           /* AFM_ALGO_* */
  int32_t reserved;
  afm_dropout drop;
  int32_t glu_rows;       /* f > 0: gated-FFN interleave (see above); wgrad form: rows of C / a_colsum are de-interleaved */
  int32_t reserved2;      /* bit 0 (pair dtype, act GELU_SAVE_GRAD / GLU_SAVE): the stored factors get their hi plane only -- the
                             consumer is the single-pass bf16 backward of the mixed precision mode, which never reads the lo plane.
                             bit 1 (2): k_live is meant in the FORWARD sense (below).
                             bit 2 (4): the rows k_live calls live are PACKED to the front of the token axis (afm_compact_plan mode 2): the
                             persistent NT kernels deal their row panels round-robin to the XCDs and the split-K units of the weight-gradient
                             kernels take every ksplit-th k-step, instead of contiguous bands / chunks that would leave the late ones with
                             nothing but dead rows.  Scheduling only, same results.
                             bit 3 (8), with bit 1: the dead tiles of C (and of a stored pre_act) are NOT written -- the caller vouches that its
                             buffers already hold finite values there (persistent buffers only these kernels ever write: stale rows of an earlier
                             step), so the zero fill -- HBM writes for rows nobody reads -- is left out.  Without bit 1 (the backward's hint): the same for the
                             zero tiles of the data-gradient forms, on the caller's word (afm_last_hint) that every consumer of C takes the hint too. */
  const uint8_t* k_live;  /* nullable: k_live[i] == 0 says the STORED rows 64 i .. 64 i + 63 of A (64 token positions) are all zero --
                             the padded positions of a training step's backward, whose activation gradients are exact zeros.
                             wgrad form (transA): the MFMA kernels leave those k-steps out (K / 64 bytes).  NT form without bias /
                             residual / accumulate, act NONE, MUL_SAVED or GLU_BWD: 256-row tiles of nothing but such blocks are written as
                             zeros without being computed (M / 64 bytes).  Same results either way; a hint, ignored elsewhere.
                             FORWARD sense (reserved2 bit 1, NT form, no residual / accumulate): k_live[i] == 0 says nobody reads rows 64 i ..
                             64 i + 63 of C (the padded positions of a training step, which are masked as keys everywhere and take no part in the
                             loss): 256-row tiles of nothing but such blocks are not computed, whatever the epilogue; their rows of C -- and of a
                             pre_act the epilogue stores -- are written as ZEROS (finite values: later kernels may still load them).  Rows of live
                             tiles are computed as always.  A hint: kernels without tile lists compute every row. */
} afm_gemm_desc;
int afm_gemm(const afm_gemm_desc* d, void* stream);

/* afm_gemm_group: `count` independent GEMMs, with the result of calling afm_gemm on each (C matrices must not alias).
 *   What it is for: the weight gradients of one transformer layer -- in the reference one `addmm` each inside
 *   torch.autograd's Linear backward (custom_modeling.py:117-160 layers under Lightning's backward, trainer/trainer.py:60-75).
 *   Each is a small matrix (dW: 512 x 512 .. 2048 x 512) over a long token axis; alone, each needs split-K 16 .. 64 to fill the
 *   chip, and every split adds dW once more through fp32 atomics.  Weight-gradient problems (transA, !transB, both operands
 *   AFM_BF16 or both AFM_F16, AFM_F32 C with accumulate = 1, no epilogue, M, N >= 256, K % 64 == 0) are fused, up to 8 per
 *   launch, into ONE grid sharing one split-K budget; any other problem runs through afm_gemm, in order. */
int afm_gemm_group(const afm_gemm_desc* descs, int32_t count, void* stream);

/* ------------------------------------------------------------------------------------------
 * Embedding rows.  nn.Embedding forward / embedding_dense_backward (modeling/utils.py:102-106,
 * 155-162): out[i,:] = table[ids[i],:] * (scale ? scale[i] : 1).  Backward adds
 * dout[i,:]*scale[i] into dtable[ids[i],:] and skips ids == padding_idx (its row gets no
 * gradient, as torch).  ids outside [0,V) are clamped (memory safety only).
 * ---------------------------------------------------------------------------------------- */
int afm_gather_rows(const int64_t* ids, const float* scale, const float* table, float* out,
                    int64_t n, int32_t d, int32_t V, void* stream);
int afm_scatter_add_rows(const int64_t* ids, const float* scale, const float* dout, float* dtable,
                         int64_t n, int32_t d, int32_t V, int64_t padding_idx, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim, eps inside the sqrt, biased variance (aten::native_layer_norm;
 * call sites modeling/utils.py:165-168,271, torch:nn/modules/transformer.py:946-950,1131-1143,
 * custom_modeling.py:350,399), with the two layout fusions the embedding needs
 * (modeling/utils.py:176-180): an optional positional table added AFTER the norm, and a row
 * remap so each modality writes straight into its slice of the concatenated sequence:
 *     out_row(r) = (r / seg_len) * out_seg_stride + out_off + (r % seg_len)
 *     y[out_row(r), :] = LN(x[r, :]) * gamma + beta + (pos ? pos[(out_off + r % seg_len), :] : 0)
 * seg_len == 0 means identity mapping.  x is fp32 (the residual stream); y is fp32 or bf16.
 * mean/rstd (rows, fp32) are saved for backward.
 * Residual fusion: when `add` is given (rows x d, dtype add_dtype) the kernel first forms
 * x_sum = x + add, writes it to `x_sum` (fp32, may alias x) and normalises THAT: the residual add
 * `x + dropout(branch)` of the pre-LN blocks (torch:nn/modules/transformer.py:946-950) rides on
 * the LayerNorm that reads the stream next, so the projection GEMMs write plain bf16 branches; with
 * `add_drop` the branch dropout is applied here too (this kernel is HBM-bound and has the VALU slack for
 * the hash, the GEMM epilogue does not).
 * Backward: dx[r,:] = (dres ? dres[r,:] : 0) + LN'(dy[out_row(r),:]); dgamma/dbeta are
 * ACCUMULATED (+=) into fp32 buffers; `partial` is workspace of afm_layernorm_bwd_ws_floats() floats (0 for the shapes the
 * vectorised kernel takes -- d % 8 == 0, d <= 2048, rows >= 64 --, which adds its block sums with atomics: `partial` may then be null).
 * `dx_drop` (optional, dtype y_dtype, rows x d): dropout(dx) with stream `drop`, i.e. the gradient
 * of the dropped-out residual branch that was added in front of this LayerNorm, handed to the
 * branch's GEMMs in their operand dtype without another pass over dx.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t rows;
  int32_t d;
  int32_t y_dtype;
  int64_t seg_len, out_seg_stride, out_off;
  float eps;
  int32_t add_dtype;      /* dtype of `add` (forward only) */
  afm_dropout add_drop;   /* forward only: x_sum = x + dropout(add) with this stream (p = 0: plain add); the
                             element index is row-major in `add`, the same index the GEMM epilogue and the
                             backward's dx_drop use for that site */
  const uint8_t* row_live; /* nullable, rows / 64 bytes (rows % 64 == 0), identity row mapping only.  Backward: row_live[i] == 0 says rows
                             64 i .. 64 i + 63 of dy (and of dres) are all zero (padded positions): their dx / dx_drop rows are written as
                             zeros without reading anything and they add nothing to dgamma / dbeta.  Forward (ABI 6): row_live[i] == 0 says
                             nobody reads those rows of the outputs (padded positions of a training step): nothing is loaded, y / x_sum /
                             mean / rstd get zeros there.  A hint: the scalar kernels (d % 8 != 0, rows < 64) ignore it. */
  int32_t flags;           /* bit 0, with row_live: the rows row_live calls dead are NOT written (no zeros).  Forward: the caller's y / x_sum / mean /
                             rstd already hold finite values there (a persistent buffer only these kernels ever write).  Backward: the caller has
                             checked (afm_last_hint) that every consumer of dx / dx_drop takes the hint too, so nothing loads those rows. */
  int32_t reserved;
  const int32_t* row_map;  /* nullable, placement form (seg_len != 0) only, forward and backward: position p = out_off + r % seg_len of sample
                             b = r / seg_len lives in row row_map[b * out_seg_stride + p] of the concatenated-sequence matrix instead of row
                             b * out_seg_stride + p (afm_compact_plan's dest: live positions moved to the front of the sample's slot, or the
                             whole batch packed).  The positional row added stays pos[p]: positions are encoded before the move. */
} afm_ln_shape;
int afm_layernorm_fwd(const afm_ln_shape* s, const float* x, const float* gamma, const float* beta,
                      const float* pos, void* y, float* mean, float* rstd, const void* add,
                      float* x_sum, void* stream);
int64_t afm_layernorm_bwd_ws_floats(const afm_ln_shape* s);
int afm_layernorm_bwd(const afm_ln_shape* s, const void* dy, const float* x, const float* gamma,
                      const float* mean, const float* rstd, const float* dres, float* dx,
                      float* dgamma, float* dbeta, float* partial, void* dx_drop,
                      const afm_dropout* drop, void* stream);

/* Embedding rows WITHOUT the per-modality LayerNorm (`multimodal_norm: false`, modeling/utils.py:165-168 skipped), same layout
 * fusion as above:  gather == 0: y[out_row(r), :] = x[r, :] + (pos ? pos[out_off + r % seg_len, :] : 0);
 *                   gather == 1: y[r, :] = x[out_row(r), :]  (backward: the modality's slice of the stream gradient). */
int afm_place_rows(const float* x, const float* pos, float* y, int64_t rows, int32_t d, int64_t seg_len,
                   int64_t out_seg_stride, int64_t out_off, int32_t gather, void* stream);

/* ------------------------------------------------------------------------------------------
 * Padded positions out of a training step's forward pass (ABI 6).  The collator pads every modality to its own length, so a sample's
 * padding sits in several runs inside its S-row slot (data/datamodules.py:230-351 builds the mask, one run per modality); the layer
 * stacks mask those positions as keys everywhere (custom_modeling.py:238,312-318) and pool them out of the alignment head (:469-470), so
 * nothing reads their rows.  afm_compact_plan turns a key-padding mask into a per-sample stable partition -- live positions first, in
 * order -- and the block flags the kernels above take:
 *   dest[b*S + s]      the ROW (of the B*S-row activation matrices) position s of sample b moves to
 *   seq_off[b]         first row of sample b, B + 1 entries (seq_off[B] = rows in use)
 *   pad_out[b*S + p]   the mask over the sample's positions in their new order (p-th position of its slot; 1 for p >= n_live[b] when
 *                      compacting); may alias key_pad only when compact == 0
 *   live64[i]          1 if rows 64 i .. 64 i + 63 hold a live position: k_live / row_live of the BACKWARD (exact zeros elsewhere)
 *   live_tile[i]       the same widened to whole groups of tile_rows rows (a multiple of 64 dividing S: 256, the tallest GEMM tile):
 *                      1 if the group holds a live position -- k_live / row_live in the FORWARD sense.  Every forward kernel then
 *                      agrees on which rows exist, whatever its own tile height.
 *   n_live[b]          live positions of sample b
 * compact == 0: rows stay where the collator put them (dest = identity, seq_off[b] = b S); flags only.
 * compact == 1: per-sample partition inside the sample's own S-row slot (seq_off[b] = b S): row b S + rank.
 * compact == 2: PACKED -- the samples' slots shrink to their live length rounded up to 32 rows (what one wave of the attention kernels owns)
 *               and follow each other: seq_off[b] = sum_{b' < b} ceil32(n_live[b']), live positions first inside the slot, the padded positions of the batch behind
 *               all slots (rows seq_off[B] .. B S, in order).  The attention kernels address such rows through afm_attn_shape.q_off / k_off.
 * S <= 4096, S % tile_rows == 0.  Encoder self-attention, cross-attention over the memory and the masked mean are indifferent to the
 * order of the key positions once the mask moves with them; positional encodings are added before the move (afm_ln_shape.row_map).
 * afm_permute_rows moves fp32 rows through such a map where no LayerNorm does it on the way:
 *   gather == 0: y[map[i], :] = x[i, :]       gather == 1: y[i, :] = x[map[i], :]        (i over the B*S rows)
 * ---------------------------------------------------------------------------------------- */
int afm_compact_plan(const uint8_t* key_pad, int32_t B, int32_t S, int32_t tile_rows, int32_t compact, int32_t* dest, int32_t* seq_off,
                     uint8_t* pad_out, uint8_t* live64, uint8_t* live_tile, int32_t* n_live, void* stream);
int afm_permute_rows(const float* x, float* y, const int32_t* map, int32_t B, int32_t S, int32_t d, int32_t gather, void* stream);

/* ------------------------------------------------------------------------------------------
 * Masked multi-head attention, flash style (no T_q x T_k tensor in HBM).
 * F.scaled_dot_product_attention as reached from nn.MultiheadAttention
 * (torch:nn/functional.py:6206-6640; reference call sites custom_modeling.py:122-130 encoder
 * self-attention, :169-177 + :308-318 decoder causal self-attention and cross-attention):
 *     P = softmax(Q K^T * scale + mask) ; O = dropout(P) V
 * Element (b, t, h, j) of Q lives at Q[(b*Tq + t)*ldq + h*dh + j] (same for K/V with Tk, O with
 * Tq): packed in-projection outputs are addressed in place.  key_pad (B x Tk, 1 = masked key,
 * nullable) is src_/tgt_/memory_key_padding_mask; causal != 0 masks key > query
 * (nn.Transformer.generate_square_subsequent_mask).  A row whose keys are all masked yields
 * zeros (torch _safe_softmax).  lse (B x H x Tq fp32) = log-sum-exp of the scaled, masked
 * scores (+inf for an all-masked row), kept for backward.
 * Backward recomputes P from lse; delta (B x H x Tq fp32) is workspace.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t B, H, Tq, Tk, dh;
  int32_t dtype;
  int32_t ldq, ldk, ldv, ldo;
  int32_t causal;
  int32_t algo;
  float scale;
  int32_t reserved;       /* afm_attn_bwd, bits 0-1: 0 = dQ and dK/dV kernels; 1 = dQ (+ delta) only; 2 = dK/dV only (delta from an
                             earlier call): lets bench.py / the profiler time the two backward kernels separately.
                             bit 6 (64): afm_attn_bwd, self-attention (Tq == Tk, key_pad given): the caller vouches that dO is zero in the query rows
                             key_pad marks (a training step's padded positions); the single-pass kernels then skip them -- dQ rows stay
                             zero, dK / dV lose exact zeros.  afm_attn_fwd (ABI 6), same condition: nobody reads the outputs of the query rows
                             key_pad marks; workgroups (128 queries) of nothing but such rows write O = 0, lse = +inf (the all-masked-row
                             convention) and return.  bit 17 (131072), afm_attn_fwd with q_off: the blocks beyond the slots write nothing to the dead
                             tail of O (the caller's O already holds finite values there); afm_attn_bwd with q_off / k_off: nothing to the dead tail of dQ / dK /
                             dV (every consumer takes the hint: afm_last_hint).  bit 5 (32): forward reads drop_bits (afm_attn_drop_bits_fill ran before); bit 4 (16): the 8-wave staggered forms of the single-pass forward / dQ kernels (Tq >= 256), see DESIGN.md 4.
                             afm_attn_bwd, dK/dV kernel selection (A / B tests; every form gives bit-identical dK / dV): bit 7 (128) the round-3
                             kernel instead of the software-pipelined one (csrc/afm_attn_pipe_impl.h: default where there is no causal mask, Tq % 64 == 0
                             and dropout runs through drop_bits or is off); bit 8 (256) its eight-wave form; bit 9 (512) its form with 64 keys per
                             wave.  Round 5: the four-wave pipelined kernel runs on v_mfma_f32_16x16x32 by default (csrc/afm_attn_pipe16_impl.h;
                             equal to the others to rounding, not bit for bit); bit 14 (16384) keeps its 32 x 32 x 16 form, bit 12 (4096) selects the
                             round-3 kernel restated on 16 x 16 x 32, bits 10-11 (1024, 2048) the 16 x 16 x 32 dQ kernels (the default without dropout and where the
                             hash is re-evaluated; bit 15 (32768) keeps the 32 x 32 x 16 dQ kernel there); bit 16 (65536) the short-query dK/dV kernel
                             (csrc/afm_attn_sq_impl.h: Tq <= 192 < 256 <= Tk, no causal mask; an A / B form); bit 18 (262144), round 6: dQ, dK and dV in ONE
                             kernel where Tq <= 128, dense query rows (k_off allowed without a causal mask), a causal mask only with Tq == Tk, dropout through drop_bits
                             or off, bits 0-1 zero (csrc/afm_attn_fsq_impl.h: the decoder's cross- and self-attention; the two kernels elsewhere; delta as always,
                             dQ / dK / dV to rounding).  afm_attn_fwd: bits 10-11 select
                             the forward restated on 16 x 16 x 32 (csrc/afm_attn_fwd16_impl.h; bit 11: its three-workgroup build; A / B forms).  Bits 20-27 select timing ablations in AFM_ATTN_ABLATIONS builds (never in the product library). */
  const uint8_t* key_pad;
  afm_dropout drop;
  /* batch strides in ELEMENTS of Q, K, V, O (0 = dense: Tq*ldq, Tk*ldk, Tk*ldv, Tq*ldo).  Non-dense
   * strides address a KV cache laid out (batch, Tmax, 2d) during incremental decode; forward only. */
  int64_t sqb, skb, svb, sob;
  /* Optional keep-bit tensor of the attention-probability dropout (nullable; only read / written when drop.p > 0):
   * B*H * NQ * NK blocks of 16 uint64 with NQ = 4 ceil(Tq/128) query blocks and NK = 2 ceil(Tk/64) key blocks of 32 (whole kernel
 * tiles: blocks past Tq / Tk exist and are never meaningful).  Block (bh, qb, kb), word r, bit l = keep of query 32 qb + (l & 31),
   * key 32 kb + (r & 3) + 8 (r >> 2) + 4 (l >> 5): the 64-lane masks of the kernels' score registers.  afm_attn_fwd writes it
   * (the MFMA kernels as a by-product of their own dropout, through the scalar store path), afm_attn_bwd's MFMA kernels read it
   * instead of re-evaluating the hash per score: the dQ kernel as SGPR lane masks (one v_cndmask per score), the dK/dV kernel as one
   * word per lane.  The bits ARE the counter-based stream above, so results do not depend on whether the tensor is used. */
  uint64_t* drop_bits;
  /* PACKED rows (ABI 6, nullable, B + 1 int32 each; afm_compact_plan's seq_off): sample b's query-side rows (Q, O, dO, dQ) start at row
   * q_off[b] instead of b * Tq, its key-side rows (K, V, dK, dV) at k_off[b] instead of b * Tk; off[b + 1] - off[b] is the sample's slot, a
   * multiple of 32 rows that holds its live positions first (key_pad, lse, delta and drop_bits keep their padded (B, T) indexing).
   * 128-row blocks (and, in a slot's partly used last block, 32-row waves) beyond a slot have no rows of their own: they write zeros to
   * the dead tail [off[B], B * T) of O / dQ / dK / dV instead (in-sample row r >= slot <-> tail row off[B] + (b T - off[b]) + (r - slot)),
   * so every row of an output is written.  Single-pass MFMA kernels only (dh 64, no causal mask,
   * T % 128 == 0, key_pad given for a packed key side, dropout through drop_bits or off, q_off == k_off when both are set -- the encoder's
   * self-attention, where the padded-query skip of reserved bit 6 is implied); AFM_ERR_UNSUPPORTED otherwise, nothing launched. */
  const int32_t* q_off;
  const int32_t* k_off;
} afm_attn_shape;
int afm_attn_fwd(const afm_attn_shape* s, const void* Q, const void* K, const void* V, void* O,
                 float* lse, void* stream);
/* afm_attn_drop_bits_fill: write s->drop_bits ahead of the forward -- the bits depend on (drop.seed, drop.site, B, H, Tq, Tk) only,
 * so the caller can run this on another stream, under an HBM-bound kernel that precedes the attention block (the hash is pure
 * vector work).  A forward whose shape carries reserved |= 32 then READS the tensor (one select per score) instead of hashing and
 * writing it; results are bit-identical either way.  AFM_ERR_UNSUPPORTED where the single-pass MFMA kernels would not run the
 * shape (the caller then leaves the flag off).  Reference: the dropout inside F.scaled_dot_product_attention / nn.MultiheadAttention
 * (custom_modeling.py:117-160), whose mask torch also draws ahead of the product. */
int afm_attn_drop_bits_fill(const afm_attn_shape* s, void* stream);
int afm_attn_bwd(const afm_attn_shape* s, const void* Q, const void* K, const void* V, const void* O,
                 const void* dO, const float* lse, float* delta, void* dQ, void* dK, void* dV,
                 int32_t lddq, int32_t lddk, int32_t lddv, void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise pieces of the FFN and the residual branches.
 * afm_glu_fwd:  g = act(u) * (v ? v : 1), then dropout       (torch:nn/modules/transformer.py:980-982,
 *               custom_modeling.py:145-148,192-195); act = AFM_ACT_GELU (exact erf form, the reference default) or
 *               AFM_ACT_RELU: config.activation_function goes straight to the torch layers (custom_modeling.py:127,174)
 * afm_glu_bwd:  du = dg' * (v ? v : 1) * act'(u), dv = dg' * act(u), dg' = dropout'(dg)
 * afm_dropout_cast: y = dropout(x) cast to y_dtype  (backward of the dropout1/2/3 residual branches:
 *               the fp32 residual gradient masked and handed to the GEMMs in their operand dtype)
 * u, v, g are rows x f with row strides ld*, dtype `dtype`.
 * ---------------------------------------------------------------------------------------- */
int afm_glu_fwd(const void* u, const void* v, void* g, int64_t rows, int32_t f, int32_t ldu,
                int32_t ldv, int32_t ldg, int32_t dtype, int32_t act, const afm_dropout* drop, void* stream);
int afm_glu_bwd(const void* u, const void* v, const void* dg, void* du, void* dv, int64_t rows,
                int32_t f, int32_t ldu, int32_t ldv, int32_t lddg, int32_t lddu, int32_t lddv,
                int32_t dtype, int32_t act, const afm_dropout* drop, void* stream);
int afm_dropout_cast(const float* x, void* y, int64_t rows, int32_t n, int32_t ldx, int32_t ldy,
                     int32_t y_dtype, const afm_dropout* drop, void* stream);
/* ReLU backward: dx[i] = act[i] > 0 ? dy[i] : 0 (fp32; dx may alias dy).  The hidden layers of the patch
 * embedders (linear_2_layer / linear_3_layer, modeling/utils.py:107-136) and of the alignment head
 * (custom_modeling.py:363-396). */
int afm_relu_bwd(const float* dy, const float* act, float* dx, int64_t n, void* stream);
/* dst = (dst_dtype) src, any pair of AFM_F32 / AFM_BF16 / AFM_BF16X2 / AFM_F16, rows x n with row strides lds / ldd
 * (elements of the respective dtype's planes, see the AFM_BF16X2 convention above). */
int afm_convert(const void* src, int32_t src_dtype, int32_t lds, void* dst, int32_t dst_dtype, int32_t ldd,
                int64_t rows, int32_t n, void* stream);
/* Column sums (bias gradients): out[j] (+)= sum_i x[i*ld + j].  Backward of the bias add of
 * every aten::linear above. */
int afm_colsum(const void* x, float* out, int64_t rows, int32_t n, int32_t ld, int32_t dtype,
               int32_t accumulate, void* stream);
/* y[i] += x[i] (fp32): e.g. the positional-table gradient summed over the batch. */
int afm_add_inplace(float* y, const float* x, int64_t n, void* stream);
/* out[s,:] (+)= sum_b x[(b*S + s),:]  (batch reduction for the learned positional encoding
 * gradient, modeling/utils.py:267-271). */
int afm_batch_sum(const float* x, float* out, int32_t B, int64_t S, int32_t d, int32_t accumulate,
                  void* stream);
/* fp32 -> bf16 copies of the weights for the MFMA GEMMs: dst (rows x cols) and, when dst_t is
 * given, the transpose (cols x rows) used by dgrad. */
int afm_cast_bf16(const float* src, void* dst, void* dst_t, int32_t rows, int32_t cols, void* stream);
/* The same for the split-pair dtype: dst is (rows x cols) AFM_BF16X2 with row stride 2*cols, dst_t the
 * transpose (cols x rows) with row stride 2*rows. */
int afm_cast_x2(const float* src, void* dst, void* dst_t, int32_t rows, int32_t cols, void* stream);
/* Both of the above for dtype AFM_BF16, AFM_BF16X2 or AFM_F16 (glu_rows = 0: plain copies), and with the gated-FFN row
 * interleave: glu_rows = f > 0 (rows = 2f): source row
 * r (r < f: W1, else Wg) lands in row ((j>>2)<<3) + (j&3) + 4*(r >= f), j = r mod f, of dst (column of dst_t). */
int afm_cast_weights(const float* src, void* dst, void* dst_t, int32_t rows, int32_t cols, int32_t dtype, int32_t glu_rows,
                     void* stream);
/* The same for a whole list of matrices in ONE launch (the weight shadows behind an optimiser step: 81 launches of a few
 * microseconds each at the c2 model otherwise).  `items` is a DEVICE array; item i covers the 64 x 64 tiles [tile0, tile0 +
 * ceil(rows/64) * ceil(cols/64)) of the launch, tile0 ascending from 0, `tiles` their total.  dst or dst_t may be null per item.
 * Replaces: the per-parameter `.to(dtype)` casts autocast performs inside every torch Linear (torch/amp/autocast_mode.py). */
typedef struct afm_cast_item {
  const float* src;
  void* dst;
  void* dst_t;
  int32_t rows, cols, glu_rows, tile0;
} afm_cast_item;
int afm_cast_weights_batch(const afm_cast_item* items, int32_t n, int32_t tiles, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Encoder alignment head (custom_modeling.py:363-396 network, 453-475 use).
 * masked mean: out[b,:] = sum_s keep[b,s] x[b,s,:] / sum_s keep[b,s]   (keep = !key_pad;
 *   `(last_hidden_state * mask).sum(1) / mask.sum(1)`, custom_modeling.py:469-470); x fp32 or bf16.
 * bwd: dx[b,s,:] (+)= keep[b,s] dy[b,:] / n_b  (fp32; accumulate = 0 also writes the zeros of the
 *   padded rows, so it can initialise the encoder-output gradient).
 * loss: pred = sigmoid(z); kind 0 mse = mean (pred-t)^2, 1 mae = mean |pred-t| (nn.MSELoss / nn.L1Loss),
 *   2 sid = the reference's own kl_div pair (modeling/utils.py:8-22): both clamped to >= 1e-16,
 *   [sum p log(p/t) + sum t log(t/p)] / B.  stats[0] += loss; dz = grad_scale * (scale_dev ? scale_dev[0] : 1) * dloss/dz.
 * The Linear / ReLU / centre-tap Conv1d layers in between are afm_gemm calls.
 * ---------------------------------------------------------------------------------------- */
enum { AFM_ALIGN_MSE = 0, AFM_ALIGN_MAE = 1, AFM_ALIGN_SID = 2 };
int afm_masked_mean_fwd(const void* x, int32_t x_dtype, const uint8_t* key_pad, int32_t B, int32_t S,
                        int32_t d, float* out, void* stream);
int afm_masked_mean_bwd(const float* dy, const uint8_t* key_pad, int32_t B, int32_t S, int32_t d,
                        float* dx, int32_t accumulate, void* stream);
int afm_align_loss(const float* z, const float* target, int32_t kind, int32_t B, int32_t n,
                   float grad_scale, const float* scale_dev, float* stats, float* dz, void* stream);

/* ------------------------------------------------------------------------------------------
 * Mixture generator (data/datasets.py:58-141 mix_spectra + normalize_spectrum
 * :49-56).  For output row r: combined = np.average(table[idx[r, 0..c)], weights=ratio, axis=0)
 * evaluated in fp64 in numpy's order (sum_c table[idx[r,c]] * ratio[c], then / sum(ratio));
 * normalize != 0: min / max of the combined row, negatives clipped to 0, then
 * (x - min) / (max - min) (all zeros when max == min); zero-padded to out_len (1800 in the reference)
 * and rounded to fp32 (the collator's torch.Tensor(...)).  table (N x L) fp32, idx (n x c) int64.
 * Pinned: tests/golden/mixture.npz holds the records the reference's own two functions yield on a seeded table
 * (oracle/make_mixture_goldens.py runs them out of the reference's source file; the module itself cannot be imported
 * here: omegaconf is missing); the oracle's restatement and this kernel reproduce them bit for bit.
 * ---------------------------------------------------------------------------------------- */
int afm_mix_spectra(const float* table, int64_t N, int32_t L, const int64_t* idx, int32_t n, int32_t c,
                    const double* ratio, int32_t normalize, int32_t out_len, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Input path: PatchPreprocessor.__call__ (data/preprocessing/patches.py:54-107)
 * on the device.  spectra (B x L) fp32 rows, present[b] = 0 marks the reference's `None` spectrum
 * (zeros BEFORE standardisation, patches.py:63-67).  Per row, in the reference's order:
 *   interpolation (patches.py:47-52): scipy interp1d from the grid 400 + 2 i (L points) onto
 *     650 + 2 i (1625 points), evaluated in fp64 as slope * (x_new - x_lo) + y_lo, rounded to fp32;
 *   standardise in fp32: (x - (float)mean) / (float)std (patches.py:76);
 *   trim to n = len / patch_size whole patches; patches = view (n, patch_size) or, for
 *     step = patch_size / overlap < patch_size, unfold(patch_size, step) (patches.py:79-90);
 *   derivative (patches.py:92-96): torch.gradient of the RAW spectrum (central differences, one-sided
 *     ends), trimmed, n more patches appended;
 *   mask (patches.py:99-105): masking -> (patch sum == 0), else 1 for every patch of an absent row.
 * patches: (B, P, patch_size) fp32, or (P, B, patch_size) when seq_first (the layout the collator
 * hands to the model, datamodules.py:201-218); mask: (B, P) or (P, B) bytes, 1 = pad.
 * afm_patch_count returns P for the descriptor (<= 0: invalid descriptor).
 * ---------------------------------------------------------------------------------------- */
typedef struct afm_patch_desc {
  int32_t B, L;
  int32_t patch_size, step;      /* step = patch_size / overlap */
  int32_t interpolation, derivative, masking, seq_first;
  double mean, std;              /* PatchPreprocessor.initialise statistics (non-zero entries) */
} afm_patch_desc;
int32_t afm_patch_count(const afm_patch_desc* d);
int afm_patch_preprocess(const afm_patch_desc* d, const float* spectra, const uint8_t* present,
                         float* patches, uint8_t* mask, void* stream);

/* ------------------------------------------------------------------------------------------
 * LM-head loss.  nn.CrossEntropyLoss() over logits.view(-1,V) with ignore_index -100
 * (custom_modeling.py:490-491) fused with the teacher-forced argmax of
 * HFWrapper._calc_token_acc (wrapper.py:641-655).
 * fwd: per row log-sum-exp and argmax (first maximal index, as torch.argmax); stats[0] += sum of
 *      -log_softmax[label] over labels != -100, stats[1] += number of such labels.
 * bwd: dlogits = (softmax - onehot(label)) * grad_scale * (scale_dev ? scale_dev[0] : 1) / stats[1] for kept rows, 0 otherwise;
 *      scale_dev (DEVICE, nullable) is the dynamic loss scale of the fp16 mode (word 0 of the scaler state, below).
 * ---------------------------------------------------------------------------------------- */
int afm_ce_fwd(const float* logits, const int64_t* labels, int64_t rows, int32_t V, int32_t ld,
               float* row_lse, int64_t* argmax, float* stats, void* stream);
int afm_ce_bwd(const float* logits, const int64_t* labels, const float* row_lse, const float* stats,
               float grad_scale, const float* scale_dev, void* dlogits, int32_t dl_dtype, int32_t lddl, int64_t rows,
               int32_t V, int32_t ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * Beam search bookkeeping on the device: what transformers' GenerationMixin beam search does on the
 * host between two decoder steps when called as the reference calls it (modeling/wrapper.py:306-313,443-451:
 * num_beams = num_return_sequences = k, length_penalty 1, early_stopping False, forced EOS at max_length).
 * afm_beam_step, one launch per generated token: log-softmax of the (B*k x V) logits + running beam scores, the 2k best
 * of the k*V candidates per sample in descending order, then per sample: an EOS candidate inside the top k closes a
 * hypothesis (score = sum_logprobs / generated_len, the k best are kept), the first k other candidates continue;
 * the sample is done when it holds k hypotheses and worst_kept >= best / generated_len, best = the step's best
 * candidate (stop_rule 0, transformers 4.48.3: the reference's pin) or its best running beam (stop_rule 1, 5.x).
 * Writes the new running sequences (seq_out rows = seq_in[source row] + token), beam scores, beam_idx (source row of
 * every new row: the KV-cache reorder) and n_open = number of samples still open (the only word the host reads).
 * afm_beam_finalize: open beams of unfinished samples become hypotheses; the k best per sample, best first, as
 * `tokens, eos (if shorter than max_length), pad...` into out (B*k x max_length), their scores and lengths.
 * afm_cache_reorder: dst block r = first used_bytes of src block beam_idx[r] (blocks of block_bytes: one row's KV cache).
 * Pinned to transformers' own generate on a table-lookup model: tests/golden/beam_cases.npz.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t B, k, V, ldl;          /* logits (B*k x V) fp32, row stride ldl */
  int32_t cur_len, max_length;   /* tokens in every running row before this step (>= 1: the start token) */
  int32_t eos, pad, stop_rule, Lmax;   /* Lmax: row stride of seq_in / seq_out / hyp_seq (>= max_length) */
  const float* logits;
  const int64_t* seq_in;         /* (B*k x Lmax) running sequences */
  int64_t* seq_out;
  float* beam_scores;            /* (B*k) in/out: initialise to 0 for beam 0 of every sample, -1e9 for the others */
  int32_t* beam_idx;             /* (B*k) out */
  int64_t* hyp_seq;              /* (B x k x Lmax) closed hypotheses (without their eos) */
  float* hyp_score;              /* (B x k) */
  int32_t* hyp_len;              /* (B x k) */
  int32_t* hyp_count;            /* (B) zero-initialised */
  int32_t* done;                 /* (B) zero-initialised */
  int32_t* n_open;               /* (1) out */
} afm_beam_desc;
int afm_beam_step(const afm_beam_desc* d, void* stream);
int afm_beam_finalize(const afm_beam_desc* d, int64_t* out, float* out_scores, int32_t* out_len, void* stream);
int afm_cache_reorder(const void* src, void* dst, const int32_t* beam_idx, int32_t rows, int64_t block_bytes,
                      int64_t used_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser over ONE flat fp32 parameter buffer (all tensors of the model are views of it).
 * afm_sumsq:  out[0] += sum g[i]^2   (torch.nn.utils.clip_grad_norm_, Lightning
 *             gradient_clip_val, trainer/trainer.py:65).  Two stages through `partial` (workspace of AFM_SUMSQ_PARTIALS
 *             floats), no atomics: the sum is bit-reproducible, so data-parallel replicas derive identical clip coefficients
 *             from the all-reduced buffer and stay bit-identical
 * afm_adam_step: torch.optim.Adam / AdamW (wrapper.py:29,333-338) with the clip folded in:
 *     coef = grad_mult * min(1, max_norm / (sqrt(sumsq[0]) * grad_mult + 1e-6)) ; g = g * coef
 *     (grad_mult = 1/world under data parallelism: sumsq is taken over the SUMMED gradients, the clip acts on their mean)
 *     Adam  (decoupled == 0): g += wd * p        AdamW (decoupled == 1): p *= 1 - lr*wd
 *     m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
 *     p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * hyper (DEVICE, 10 floats) = { lr, beta1, beta2, eps, weight_decay, 1-beta1^t, 1-beta2^t,
 *     max_norm (<=0: no clip), grad_mult, decoupled } so a captured graph replays with new
 * values.  g is zeroed when zero_grad != 0.  p_lowp (nullable) receives the 16-bit copy of p (lowp_dtype AFM_BF16 / AFM_F16).
 * Dynamic loss scaling of the fp16 mode (torch.amp.GradScaler as Lightning's "16-mixed" drives it, trainer/trainer.py:69):
 * `scaler` (DEVICE, 4 floats, nullable) = { loss scale S, growth tracker, optimiser steps taken, steps skipped }.  With it the
 * gradients in g are S times the true ones (afm_ce_bwd / afm_align_loss multiplied by scaler[0]): grad_mult is divided by S,
 * a non-finite sumsq[0] SKIPS the step (p, m, v untouched, g zeroed) and the bias corrections use t = scaler[2] + 1, the
 * number of steps actually taken (torch's per-parameter `step`), instead of hyper[5..6].  afm_scaler_update then applies
 * GradScaler.update(): skipped -> S *= backoff, tracker = 0; else tracker += 1 and after `interval` good steps S *= growth.
 * ---------------------------------------------------------------------------------------- */
#define AFM_SUMSQ_PARTIALS 2048
int afm_sumsq(const float* g, int64_t n, float* out, float* partial, void* stream);
int afm_adam_step(float* p, float* g, float* m, float* v, int64_t n, const float* hyper,
                  const float* sumsq, void* p_lowp, int32_t lowp_dtype, int32_t zero_grad, const float* scaler, void* stream);
int afm_scaler_update(float* scaler, const float* sumsq, float growth, float backoff, int32_t interval, void* stream);

/* ------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange (reference: Lightning DDP, trainer/trainer.py:58 + cli/training.py:49-59: one
 * all-reduce of the gradients per optimiser step).  One process per GPU; the flat fp32 gradient buffer is summed over the ranks
 * in buckets, back to front while the backward pass still runs, each bucket as ONE RCCL all-reduce (xGMI rings inside a node)
 * enqueued on `stream` (the caller's side stream; events order it against the compute stream).  The mean (1/world) is folded
 * into afm_adam_step's grad_mult.  librccl.so is resolved at run time (dlopen; AFM_RCCL_PATH overrides): AFM_ERR_UNSUPPORTED
 * when it cannot be found.  Bootstrap: rank 0 calls afm_comm_unique_id, the 128 bytes travel to the other ranks by any means
 * (the host side uses its torch.distributed store), every rank calls afm_comm_create with the current device selected.
 * ---------------------------------------------------------------------------------------- */
typedef struct afm_comm afm_comm;
int afm_comm_unique_id(void* out128);
int afm_comm_create(afm_comm** out, const void* id128, int32_t rank, int32_t world);
int afm_allreduce_bucket(afm_comm* comm, float* buf, int64_t n, void* stream);      /* buf[0..n) = sum over ranks, in place */
int afm_comm_count(afm_comm* comm, int32_t* rank, int32_t* world);                 /* what RCCL itself reports: ncclCommUserRank / ncclCommCount */
int afm_comm_destroy(afm_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* AFM_HIP_H */
