"""Hand-scheduled forward / backward of the spectra->SMILES encoder-decoder on libafm_hip.so.

This is the MI355X-native counterpart of `CustomModel.forward` + autograd
(reference custom_modeling.py:420-508, modeling/utils.py:142-182): the layer schedule is
written out explicitly (no autograd graph, no torch.nn), every arithmetic step is one call
into the C ABI (ops.py), activations needed by backward stay resident in HBM (288 GB: nothing
is recomputed except dropout masks and attention probabilities), the residual stream, LayerNorm
statistics, logits, loss and all parameter gradients are fp32, GEMM / attention operands are
`compute_dtype` (bf16 for speed, fp32 for the exact-parity mode).

Row conventions: encoder rows R = B*S, decoder rows Rt = B*T, all tensors batch-first 2-D
(rows, features) views so a (b, t) token is row b*T + t.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import os

import torch

_DEBUG_LIVE = os.environ.get("AFM_DEBUG_LIVE", "0") == "1"      # verify the padded-row hints against the data (host sync)

from . import ops
from .lib import (ACT_GELU, ACT_GELU_BWD, ACT_GELU_SAVE_GRAD, ACT_GLU, ACT_GLU_BWD, ACT_GLU_SAVE, ACT_MUL_SAVED, ACT_NONE, ACT_RELU,
                  AFM_BF16, ALGO_AUTO)
from .params import align_dict, PATCH_TYPES, TEXT_TYPES, ParamStore, build_specs, patch_layers
from .x2 import X2

# compute dtypes: torch.float32 (exact-fp32 FMA kernels), torch.bfloat16 (one bf16 MFMA pass per product),
# torch.float16 (one fp16 MFMA pass per product + dynamic loss scaling: the reference's own GPU precision, Lightning "16-mixed",
# trainer/trainer.py:69), X2.dtype = "bf16x2" (split bf16 pairs, three MFMA passes per product: fp32-grade results)
BF16X3 = X2.dtype
LOSS_SCALE_INIT = 65536.0      # torch.amp.GradScaler defaults (init_scale, growth_factor, backoff_factor, growth_interval)


def sincos_table(d_model: int, max_len: int) -> torch.Tensor:
    """SincCosPositionalEncoding._positional_encs (modeling/utils.py:226-239): interleaved
    sin/cos, float32 arithmetic in the reference's order of operations (setup-time, host)."""
    frac = (torch.arange(0, d_model, 2, dtype=torch.float64) / d_model).float()
    div = 10000 ** frac
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    ang = pos * div.reciprocal()
    return torch.stack((torch.sin(ang), torch.cos(ang)), dim=2).flatten(1)[:, :d_model].contiguous()


class Seq2SeqEngine:
    """Parameters + explicit fwd/bwd schedule.  `cfg` holds the CustomConfig fields
    (custom_modeling.py:43-65) as a plain dict."""

    def __init__(self, cfg: Dict[str, Any], data_config: Dict[str, Any], target_modality: str,
                 vocab_out: int, device="cuda:0", compute_dtype=torch.bfloat16, seed: int = 3247,
                 algo: int = ALGO_AUTO, side_wgrad: bool = False, backward_dtype=None):
        self.cfg = dict(cfg)
        self.cfg.setdefault("multimodal_norm", True)
        self.cfg.setdefault("gated_linear", False)
        self.cfg.setdefault("positional_encoding_type", "sin_cos")
        self.cfg.setdefault("max_position_embeddings", 1024)
        self.cfg.setdefault("dropout", 0.1)
        self.dc = data_config
        self.tm = target_modality
        self.V = int(vocab_out)
        self.dev = torch.device(device)
        self.cd = compute_dtype
        self.algo = algo
        self.d = int(cfg["d_model"])
        self.gated = bool(self.cfg["gated_linear"])
        self.norm = bool(self.cfg["multimodal_norm"])      # False: no per-modality LayerNorm (modeling/utils.py:165-168)
        self.align = align_dict(self.cfg.get("align_config"))
        self.cfg["align_config"] = self.align
        self.x3 = compute_dtype == BF16X3
        self.lowp = compute_dtype != torch.float32          # operands are not plain fp32: MFMA kernels, transposed weight copies
        # "bf16x3 forward / bf16 backward": the forward (what the parity bar is stated on: logits, token ids) runs on split
        # pairs, the backward on the single-pass bf16 kernels reading the HI planes of the saved pair tensors and of the pair
        # weight copies in place (a hi plane is an ordinary bf16 view) -- gradients at the precision class of the reference's
        # own 16-bit mixed training (trainer/trainer.py:69), two thirds of the step's products at one MFMA pass instead of three
        self.bd = backward_dtype if backward_dtype is not None else compute_dtype
        self.mixed = self.x3 and self.bd == torch.bfloat16
        if self.bd != compute_dtype and not self.mixed:
            raise ValueError("backward_dtype: only bfloat16 under the bf16x3 forward is supported")
        # attention-probability dropout: forward stores 1 keep bit per score, backward reads it (AFM_ATTN_KEEP_BITS=0: re-hash)
        self.keep_bits = os.environ.get("AFM_ATTN_KEEP_BITS", "1") != "0"
        self.branch_dtype = torch.float32 if self.x3 else None   # residual branches: fp32 out of the x3 GEMMs (same bytes as a pair)
        self.single16 = compute_dtype in (torch.bfloat16, torch.float16)      # one 16-bit MFMA pass per product
        self.ps = ParamStore(build_specs(self.cfg, data_config, self.V), self.dev,
                             with_bf16=self.single16, with_x2=self.x3, lowp_dtype=compute_dtype if self.single16 else torch.bfloat16)
        # fp16: gradients leave the loss S times too large (S = scaler[0], device-resident) and the optimiser divides it out;
        # state = {S, growth tracker, steps taken, steps skipped} (include/afm_hip.h, afm_adam_step)
        self.scaler = None
        if compute_dtype == torch.float16:
            self.scaler = torch.tensor([LOSS_SCALE_INIT, 0.0, 0.0, 0.0], dtype=torch.float32, device=self.dev)
        self.ps.init_(seed)
        if self.cfg["positional_encoding_type"] == "sin_cos":
            self.pos_enc = sincos_table(self.d, self.cfg["max_position_embeddings"]).to(self.dev)
        else:
            self.pos_enc = None
        self.wt: Dict[str, torch.Tensor] = {}  # transposed bf16 weights for dgrad
        self.w_glu: Dict[str, Any] = {}        # gated FFN: [W1 ; Wg] with rows interleaved in fours (fused GLU epilogues), and
        self.wt_glu: Dict[str, Any] = {}       # its transpose (d x 2f, columns interleaved) for the data gradient
        self.wt_kv_all = None                  # see _refresh_kv_concat
        self._cast_batch = None                # ops.CastBatch of refresh_transposes (built on first use; the buffers it points at never move)
        self._graph_states: Dict[Any, dict] = {}   # decode_init_graphed
        self.training = True
        self.dropout_seed = int(seed)
        self.micro_step = 0
        self._site_ids: Dict[str, int] = {}
        self.grad_ready_hook = None  # callable(flat_offset): grads at >= offset are final (DDP overlap)
        # weight-gradient GEMMs are off the backward critical path (their outputs are only read by the
        # optimiser / all-reduce): they run on a side HIP stream and overlap the LDS-free kernels
        # (LayerNorm backward, casts) of the main stream on the same CUs
        # (round 3: nothing gained over the whole backward -- a 240-workgroup launch with 128 KB of LDS leaves no room for a neighbour.
        # AFM_SIDE_WGRAD=dec: only the DECODER layers' weight gradients, whose critical path is small kernels that do not fill the chip)
        sw = "all" if side_wgrad else os.environ.get("AFM_SIDE_WGRAD", "")
        self.side_wgrad_roles = ("dec", "enc", None) if sw == "all" else (("dec",) if sw == "dec" else ())
        self.wgrad_stream = torch.cuda.Stream(device=self.dev) if (self.dev.type == "cuda" and self.side_wgrad_roles) else None
        self.group_wgrad = os.environ.get("AFM_GROUP_WGRAD", "1") != "0"    # (0: one launch per weight gradient, for A/B timing)
        # AFM_WGRAD_FLUSH_FFN=1: a layer's two FFN weight gradients go out as their own group right behind the FFN's data gradients (their
        # operands still near the memory-side cache) instead of with the attention block's at the end of the layer (A/B: DESIGN 4.0r5 item 10)
        self.flush_ffn_wgrads = os.environ.get("AFM_WGRAD_FLUSH_FFN", "0") == "1"
        self.wgrad_layers = max(1, int(os.environ.get("AFM_WGRAD_LAYERS", "1")))
        self._wg_layers_pending = 0
        self.row_skip = os.environ.get("AFM_ROW_SKIP", "1") != "0"          # (0: the backward computes padded rows like any other)
        # layer options of configs/model/*.yaml beside the defaults: the reference's `post_layer_normalisation` IS torch's norm_first
        # (custom_modeling.py:129,176: True = pre-LN, the shipped setting); `activation_function` goes to the torch layers as is
        self.pre_ln = bool(cfg.get("post_layer_normalisation", True))
        self.act = str(cfg.get("activation_function", "gelu"))
        if self.act not in ("gelu", "relu"):
            raise NotImplementedError(f"activation_function {self.act!r}: 'gelu' and 'relu' are built")
        # AFM_BITS_AHEAD=1: keep-bit tensors filled ahead of the attention forward on a side stream (_bits_ahead).  Built, verified
        # bit-identical and measured in round 3: the forward drops from 0.52 to 0.44 ms but the fill takes 0.195 ms and does not
        # hide under the LayerNorm (whose grid already holds every wave slot): step -1.1 %.  Off by default.
        # kernel-selection bits OR-ed into afm_attn_shape.reserved of every attention backward call (include/afm_hip.h: A / B runs; the
        # keep-bit / re-hash equivalence test pins both paths to the same MFMA shape with it)
        self.attn_bwd_flags = int(os.environ.get("AFM_ATTN_BWD_FLAGS", "0"), 0)
        # the decoder's cross-attention backward as ONE kernel (csrc/afm_attn_fsq_impl.h; afm_attn_shape.reserved bit 18) where its
        # conditions hold (<= 128 decoder positions, keep-bit dropout or none); the library falls back to the two kernels elsewhere
        # (round 6, same box, alternating runs: c2 3 098 -> 3 139 samples/s, c3 5 356 -> 5 419, c4 1 158 -> 1 163; AFM_XATTN_FUSED=0: the two kernels)
        self.xattn_fused = os.environ.get("AFM_XATTN_FUSED", "1") == "1"
        self.bits_stream = (torch.cuda.Stream(device=self.dev)
                            if self.dev.type == "cuda" and os.environ.get("AFM_BITS_AHEAD", "0") == "1" else None)
        self._wg_pending = []
        # Padded positions out of the FORWARD pass of a training step (round 6; include/afm_hip.h ABI 6): the live encoder positions of every
        # sample are moved to the front of its S-row slot at the embedder (afm_compact_plan; AFM_FWD_COMPACT=0: flags only, rows stay
        # where the collator put them) and the encoder-row kernels of the forward -- LayerNorm, the NT GEMMs with their fused
        # epilogues, the self-attention's query blocks -- leave 256-row groups of nothing but padding uncomputed (zeros written).
        # Only with backward pending: eval / generate return the reference's encoder_hidden_states rows.  AFM_FWD_ROW_SKIP=0: off.
        # AFM_FWD_COMPACT: 0 flags only, 1 live positions to the front of every sample's own S rows, 2 (default where the single-pass
        # attention kernels run: 64-wide heads, no alignment head) the whole batch PACKED -- slots of ceil32(live) rows one behind the
        # other, so the 256-row tiles of the GEMMs straddle samples and only the batch's last tile is partly empty (c3: 58 % -> 52 % of
        # the B*S rows computed, c4: 69 % -> 63 %); the attention kernels address the slots through afm_attn_shape.q_off / k_off.
        # AFM_FWD_ROW_SKIP: 1 on, 0 off, "auto" (default): on, but PROBED -- every 64th planned step (the first included) reads back how many
        # rows the plan put in use (one int, one host synchronisation per 64 micro-batches); where that is more than 15/16 of B * S -- c2: the
        # IR patches carry no padding, not one 256-row group is dead -- the next steps of that shape run without plan, hints and tile lists,
        # which cost 0.6 % of the c2 step and had nothing to leave out.
        _fs = os.environ.get("AFM_FWD_ROW_SKIP", "auto")
        self.fwd_skip = _fs != "0"
        self.fwd_skip_auto = _fs == "auto"
        self._skip_probe: Dict[Any, list] = {}      # (B, S) -> [planned calls so far, skip is worth it]
        self.fwd_compact = int(os.environ.get("AFM_FWD_COMPACT", "2"))
        self._enc_off = None  # packed rows: the encoder rows' offsets (B + 1 int32) for the attention shapes of this step
        # AFM_FWD_ARENA (default 1, packed rows only): the encoder-row activations of a training step's forward live in PERSISTENT buffers
        # (one per layer and tensor, zero-filled when first allocated, written by nothing but these kernels).  The dead tail of every such
        # tensor then already holds finite values -- zeros, or rows an earlier step computed -- so the forward kernels write NOTHING there
        # (RowFlags.nofill) instead of zeros: the LayerNorm forward alone spent 40 % of its c3 time filling rows nobody reads.  The
        # backward keeps its zero fills: ITS dead rows must be exact zeros for any consumer that runs without the hints.
        self.fwd_arena = os.environ.get("AFM_FWD_ARENA", "1") != "0"
        self._arena: Dict[str, torch.Tensor] = {}
        # AFM_BWD_NOFILL (default 1, packed rows only): the BACKWARD's hinted kernels stop zero-filling dead rows too -- but only after a
        # step of the same (B, S, T) shape ran WITH the fills and every hinted call reported that its kernels took the hint
        # (ops.hint_log / afm_last_hint): then nothing ever loads those rows.  From there on an ignored hint raises.  The gradient of the
        # stream that reaches the embedder's backward (which reads every position) keeps its zeros.
        # AFM_XATTN_SIDE=1 (A / B probe of round 6, OFF by default): the decoder's cross-attention dK/dV kernel -- 128 queries against 1 024
        # keys: a workgroup's life is its prologue, 2.7 TB/s and 0.13 of the MFMA peak, 3 ... 5 % of a step -- on a SIDE stream.  Nothing of the
        # layer needs dK | dV before the weight-gradient group at the layer's end, so the kernel could overlap the chain of small launches
        # behind it.  Measured (tools/r6_ab.sh, one box, alternating): c2 3 188 vs 3 195, c3 5 673 vs 5 615, c4 1 210 vs 1 207 samples/s
        # (off vs on): nothing, as for the side-stream weight gradients of round 3 -- two queues do not fill each other's gaps here.
        self.xattn_stream = (torch.cuda.Stream(device=self.dev)
                             if self.dev.type == "cuda" and os.environ.get("AFM_XATTN_SIDE", "0") == "1" else None)
        self._xattn_pending = False
        self.bwd_nofill = os.environ.get("AFM_BWD_NOFILL", "1") != "0"
        self.debug_poison = os.environ.get("AFM_DEBUG_POISON", "0") == "1"      # (tests) backward tensors start as NaN
        self._bwd_verified: Dict[Any, bool] = {}      # (B, S, T) -> every hint of a filled backward was honoured
        self._verify_key = None
        # AFM_HEAD_X3 (default 1, single-pass 16-bit modes): token_ff's FORWARD on split bf16 pairs (three MFMA passes, fp32-grade) from the
        # final decoder LayerNorm's fp32 output.  The head is 0.02 % of the step's FLOPs and what the parity bar is stated on: measured
        # (tools/experiments/head_precision.py, fresh init, B = 2) an exact head takes the fp16 mode's logits error from 6.4e-4 to 5.7e-4 at c2,
        # 5.3e-4 to 5.0e-4 at c3, 7.4e-4 to 5.9e-4 at c4.  The backward is unchanged (fp16 operands, as every other product of the mode).
        self.head_x3 = self.single16 and os.environ.get("AFM_HEAD_X3", "1") != "0"
        self._head_w = X2.empty(self.V, self.d, self.dev) if (self.head_x3 and self.dev.type == "cuda") else None
        self._arena_rows = 0  # > 0 while a packed training-step forward is under way: tensors of that many rows come from the arena
        self._fwd_live = {}   # forward of a training step: role -> uint8 per 64-row block, 0 = its whole 256-row group is padding (encode)
        self._frole = None    # whose rows the forward is working on (set by encode around the encoder stack)
        self._live = {}       # backward only: role ("enc" / "dec") -> uint8 per 64-row block, 0 = nothing but padded positions (_backward)
        self._role = None     # whose rows the backward is working on: set by _backward around the layer stacks (None: no hints)
        self.refresh_shadows()

    # ------------------------------------------------------------------ parameters
    def _gemm_weight_groups(self):
        """(first key, rows, cols) of every matrix used as a GEMM B operand."""
        d, out = self.d, []
        for side, n, f in (("encoder", self.cfg["encoder_layers"], self.cfg["encoder_ffn_dim"]),
                           ("decoder", self.cfg["decoder_layers"], self.cfg["decoder_ffn_dim"])):
            for i in range(n):
                p = f"{side}.layers.{i}."
                out.append((p + "self_attn.in_proj_weight", 3 * d, d))
                out.append((p + "self_attn.out_proj.weight", d, d))
                if side == "decoder":
                    out.append((p + "multihead_attn.in_proj_weight", 3 * d, d))
                    out.append((p + "multihead_attn.out_proj.weight", d, d))
                out.append((p + "linear1.weight", (2 if self.gated else 1) * f, d))
                out.append((p + "linear2.weight", d, f))
        out.append(("token_ff.weight", self.V, d))
        return out

    def refresh_shadows(self) -> None:
        """bf16 copies (and transposes, for dgrad) of the GEMM weights; call after any change to
        the fp32 parameters that did not come from `afm_adam_step` (init, load_state_dict)."""
        if not self.lowp:
            return
        for name, rows, cols in self._gemm_weight_groups():
            src = self.ps.span(self.ps.flat, name, rows, cols)
            if name not in self.wt:
                self.wt[name] = ops.empty(cols, rows, self.cd, self.dev)
            if self.x3:
                ops.cast_x2(src, self.ps.span_x2(name, rows, cols), self.wt[name])
            elif self.cd == torch.float16:
                ops.cast_weights(src, self.ps.span(self.ps.bf16, name, rows, cols), self.wt[name])
            else:
                ops.cast_bf16(src, self.ps.span(self.ps.bf16, name, rows, cols), self.wt[name])
        self._refresh_glu()
        self._refresh_kv_concat()
        self._refresh_head()

    def _refresh_head(self) -> None:
        """Split-pair copy of token_ff.weight for the x3 head of the single-pass modes (`head_x3`)."""
        if self._head_w is not None:
            ops.cast_x2(self.ps.span(self.ps.flat, "token_ff.weight", self.V, self.d), self._head_w, None)

    def refresh_transposes(self) -> None:
        """After an optimiser step (which already wrote the flat bf16 shadow)."""
        if not self.lowp:
            return
        if self.x3:     # the Adam kernel writes no split-pair shadow: both copies are made here
            return self.refresh_shadows()
        if self.dev.type == "cuda" and os.environ.get("AFM_CAST_BATCH", "1") != "0":
            # one launch for every transpose and gated-FFN shadow (afm_cast_weights_batch; AFM_CAST_BATCH=0: one launch per matrix, for A/B runs)
            if self._cast_batch is None:
                entries = [(self.ps.span(self.ps.flat, name, rows, cols), None, self.wt[name], 0) for name, rows, cols in self._gemm_weight_groups()]
                if self.gated:
                    for name, rows, cols in self._gemm_weight_groups():
                        if name.endswith("linear1.weight"):
                            if name not in self.w_glu:
                                self.w_glu[name] = ops.empty(rows, cols, self.cd, self.dev)
                                self.wt_glu[name] = ops.empty(cols, rows, self.cd, self.dev)
                            entries.append((self.ps.span(self.ps.flat, name, rows, cols), self.w_glu[name], self.wt_glu[name], rows // 2))
                self._cast_batch = ops.CastBatch(entries, self.cd)
            self._cast_batch.run()
            self._refresh_kv_concat()
            self._refresh_head()
            return
        for name, rows, cols in self._gemm_weight_groups():
            if self.cd == torch.float16:
                ops.cast_weights(self.ps.span(self.ps.flat, name, rows, cols), None, self.wt[name])
            else:
                ops.cast_bf16(self.ps.span(self.ps.flat, name, rows, cols), None, self.wt[name])
        self._refresh_glu()
        self._refresh_kv_concat()
        self._refresh_head()

    def _refresh_glu(self) -> None:
        """Interleaved shadows of the gated up-projections [linear1 ; gate] (include/afm_hip.h, AFM_ACT_GLU*)."""
        if not (self.gated and self.lowp):
            return
        for name, rows, cols in self._gemm_weight_groups():
            if not name.endswith("linear1.weight"):
                continue
            if name not in self.w_glu:
                self.w_glu[name] = ops.empty(rows, cols, self.cd, self.dev)
                self.wt_glu[name] = ops.empty(cols, rows, self.cd, self.dev)
            ops.cast_weights(self.ps.span(self.ps.flat, name, rows, cols), self.w_glu[name], self.wt_glu[name], glu_rows=rows // 2)

    def _glu_fusable(self, rows: int, f: int) -> bool:
        """Whole 256 x 128 tiles of the (rows x 2f) up-projection and MFMA-sized K: the fused kernels' domain."""
        from .lib import ALGO_GENERIC
        return (self.gated and self.lowp and self.algo != ALGO_GENERIC and rows % 256 == 0 and (2 * f) % 128 == 0 and f % 128 == 0
                and self.d % 64 == 0 and rows * 2 * f <= 0xFFFFFFFF)

    def _refresh_kv_concat(self) -> None:
        """(d x Ld*2d) bf16: the transposed cross-attention K/V projection weights of every decoder layer side
        by side, so the gradient w.r.t. the encoder output is ONE GEMM with K = Ld*2d over the concatenated
        dK|dV of all layers instead of Ld fp32 read-modify-write passes over the (B*S x d) accumulator."""
        Ld, d = self.cfg["decoder_layers"], self.d
        if self.wt_kv_all is None:
            self.wt_kv_all = ops.empty(d, Ld * 2 * d, self.cd, self.dev)
        for i in range(Ld):
            src = self.wt[f"decoder.layers.{i}.multihead_attn.in_proj_weight"][:, d:3 * d]
            dst = self.wt_kv_all[:, i * 2 * d:(i + 1) * 2 * d]
            if self.x3:
                dst.hi.copy_(src.hi); dst.lo.copy_(src.lo)
            else:
                dst.copy_(src)

    def W(self, name, rows, cols, r0=0, r1=None):
        """GEMM weight rows [r0:r1) of the (rows x cols) group at `name`, compute dtype."""
        if self.x3:
            return self.ps.span_x2(name, rows, cols)[r0:r1]
        buf = self.ps.bf16 if self.single16 else self.ps.flat
        return self.ps.span(buf, name, rows, cols)[r0:r1]

    def G(self, name, rows, cols, r0=0, r1=None):
        return self.ps.span(self.ps.grad, name, rows, cols)[r0:r1]

    def state_dict(self):
        sd = self.ps.state_dict()
        if self.pos_enc is not None:
            sd["embedding.positional_encodings.pos_enc"] = self.pos_enc.clone()
        for k in [k for k in sd if k.startswith("embedding.")]:
            sd["decoder." + k] = sd[k]  # reference alias (custom_modeling.py:268)
        return sd

    def load_state_dict(self, sd, strict=True):
        self.ps.load(sd, strict)
        if self.pos_enc is not None and "embedding.positional_encodings.pos_enc" in sd:
            self.pos_enc.copy_(sd["embedding.positional_encodings.pos_enc"].to(self.dev))
        self.refresh_shadows()

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    # ------------------------------------------------------------------ helpers
    def _drop(self, site: str):
        p = float(self.cfg["dropout"]) if self.training else 0.0
        if p <= 0.0:
            return ops.NO_DROP
        sid = self._site_ids.setdefault(site, len(self._site_ids) + 1)
        return ops.drop(p, self.dropout_seed + 7919 * self.micro_step, sid)

    def _empty(self, rows, cols, dtype=None):
        return ops.empty(rows, cols, dtype or self.cd, self.dev)

    def _fbuf(self, key: str, rows, cols=None, dtype=None):
        """An encoder-row activation of the forward pass: from the persistent arena while a packed training step is under way (see
        `fwd_arena`), else a fresh tensor.  cols = None: a vector of `rows` fp32 (LayerNorm statistics, lse)."""
        dt = torch.float32 if cols is None else (dtype or self.cd)
        if not (self._arena_rows and key is not None):
            return torch.empty(rows, dtype=dt, device=self.dev) if cols is None else ops.empty(rows, cols, dt, self.dev)
        shape = (rows,) if cols is None else (rows, cols)
        t = self._arena.get(key)
        if t is None or tuple(t.shape) != shape or t.dtype != dt:
            t = torch.zeros(shape, dtype=dt, device=self.dev)
            self._arena[key] = t
        return t

    def _empty_b(self, rows, cols, dtype=None, like=None):
        """Backward-pass activation gradient.  `like`: a saved tensor read by the same GEMM epilogue as `pre_act` (which
        shares C's row stride): in mixed mode that is the hi plane of a pair tensor, so C gets the pair's row stride."""
        dt = dtype or self.bd
        if self.mixed and like is not None and isinstance(like, X2) and dt == torch.bfloat16:
            return torch.empty(rows, like.ld, dtype=dt, device=self.dev)[:, :cols]
        t = ops.empty(rows, cols, dt, self.dev)
        if self.debug_poison and torch.is_tensor(t):      # tests: a backward tensor starts as NaN, so a row that is read without having been written shows
            t.fill_(float("nan"))
        return t

    def _empty_dkv_all(self, rows, cols):
        """The all-layers [dK | dV] buffer of the decoder's cross-attention backward (one data-gradient GEMM over K = layers * 2d at the
        end).  Its rows are 12 KB (c2) ... 36 KB (c4) of 16-bit values, and each layer's dK/dV kernel writes 128-byte pieces of them:
        pieces of consecutive keys a multiple of 4 KB apart run into a channel pattern of the memory system -- 227 us per launch against
        189 us with the rows 128 bytes longer (tools/experiments/attn_cross_layout.py; forward and dQ, which only READ such rows, do
        not care; GEMM operands and the packed Q | K | V of the self-attention do not either, tools/experiments/ld_pad.py).  So the row
        stride is padded by 64 elements where the buffer is a plain 16-bit matrix."""
        if self.bd in (torch.float16, torch.bfloat16) and os.environ.get("AFM_DKV_PAD", "1") != "0":
            return torch.empty(rows, cols + 64, dtype=self.bd, device=self.dev)[:, :cols]
        return self._empty_b(rows, cols)

    def _hb(self, t):
        """Operand of a backward kernel: in mixed mode the hi plane of a pair tensor."""
        return t.hi if (self.mixed and isinstance(t, X2)) else t

    def _packable(self, S: int, T: Optional[int]) -> bool:
        """Packed rows need the single-pass MFMA attention kernels on both the encoder's self-attention and the decoder's
        cross-attention (head size 64, whole 128-row blocks, dropout through the keep-bit tensor) and a consumer of the memory that
        takes offsets: not the alignment head's masked mean."""
        from .lib import ALGO_GENERIC
        heads_ok = all(self.d // int(self.cfg[k]) == 64 and self.d % int(self.cfg[k]) == 0 for k in ("encoder_attention_heads", "decoder_attention_heads"))
        drops = float(self.cfg["dropout"]) > 0.0 and self.training
        return (self.single16 and heads_ok and S % 128 == 0 and T is not None and T % 64 == 0 and not self.align and self.pre_ln
                and self.algo != ALGO_GENERIC and (self.keep_bits or not drops) and self.bits_stream is None and self.attn_bwd_flags == 0)

    def _fwd_hint(self, t, role=None):
        """Forward-sense padded-row hint for an operand with the role's rows (None outside a training step's forward): one byte per
        64-row block, 0 = the block's whole 256-row group is padding, nobody reads its rows of any activation."""
        role = role or self._frole
        h = self._fwd_live.get(role) if role is not None else None
        if h is not None and h.numel() * 64 != t.shape[0]:
            h = None
        return h

    def _linear(self, x, name, rows, cols, r0=0, r1=None, out=None, out_dtype=None, bias_name=None,
                residual=None, dropout=ops.NO_DROP, act=ACT_NONE, pre_act=None, sg_hi_only=False, role=None, key=None):
        w = self.W(name, rows, cols, r0, r1)
        n = w.shape[0]
        fresh = False
        if out is None:
            arena = key is not None and self._arena_rows == x.shape[0] and (out_dtype or self.cd) == self.cd
            out = self._fbuf(key, x.shape[0], n) if arena else self._empty(x.shape[0], n, out_dtype)
            fresh = not arena
        bias = None
        if bias_name is not None:
            bias = self.ps.vec_span(self.ps.flat, bias_name, r0, r0 + n)
        unread = self._fwd_hint(x, role) if (self.single16 and residual is None) else None
        if fresh and isinstance(unread, ops.RowFlags) and unread.nofill:      # a fresh output tensor: its dead rows get the zero fill
            unread = ops.RowFlags(unread.t, unread.dealt, False, unread.tag)
        return ops.gemm(x, w, out, trans_b=True, bias=bias, residual=residual, dropout=dropout, act=act,
                        pre_act=pre_act, algo=self.algo, sg_hi_only=sg_hi_only, rows_unread=unread)

    def _dgrad(self, dy, name, rows, cols, r0=0, r1=None, out=None, out_dtype=None, accumulate=False,
               act=ACT_NONE, pre_act=None, dropout=ops.NO_DROP, role=None):
        """dx = dy @ W[r0:r1]  (W rows = output features)."""
        r1 = rows if r1 is None else r1
        if out is None:
            out = self._empty_b(dy.shape[0], cols, out_dtype)
        kw = dict(accumulate=accumulate, algo=self.algo, act=act, pre_act=self._hb(pre_act), dropout=dropout)
        if self.lowp and torch.is_tensor(dy):
            kw["k_live"] = self._live_hint(dy, role)      # row tiles of nothing but padded positions: zeros in, zeros out
        if self.lowp:
            wt = self._hb(self.wt[name][:, r0:r1])  # (cols, n): NT form for the MFMA kernel
            return ops.gemm(dy, wt, out, trans_b=True, **kw)
        w = self.W(name, rows, cols, r0, r1)
        return ops.gemm(dy, w, out, trans_b=False, **kw)

    def _wgrad(self, dy, x, name, rows, cols, r0=0, r1=None, bias_name=None, role=None):
        """dW[r0:r1] += dy^T x ; db[r0:r1] += colsum(dy)."""
        gw = self.G(name, rows, cols, r0, r1)
        gb = None
        if bias_name is not None:
            gb = self.ps.vec_span(self.ps.grad, bias_name, r0, r0 + gw.shape[0])
        self._wgrad_raw(dy, self._hb(x), gw, gb, role=role)

    def _wgrad_raw(self, dy, x, gw, gb, glu_rows=0, role=None):
        kw = dict(trans_a=True, trans_b=False, accumulate=True, algo=self.algo, a_colsum=gb, glu_rows=glu_rows)
        if torch.is_tensor(dy) and dy.dtype != torch.float32 and gw.shape[0] >= 256 and gw.shape[1] >= 256:
            # padded 64-token blocks carry exact zeros: left out of the token axis (matrices the list-keeping kernels take: the LM head's
            # 128-row gradient runs on a kernel without lists, where a hint would only count as ignored)
            kw["k_live"] = self._live_hint(dy, role)
        if self.group_wgrad and torch.is_tensor(dy) and torch.is_tensor(x) and dy.dtype == x.dtype and dy.dtype != torch.float32:
            # 16-bit operands: the layer's weight gradients go out together at the end of its backward (afm_gemm_group: one
            # launch, one split-K budget); the list keeps dy / x alive until then
            self._wg_pending.append((ops.gemm_desc(dy, x, gw, **kw), dy, x))
            return
        if self.wgrad_stream is None or self._role not in self.side_wgrad_roles:
            ops.gemm(dy, x, gw, **kw)
            return
        side = self.wgrad_stream
        side.wait_stream(torch.cuda.current_stream())          # dy / x are produced on the main stream
        with torch.cuda.stream(side):
            ops.gemm(dy, x, gw, **kw)
        dy.record_stream(side); x.record_stream(side)          # keep their memory until the side stream is done

    def _wgrad_flush(self) -> None:
        """Launch the weight gradients collected since the last flush (afm_gemm_group)."""
        if self._xattn_pending:      # the cross-attention dK | dV of this layer (side stream) are operands of the group below
            torch.cuda.current_stream().wait_stream(self.xattn_stream)
            self._xattn_pending = False
        pending, self._wg_pending = self._wg_pending, []
        if not pending:
            return
        descs = [p[0] for p in pending]
        if self.wgrad_stream is None or self._role not in self.side_wgrad_roles:
            ops.gemm_group(descs)
            return
        side = self.wgrad_stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.gemm_group(descs)
        for _, dy, x in pending:
            dy.record_stream(side); x.record_stream(side)

    # ------------------------------------------------------------------ embedding
    def _pos_rows(self, S: int, saved: Optional[dict]):
        """(S x d) fp32 positional rows added after the per-modality LayerNorm."""
        if S > self.cfg["max_position_embeddings"]:
            raise ValueError(f"sequence length {S} exceeds max_position_embeddings")
        if self.pos_enc is not None:
            return self.pos_enc[:S]
        tab = self.ps.p("embedding.positional_encodings.pos_encodings.weight")[:S]
        pe = torch.empty(S, self.d, dtype=torch.float32, device=self.dev)
        mean = torch.empty(S, dtype=torch.float32, device=self.dev)
        rstd = torch.empty(S, dtype=torch.float32, device=self.dev)
        ops.layernorm_fwd(tab, self.ps.p("embedding.positional_encodings.norm.weight"),
                          self.ps.p("embedding.positional_encodings.norm.bias"), pe, mean, rstd)
        if saved is not None:
            saved["pe_stats"] = (mean, rstd)
        return pe

    def embed_fwd(self, inputs: Dict[str, Any], saved: Optional[dict], row_map: Optional[torch.Tensor] = None) -> torch.Tensor:
        """MultimodalEmbedding.forward (modeling/utils.py:142-182) -> (B*S, d) fp32.  `row_map` (B*S int32, ops.compact_plan): position p of
        sample b is written to row b*S + row_map[b*S + p] -- its positional row stays pos[p]."""
        lens = []
        for m, x in inputs.items():
            t = x["tokenized_input"] if isinstance(x, dict) else x
            lens.append(int(t.shape[1]))
            B = int(t.shape[0])
        S = sum(lens)
        x_out = torch.empty(B * S, self.d, dtype=torch.float32, device=self.dev)
        x_dst = x_out
        if row_map is not None and not self.norm:      # no LayerNorm to move the rows on the way: placed as collated, then permuted
            x_dst = torch.empty_like(x_out)
        pe = self._pos_rows(S, saved)
        off = 0
        mods = []
        for (m, x), Sm in zip(inputs.items(), lens):
            mc = self.dc[m]
            rec = {"name": m, "S": Sm, "off": off}
            p = f"embedding.embedding_layer_dict.{m}."
            if mc["type"] in TEXT_TYPES:
                scale = None
                if isinstance(x, dict):  # xVal (utils.py:154-160)
                    scale = x["numerical_values"].to(torch.float32).contiguous().view(-1)
                    x = x["tokenized_input"]
                ids = x.contiguous().view(-1)
                e = torch.empty(B * Sm, self.d, dtype=torch.float32, device=self.dev)
                ops.gather_rows(ids, self.ps.p(p + "weight"), e, scale)
                rec.update(ids=ids, scale=scale, pad=int(mc.get("pad_token_id", -1)))
            elif mc["type"] in PATCH_TYPES:
                h = x.to(torch.float32).contiguous().view(B * Sm, -1)
                layers = patch_layers(mc, self.d)
                acts = [h]
                for li, (suf, i, o) in enumerate(layers):
                    out = torch.empty(B * Sm, o, dtype=torch.float32, device=self.dev)
                    last = li == len(layers) - 1
                    ops.gemm(h, self.ps.p(p + suf + "weight"), out, trans_b=True,
                             bias=self.ps.p(p + suf + "bias"), act=ACT_NONE if last else ACT_RELU)
                    h = out
                    acts.append(h)
                e = h
                rec.update(acts=acts, layers=layers)
            else:
                raise NotImplementedError(mc["type"])
            if self.norm:
                mean = torch.empty(B * Sm, dtype=torch.float32, device=self.dev)
                rstd = torch.empty(B * Sm, dtype=torch.float32, device=self.dev)
                ops.layernorm_fwd(e, self.ps.p(f"embedding.embedding_norm_dict.{m}.weight"),
                                  self.ps.p(f"embedding.embedding_norm_dict.{m}.bias"), x_out, mean, rstd,
                                  pos=pe, seg_len=Sm, out_seg_stride=S, out_off=off, row_map=row_map)
                rec.update(e=e, mean=mean, rstd=rstd)
            else:
                ops.place_rows(e, x_dst, pos=pe, seg_len=Sm, out_seg_stride=S, out_off=off)
            mods.append(rec)
            off += Sm
        if x_dst is not x_out:
            ops.permute_rows(x_dst, x_out, row_map, B, S)
        if saved is not None:
            saved.update(mods=mods, B=B, S=S, row_map=row_map)
        return x_out

    def embed_bwd(self, dx: torch.Tensor, saved: dict) -> None:
        B, S, d = saved["B"], saved["S"], self.d
        row_map = saved.get("row_map")
        if row_map is not None and (self.pos_enc is None or not self.norm):
            # the batch sum of the learned positional table and the un-normed placement read dx position by position: back to the
            # collated order first (one pass); the per-modality LayerNorm backward otherwise reads dx through the map itself
            dx = ops.permute_rows(dx, torch.empty_like(dx), row_map, B, S, gather=True)
            row_map = None
        if self.pos_enc is None:  # learned table: d pe = sum over the batch, through its LayerNorm
            dpe = torch.empty(S, d, dtype=torch.float32, device=self.dev)
            ops.batch_sum(dx, dpe, B, S, d, accumulate=False)
            mean, rstd = saved["pe_stats"]
            name = "embedding.positional_encodings.pos_encodings.weight"
            tab = self.ps.p(name)[:S]
            dtab = torch.empty(S, d, dtype=torch.float32, device=self.dev)
            ws = torch.empty(ops.layernorm_bwd_ws(S, d), dtype=torch.float32, device=self.dev)
            ops.layernorm_bwd(dpe, tab, self.ps.p("embedding.positional_encodings.norm.weight"), mean, rstd,
                              dtab, self.ps.g("embedding.positional_encodings.norm.weight"),
                              self.ps.g("embedding.positional_encodings.norm.bias"), ws)
            ops.add_inplace(self.ps.g(name)[:S], dtab)
        for rec in saved["mods"]:
            m, Sm, off = rec["name"], rec["S"], rec["off"]
            rows = B * Sm
            de = torch.empty(rows, d, dtype=torch.float32, device=self.dev)
            if self.norm:
                ws = torch.empty(ops.layernorm_bwd_ws(rows, d), dtype=torch.float32, device=self.dev)
                ops.layernorm_bwd(dx, rec["e"], self.ps.p(f"embedding.embedding_norm_dict.{m}.weight"),
                                  rec["mean"], rec["rstd"], de,
                                  self.ps.g(f"embedding.embedding_norm_dict.{m}.weight"),
                                  self.ps.g(f"embedding.embedding_norm_dict.{m}.bias"), ws,
                                  seg_len=Sm, out_seg_stride=S, out_off=off, row_map=row_map)
            else:
                ops.place_rows(dx, de, seg_len=Sm, out_seg_stride=S, out_off=off, gather=True)
            p = f"embedding.embedding_layer_dict.{m}."
            if "ids" in rec:
                ops.scatter_add_rows(rec["ids"], de, self.ps.g(p + "weight"), rec["scale"], rec["pad"])
            else:
                g = de
                layers, acts = rec["layers"], rec["acts"]
                for li in range(len(layers) - 1, -1, -1):
                    suf = layers[li][0]
                    if li < len(layers) - 1:  # ReLU backward on the hidden activation
                        g = ops.relu_bwd(g, acts[li + 1])
                    ops.gemm(g, acts[li], self.ps.g(p + suf + "weight"), trans_a=True, trans_b=False,
                             accumulate=True)
                    ops.colsum(g, self.ps.g(p + suf + "bias"), accumulate=True)
                    if li > 0:
                        gi = torch.empty(rows, layers[li][1], dtype=torch.float32, device=self.dev)
                        ops.gemm(g, self.ps.p(p + suf + "weight"), gi, trans_b=False)
                        g = gi

    # ------------------------------------------------------------------ blocks
    def _ln_fwd(self, x, prefix, saved, key, out_dtype=None, pend=None, arena=True):
        """y = LN(x + pend).  `pend` is the previous block's (dropped-out) branch output: the residual
        add is fused here, the summed stream is materialised once (fp32) and returned.  arena=False: fresh output tensors even in a
        packed training step (their dead rows then get the zero fill): for outputs that leave the engine."""
        rows = x.shape[0]
        ak = prefix if (arena and self._arena_rows == rows and out_dtype is None and saved is not None) else None      # arena key (packed training step)
        hint = self._fwd_hint(x)
        if ak is None and isinstance(hint, ops.RowFlags) and hint.nofill:
            hint = ops.RowFlags(hint.t, hint.dealt, False, hint.tag)
        y = self._fbuf(ak and ak + "y", rows, self.d, out_dtype)
        mean = self._fbuf(ak and ak + "mean", rows)
        rstd = self._fbuf(ak and ak + "rstd", rows)
        if pend is not None:
            # `pend` = (branch, its dropout stream): the GEMM that produced the branch wrote it plain, the
            # dropout is applied here on the way into the stream (this kernel is HBM-bound, the hash is free)
            br, br_drop = pend if isinstance(pend, tuple) else (pend, ops.NO_DROP)
            xs = self._fbuf(ak and ak + "xs", rows, self.d, torch.float32)   # x itself is the saved input of an earlier LayerNorm: keep it
            ops.layernorm_fwd(x, self.ps.p(prefix + "weight"), self.ps.p(prefix + "bias"), y, mean, rstd,
                              add=br, x_sum=xs, add_dropout=br_drop, row_live=hint)
            x = xs
        else:
            ops.layernorm_fwd(x, self.ps.p(prefix + "weight"), self.ps.p(prefix + "bias"), y, mean, rstd, row_live=hint)
        if saved is not None:
            saved[key] = (x, mean, rstd)
        return y, x

    def _ln_bwd(self, dy, prefix, saved, key, dres, next_site=None, fill=False):
        """Returns (dx, dx_dropped): dx is the fp32 stream gradient; dx_dropped (compute dtype) is
        dropout'(dx) for the residual branch that precedes this LayerNorm (dropout site `next_site`),
        produced by the same kernel so dx is not read again."""
        x, mean, rstd = saved[key]
        dx = torch.empty_like(x)
        ws = torch.empty(ops.layernorm_bwd_ws(x.shape[0], self.d), dtype=torch.float32, device=self.dev)
        dxd, dr = None, ops.NO_DROP
        if next_site is not None:
            dxd = self._empty(x.shape[0], self.d, dy.dtype)
            dr = self._drop(next_site)
        if self.debug_poison:
            dx.fill_(float("nan"))
            if torch.is_tensor(dxd):
                dxd.fill_(float("nan"))
        hint = self._live_hint(dy)
        if fill and isinstance(hint, ops.RowFlags) and hint.nofill:      # this dx is read by a kernel without hints: its dead rows stay zeros
            hint = ops.RowFlags(hint.t, hint.dealt, False)
        ops.layernorm_bwd(dy, x, self.ps.p(prefix + "weight"), mean, rstd, dx, self.ps.g(prefix + "weight"),
                          self.ps.g(prefix + "bias"), ws, dres=dres, dx_drop=dxd, dropout=dr, row_live=hint)
        return dx, dxd

    def _bits_ahead(self, B, H, Tq, Tk, site, saved):
        """Keep-bit tensor of an attention call, filled AHEAD of it on a side stream: the bits depend on the dropout stream and the
        shape only, so the hash (pure vector work) runs under the HBM-bound LayerNorm at the head of the block instead of inside
        the forward kernel, which then reads lane masks (afm_attn_drop_bits_fill).  Returns (bits, event) or None."""
        dr = self._drop(site)
        if saved is None or not self.single16 or dr.p <= 0.0 or not self.keep_bits or self.bits_stream is None:
            return None
        bits = torch.empty(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=self.dev)
        shp = ops.attn_shape(B, H, Tq, Tk, self.d // H, self.cd, self.d, self.d, self.d, self.d, None, False, dr, self.algo)
        ops.attn_set_drop_bits(shp, bits)
        side = self.bits_stream
        side.wait_stream(torch.cuda.current_stream())     # the memory's earlier users on the main stream are done
        with torch.cuda.stream(side):
            ok = ops.attn_fill_drop_bits(shp)
            ev = torch.cuda.Event()
            ev.record(side)
        bits.record_stream(side)
        return (bits, ev) if ok else None

    def _attach_drop_bits(self, shp, saved, ahead=None) -> None:
        """With backward pending and dropout on, the forward attention kernel also writes the keep bits of its dropout
        (1 bit per score, B*H*Tq*Tk/8 bytes) and the backward kernels read them instead of re-hashing every score.
        `ahead` = _bits_ahead()'s result: the tensor is already being filled, the forward waits for it and reads it."""
        if ahead is not None:
            torch.cuda.current_stream().wait_event(ahead[1])
            ops.attn_set_drop_bits(shp, ahead[0])
            shp.reserved |= 32
            return
        if saved is not None and self.lowp and shp.drop.p > 0.0 and self.keep_bits:
            n = ops.attn_drop_bits_words(shp.B, shp.H, shp.Tq, shp.Tk)
            ops.attn_set_drop_bits(shp, torch.empty(n, dtype=torch.int64, device=self.dev))

    def _shape_b(self, shp):
        """The forward's attention descriptor for the backward kernels: in mixed mode the same strides (a hi plane keeps the
        pair tensor's row stride), keep-bit tensor and dropout stream with the single-pass dtype."""
        if not self.mixed:
            return shp
        sb = type(shp).from_buffer_copy(shp)
        sb.dtype = AFM_BF16
        for k in ("_bits_keepalive", "_keepalive", "_off_keepalive"):
            if hasattr(shp, k):
                setattr(sb, k, getattr(shp, k))
        return sb

    def _self_attn_fwd(self, x, pend, p, B, T, H, key_pad, causal, saved, site, h=None):
        """x + pend is the incoming stream; returns (stream, this block's branch to be added).  Post-LN layers pass the block's
        input operand `h` themselves (x, pend unused) and normalise AFTER the branch is added (_post_norm)."""
        d = self.d
        ahead = self._bits_ahead(B, H, T, T, site + "attn", saved)
        if h is None:
            h, x = self._ln_fwd(x, p + "norm1.", saved, "ln1", pend=pend)
        qkv = self._linear(h, p + "self_attn.in_proj_weight", 3 * d, d, bias_name=p + "self_attn.in_proj_bias", key=p + "qkv")
        arena = self._arena_rows == B * T and not causal
        a = self._fbuf(p + "a", B * T, d) if arena else self._empty(B * T, d)
        lse = torch.empty(B * H * T, dtype=torch.float32, device=self.dev)
        lq, la = ops._ld(qkv), ops._ld(a)
        off = self._enc_off if (not causal and self._frole == "enc") else None      # packed encoder rows
        shp = ops.attn_shape(B, H, T, T, d // H, self.cd, lq, lq, lq, la, key_pad, causal,
                             self._drop(site + "attn"), self.algo, q_off=off, k_off=off)
        self._attach_drop_bits(shp, saved, ahead)
        if not causal and key_pad is not None and self._fwd_hint(h) is not None:
            shp.reserved |= 64      # encoder, training step: padded query rows are read by nobody (afm_attn_fwd: O = 0, lse = +inf there)
            if arena and off is not None:
                shp.reserved |= 131072      # ... and O is a persistent buffer: the dead tail is left as it is
        ops.attn_fwd(shp, qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], a, lse)
        shp.reserved &= ~(64 | 131072)      # (the backward sets its own sense of the bits)
        br = self._linear(a, p + "self_attn.out_proj.weight", d, d, bias_name=p + "self_attn.out_proj.bias",
                          out_dtype=self.branch_dtype, key=p + "sa_br")
        if saved is not None:
            saved["sa"] = (h, qkv, a, lse, shp)
        return x, (br, self._drop(site + "res"))

    def _self_attn_bwd(self, dx1, dy, p, saved, next_site, acc=None):
        """dx1: fp32 grad of the stream after this block; dy = dropout'(dx1) in the compute dtype.
        Returns (grad of the stream before the block, its dropped copy for `next_site`).  Post-LN: `acc` is the fp32 gradient of
        the block's input so far; the gradient through the block is accumulated into it by the last dgrad (nothing returned)."""
        d = self.d
        h, qkv, a, lse, shp = saved["sa"]
        rows = h.shape[0]
        self._wgrad(dy, a, p + "self_attn.out_proj.weight", d, d, bias_name=p + "self_attn.out_proj.bias")
        # (the attention ABI has ONE row stride for O and dO: in mixed mode dO takes the stride of O's hi plane)
        da = self._dgrad(dy, p + "self_attn.out_proj.weight", d, d, out=self._empty_b(rows, d, like=a))
        dqkv = self._empty_b(rows, 3 * d)
        delta = torch.empty_like(lse)
        ldg = ops._ld(dqkv)
        qkv_b, shp = self._hb(qkv), self._shape_b(shp)
        # encoder: padded positions are masked as keys everywhere and take no part in the loss: their rows of every activation
        # gradient are exact zeros, so the self-attention backward may skip them as queries (include/afm_hip.h, reserved bit 6).
        # (Not the decoder's: whether its padded rows carry a gradient depends on the labels the caller passes.)
        if self.row_skip and not shp.causal:
            shp.reserved |= 64
        hint_e = self._live.get("enc")
        if not shp.causal and shp.q_off and isinstance(hint_e, ops.RowFlags) and hint_e.nofill:
            shp.reserved |= 131072      # packed rows, verified hints: the dead tail of dQ / dK / dV is left unwritten
        shp.reserved |= self.attn_bwd_flags | (262144 if (self.xattn_fused and shp.causal) else 0)      # (decoder self-attention: the fused backward where T <= 128)
        ops.attn_bwd(shp, qkv_b[:, :d], qkv_b[:, d:2 * d], qkv_b[:, 2 * d:], self._hb(a), da, lse, delta,
                     dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], ldg, ldg, ldg)
        self._wgrad(dqkv, h, p + "self_attn.in_proj_weight", 3 * d, d, bias_name=p + "self_attn.in_proj_bias")
        if acc is not None:
            self._dgrad(dqkv, p + "self_attn.in_proj_weight", 3 * d, d, out=acc, accumulate=True)
            return None
        dh = self._dgrad(dqkv, p + "self_attn.in_proj_weight", 3 * d, d)
        # (next_site None: the stack's first layer -- its dx goes to the embedder's backward, which reads every position)
        return self._ln_bwd(dh, p + "norm1.", saved, "ln1", dres=dx1, next_site=next_site, fill=next_site is None)

    def _ffn_fwd(self, x, pend, p, f, norm, saved, site, h=None):
        d, k = self.d, (2 if self.gated else 1)
        if h is None:
            h, x = self._ln_fwd(x, p + norm, saved, "lnf", pend=pend)
        rows = h.shape[0]
        dr = self._drop(site + "ffn")
        arena = self._arena_rows == rows and saved is not None
        g = self._fbuf(p + "ffn_g", rows, f) if arena else self._empty(rows, f)
        if self.act != "gelu":
            # activation "relu": the projection, then act(u) [* v] + dropout in one elementwise kernel (afm_glu_fwd with its
            # activation selector); the fused epilogues below are GELU's
            uv = self._linear(h, p + "linear1.weight", k * f, d, bias_name=p + "linear1.bias")
            ops.glu_fwd(uv[:, :f], uv[:, f:] if self.gated else None, g, dr, act=ACT_RELU)
        elif self.gated and self._glu_fusable(rows, f):
            # gelu(u) * v (+ dropout) in the epilogue of ONE GEMM over the interleaved [W1 ; Wg]; with backward pending the
            # epilogue stores keep*scale*[gelu'(u) v | gelu(u)] instead of u and v, so the data-gradient epilogue of the
            # down-projection is two multiplies and afm_glu_fwd / afm_glu_bwd drop out of the step
            uv = (self._fbuf(p + "ffn_uv", rows, 2 * f) if arena else self._empty(rows, 2 * f)) if saved is not None else None
            ops.gemm(h, self.w_glu[p + "linear1.weight"], g, trans_b=True,
                     bias=self.ps.vec_span(self.ps.flat, p + "linear1.bias", 0, 2 * f),
                     act=ACT_GLU_SAVE if saved is not None else ACT_GLU, pre_act=uv, dropout=dr, algo=self.algo, glu_rows=f,
                     sg_hi_only=self.mixed, rows_unread=self._fwd_hint(h) if self.single16 else None)
            dr = (dr, "glu")
        elif self.gated:
            uv = self._linear(h, p + "linear1.weight", k * f, d, bias_name=p + "linear1.bias")
            ops.glu_fwd(uv[:, :f], uv[:, f:], g, dr)
        else:   # GELU + inner dropout fused into the up-projection's epilogue.  With backward pending the epilogue
            # also stores keep * scale * gelu'(u) (same keep bits), so the dgrad epilogue is one multiply.
            # (whole 256 x 256 tiles only: other shapes keep u and the GELU' epilogue)
            sg = saved is not None and self.lowp and rows % 256 == 0 and f % 256 == 0
            uv = (self._fbuf(p + "ffn_uv", rows, f) if arena else self._empty(rows, f)) if saved is not None else None
            self._linear(h, p + "linear1.weight", f, d, out=g, bias_name=p + "linear1.bias",
                         act=ACT_GELU_SAVE_GRAD if sg else ACT_GELU, pre_act=uv, dropout=dr, sg_hi_only=self.mixed and sg)
            dr = (dr, sg)
        br = self._linear(g, p + "linear2.weight", d, f, bias_name=p + "linear2.bias", out_dtype=self.branch_dtype, key=p + "ffn_br")
        if saved is not None:
            saved["ffn"] = (h, uv, g, dr)
        return x, (br, self._drop(site + "res2"))

    def _ffn_bwd(self, dx1, dy, p, f, norm, saved, next_site, acc=None):
        d, k = self.d, (2 if self.gated else 1)
        h, uv, g, dr = saved["ffn"]
        rows = h.shape[0]
        self._wgrad(dy, g, p + "linear2.weight", d, f, bias_name=p + "linear2.bias")
        duv = self._empty_b(rows, k * f, like=uv)

        def tail(dh_from, weight_t=None):
            """dh = duv W1 (or its interleaved transpose), then the block's LayerNorm backward -- or, post-LN, accumulated."""
            if acc is not None:
                if weight_t is not None:
                    ops.gemm(dh_from, weight_t, acc, trans_b=True, accumulate=True, algo=self.algo)
                else:
                    self._dgrad(dh_from, p + "linear1.weight", k * f, d, out=acc, accumulate=True)
                return None
            if weight_t is not None:
                dh = self._empty_b(rows, d)
                ops.gemm(dh_from, weight_t, dh, trans_b=True, algo=self.algo, k_live=self._live_hint(dh_from))
            else:
                dh = self._dgrad(dh_from, p + "linear1.weight", k * f, d)
            return self._ln_bwd(dh, p + norm, saved, "lnf", dres=dx1, next_site=next_site)

        if self.gated and isinstance(dr, tuple) and dr[1] == "glu":
            # [du | dv] (interleaved) = (dy W2) * saved factors in the dgrad epilogue; weight gradient rows de-interleaved by
            # the wgrad kernel into the reference's [linear1 ; gate] layout; dh through the interleaved transpose
            ops.gemm(dy, self._hb(self.wt[p + "linear2.weight"]), duv, trans_b=True, act=ACT_GLU_BWD, pre_act=self._hb(uv),
                     algo=self.algo, glu_rows=f, k_live=self._live_hint(dy))
            gw = self.G(p + "linear1.weight", 2 * f, d)
            gb = self.ps.vec_span(self.ps.grad, p + "linear1.bias", 0, 2 * f)
            self._wgrad_raw(duv, self._hb(h), gw, gb, glu_rows=f)
            if self.flush_ffn_wgrads:
                self._wgrad_flush()
            return tail(duv, self._hb(self.wt_glu[p + "linear1.weight"]))
        if not isinstance(dr, tuple):     # unfused forms (gated shapes outside the fused kernels' domain; activation "relu")
            dg = self._dgrad(dy, p + "linear2.weight", d, f)
            uv_b = self._hb(uv)
            ops.glu_bwd(uv_b[:, :f], uv_b[:, f:] if self.gated else None, dg, duv[:, :f], duv[:, f:] if self.gated else None, dr,
                        act=ACT_GELU if self.act == "gelu" else ACT_RELU)
        elif dr[1]:   # du = (dy W2) * [keep * scale * gelu'(u)] in the dgrad epilogue: dg never reaches HBM
            self._dgrad(dy, p + "linear2.weight", d, f, out=duv, act=ACT_MUL_SAVED, pre_act=uv)
        else:       # du = dropout'(dy W2) * gelu'(u)
            self._dgrad(dy, p + "linear2.weight", d, f, out=duv, act=ACT_GELU_BWD, pre_act=uv, dropout=dr[0])
        self._wgrad(duv, h, p + "linear1.weight", k * f, d, bias_name=p + "linear1.bias")
        if self.flush_ffn_wgrads:
            self._wgrad_flush()
        return tail(duv)

    def _cross_attn_fwd(self, x, pend, mem, p, B, T, S, H, mem_pad, saved, site, h=None):
        d = self.d
        ahead = self._bits_ahead(B, H, T, S, site + "xattn", saved)
        if h is None:
            h, x = self._ln_fwd(x, p + "norm2.", saved, "ln2", pend=pend)
        w, bname = p + "multihead_attn.in_proj_weight", p + "multihead_attn.in_proj_bias"
        q = self._linear(h, w, 3 * d, d, 0, d, bias_name=bname)
        kv = self._linear(mem, w, 3 * d, d, d, 3 * d, bias_name=bname, role="enc", key=p + "xkv")      # memory-side rows: encoder positions
        a = self._empty(B * T, d)
        lse = torch.empty(B * H * T, dtype=torch.float32, device=self.dev)
        shp = ops.attn_shape(B, H, T, S, d // H, self.cd, ops._ld(q), ops._ld(kv), ops._ld(kv), ops._ld(a), mem_pad, False,
                             self._drop(site + "xattn"), self.algo, k_off=self._enc_off)      # (packed memory rows)
        self._attach_drop_bits(shp, saved, ahead)
        ops.attn_fwd(shp, q, kv[:, :d], kv[:, d:], a, lse)
        br = self._linear(a, p + "multihead_attn.out_proj.weight", d, d,
                          bias_name=p + "multihead_attn.out_proj.bias", out_dtype=self.branch_dtype)
        if saved is not None:
            saved["ca"] = (h, q, kv, a, lse, shp)
        return x, (br, self._drop(site + "xres"))

    def _cross_attn_bwd(self, dx1, dy, mem, dmem, p, saved, next_site, dkv_all=None, layer=0, acc=None):
        d = self.d
        h, q, kv, a, lse, shp = saved["ca"]
        wo, bo = p + "multihead_attn.out_proj.weight", p + "multihead_attn.out_proj.bias"
        self._wgrad(dy, a, wo, d, d, bias_name=bo)
        da = self._dgrad(dy, wo, d, d, out=self._empty_b(h.shape[0], d, like=a))
        dq = self._empty_b(h.shape[0], d)
        if dkv_all is not None:     # this layer's dK | dV columns of the all-layers buffer (one dgrad at the end)
            dkv = dkv_all[:, layer * 2 * d:(layer + 1) * 2 * d]
        else:
            dkv = self._empty_b(mem.shape[0], 2 * d)
        ldkv = ops._ld(dkv)
        delta = torch.empty_like(lse)
        kv_b = self._hb(kv)
        shp = self._shape_b(shp)
        hint_e = self._live.get("enc")
        if shp.k_off and isinstance(hint_e, ops.RowFlags) and hint_e.nofill:
            shp.reserved |= 131072      # packed memory rows, verified hints: the dead tail of dK / dV is left unwritten
        shp.reserved |= self.attn_bwd_flags | (262144 if self.xattn_fused else 0)
        args = (self._hb(q), kv_b[:, :d], kv_b[:, d:], self._hb(a), da, lse, delta, dq, dkv[:, :d], dkv[:, d:], ops._ld(dq), ldkv, ldkv)
        side = self.xattn_stream if (dkv_all is not None and self.single16 and self.group_wgrad and self.wgrad_stream is None) else None
        if side is None:
            ops.attn_bwd(shp, *args)
        else:
            # delta and dQ here; dK | dV on the side stream behind them (they read delta), joined in front of the layer's weight gradients
            def part(bits):
                sp = type(shp).from_buffer_copy(shp)
                sp.reserved = (shp.reserved & ~3) | bits
                for k in ("_bits_keepalive", "_keepalive", "_off_keepalive"):
                    if hasattr(shp, k):
                        setattr(sp, k, getattr(shp, k))
                return sp
            ops.attn_bwd(part(1), *args)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            side.wait_event(ev)
            with torch.cuda.stream(side):
                ops.attn_bwd(part(2), *args)
            for t in (da, delta):      # (locals of this call: their memory must outlive the side kernel; everything else lives in `saved`)
                t.record_stream(side)
            self._xattn_pending = True
        w, bname = p + "multihead_attn.in_proj_weight", p + "multihead_attn.in_proj_bias"
        self._wgrad(dq, h, w, 3 * d, d, 0, d, bias_name=bname)
        self._wgrad(dkv, mem, w, 3 * d, d, d, 3 * d, bias_name=bname, role="enc")      # memory-side rows: encoder positions
        if dkv_all is None:
            self._dgrad(dkv, w, 3 * d, d, d, 3 * d, out=dmem, accumulate=True, role="enc")  # fp32 accumulator
        if acc is not None:
            self._dgrad(dq, w, 3 * d, d, 0, d, out=acc, accumulate=True)
            return None
        dh = self._dgrad(dq, w, 3 * d, d, 0, d)
        return self._ln_bwd(dh, p + "norm2.", saved, "ln2", dres=dx1, next_site=next_site)

    # ------------------------------------------------------------------ post-LN sublayers (post_layer_normalisation=False)
    def _operand(self, x32, backward=False):
        """The fp32 stream as a GEMM operand of the (forward / backward) compute dtype."""
        if not self.lowp:
            return x32
        dst = self._empty_b(x32.shape[0], x32.shape[1]) if backward else self._empty(x32.shape[0], x32.shape[1])
        return ops.convert(x32, dst)

    def _post_norm(self, x, pend, prefix, saved, key):
        """torch's norm_first=False sublayer tail (torch:nn/modules/transformer.py, `x = norm(x + dropout(block(x)))`): the new
        stream in fp32 (one fused add + dropout + LayerNorm launch) and its operand copy for the next block."""
        y, _ = self._ln_fwd(x, prefix, saved, key, out_dtype=torch.float32, pend=pend)
        return y, self._operand(y)

    def _post_sub_bwd(self, dY, prefix, saved, key, site, core):
        """Backward of one post-LN sublayer y = LN(x + dropout(block(x))): dY (fp32) -> d(x + branch) through the LayerNorm,
        its dropped copy drives the block's backward, which accumulates the gradient through the block onto the same buffer."""
        dxs, ddrop = self._ln_bwd(dY, prefix, saved, key, dres=None, next_site=site)
        core(self._operand(ddrop, backward=True), dxs)
        return dxs

    # ------------------------------------------------------------------ whole model
    def encode(self, enc_inputs, attention_mask, saved: Optional[dict] = None, dec_len: Optional[int] = None):
        """embed + CustomEncoder.forward (custom_modeling.py:220-243) -> memory (B*S, d).  `enc_inputs` is the modality dict
        (embedded here, on the engine's schedule) or an already embedded (B, S, d) tensor, as the reference's
        `inputs_embeds` (custom_modeling.py:420-445): forward / generate only, its producer is outside this engine."""
        B, S = attention_mask.shape
        self._fwd_live, self._frole, self._enc_off, self._arena_rows = {}, None, None, 0
        if torch.is_tensor(enc_inputs):
            if saved is not None:
                raise ValueError("a backward pass through externally embedded inputs is not available: pass the modality dict")
            if tuple(enc_inputs.shape) != (B, S, self.d):
                raise ValueError(f"inputs_embeds must be (B, S, d) = {(B, S, self.d)}, got {tuple(enc_inputs.shape)}")
            x = enc_inputs.to(device=self.dev, dtype=torch.float32).reshape(B * S, self.d).contiguous()
            key_pad = (attention_mask == 0).to(torch.uint8).contiguous()
        else:
            key_pad = (attention_mask == 0).to(torch.uint8).contiguous()
            plan = None
            probe = self._skip_probe.setdefault((B, S), [0, True]) if self.fwd_skip_auto else None
            if probe is not None:
                probe[0] += 1
            if saved is not None and self.fwd_skip and self.single16 and S % 256 == 0 and (probe is None or probe[1] or probe[0] % 64 == 1):
                # a training step: live positions to the front of every sample's slot, 256-row groups of nothing but padding left out
                # of the encoder-row kernels below (and of the decoder's memory-side projections)
                mode = self.fwd_compact
                if mode == 2 and not self._packable(S, dec_len):
                    mode = 1
                plan = ops.compact_plan(key_pad, B, S, 256, compact=mode)
                key_pad = plan.pad.view(B, S)
                if probe is not None and probe[0] % 64 == 1:
                    # rows in 256-row groups that hold something: the share of the forward's encoder-row work that remains (one 4-byte
                    # readback per 64 planned steps of a shape; the plan of THIS step is used either way)
                    groups = plan.live_tile.view(-1, 4)[:, 0]
                    probe[1] = float(groups.float().mean()) <= 15.0 / 16.0
                    if not probe[1]:
                        self._arena.clear()
                # packed rows + arena: the forward kernels leave the dead tail unwritten (the unfused FFN forms write through kernels that
                # take no hint: they stay on fresh tensors with the zero fill).  A shape the probe has turned off runs its probing steps on
                # plain tensors with the zero fill: no 16 ... 47 GB of persistent buffers for one step in 64.
                nofill = (plan.packed and self.fwd_arena and self.act == "gelu" and (not self.gated or self._glu_fusable(B * S, int(self.cfg["encoder_ffn_dim"])))
                          and (probe is None or probe[1]))
                self._arena_rows = B * S if nofill else 0
                self._fwd_live["enc"] = ops.RowFlags(plan.live_tile, plan.packed, nofill)
                self._enc_off = plan.seq_off if plan.packed else None
                self._last_plan_mode = plan.mode      # (tests: which layout the step really ran)
                saved["plan"] = plan
            x = self.embed_fwd(enc_inputs, None if saved is None else saved.setdefault("emb_enc", {}),
                               row_map=plan.dest if (plan is not None and plan.compact) else None)
        assert x.shape[0] == B * S, "attention_mask does not match the concatenated modalities"
        self._frole = "enc" if "enc" in self._fwd_live else None
        H = self.cfg["encoder_attention_heads"]
        layers, pend = [], None
        h = None if self.pre_ln else self._operand(x)
        for i in range(self.cfg["encoder_layers"]):
            p = f"encoder.layers.{i}."
            sv = {} if saved is not None else None
            if self.pre_ln:
                x, pend = self._self_attn_fwd(x, pend, p, B, S, H, key_pad, False, sv, f"e{i}")
                x, pend = self._ffn_fwd(x, pend, p, self.cfg["encoder_ffn_dim"], "norm2.", sv, f"e{i}")
            else:   # x = norm1(x + sa(x)); x = norm2(x + ff(x))
                _, br = self._self_attn_fwd(None, None, p, B, S, H, key_pad, False, sv, f"e{i}", h=h)
                x, h = self._post_norm(x, br, p + "norm1.", sv, "ln1")
                _, br = self._ffn_fwd(None, None, p, self.cfg["encoder_ffn_dim"], "norm2.", sv, f"e{i}", h=h)
                x, h = self._post_norm(x, br, p + "norm2.", sv, "lnf")
            layers.append(sv)
        # (the memory leaves the engine as `encoder_hidden_states`: a fresh tensor, dead rows zero-filled, never a view of a persistent buffer)
        mem, _ = self._ln_fwd(x, "encoder.norm.", saved, "enc_norm", pend=pend, arena=False)
        self._frole = None
        if saved is not None:
            saved.update(enc_layers=layers, key_pad=key_pad, B=B, S=S)
        return mem, key_pad

    def decode(self, dec_ids, mem, mem_pad, dec_attention_mask, S, saved: Optional[dict] = None):
        """CustomDecoder.forward + token_ff (custom_modeling.py:271-320,486) -> fp32 logits."""
        B, T = dec_ids.shape
        x = self.embed_fwd({self.tm: dec_ids}, None if saved is None else saved.setdefault("emb_dec", {}))
        tgt_pad = None
        if dec_attention_mask is not None:
            tgt_pad = (dec_attention_mask == 0).to(torch.uint8).contiguous()
        H = self.cfg["decoder_attention_heads"]
        layers, pend = [], None
        h = None if self.pre_ln else self._operand(x)
        for i in range(self.cfg["decoder_layers"]):
            p = f"decoder.layers.{i}."
            sv = {} if saved is not None else None
            if self.pre_ln:
                x, pend = self._self_attn_fwd(x, pend, p, B, T, H, tgt_pad, True, sv, f"d{i}")
                x, pend = self._cross_attn_fwd(x, pend, mem, p, B, T, S, H, mem_pad, sv, f"d{i}")
                x, pend = self._ffn_fwd(x, pend, p, self.cfg["decoder_ffn_dim"], "norm3.", sv, f"d{i}")
            else:
                _, br = self._self_attn_fwd(None, None, p, B, T, H, tgt_pad, True, sv, f"d{i}", h=h)
                x, h = self._post_norm(x, br, p + "norm1.", sv, "ln1")
                _, br = self._cross_attn_fwd(None, None, mem, p, B, T, S, H, mem_pad, sv, f"d{i}", h=h)
                x, h = self._post_norm(x, br, p + "norm2.", sv, "ln2")
                _, br = self._ffn_fwd(None, None, p, self.cfg["decoder_ffn_dim"], "norm3.", sv, f"d{i}", h=h)
                x, h = self._post_norm(x, br, p + "norm3.", sv, "lnf")
            layers.append(sv)
        hf, logits = self._head(x, pend, saved)
        if saved is not None:
            saved.update(dec_layers=layers, hf=hf, T=T, tgt_pad=tgt_pad)
        return logits


    def _head(self, x, pend, saved):
        """Final decoder LayerNorm + token_ff (custom_modeling.py:318,486) -> (hf in the compute dtype for the backward, fp32 logits).
        `head_x3`: the LayerNorm writes fp32, the product runs on split pairs (three MFMA passes); hf is then a cast of that output."""
        if self._head_w is None:
            hf, _ = self._ln_fwd(x, "decoder.norm.", saved, "dec_norm", pend=pend)
            return hf, self._linear(hf, "token_ff.weight", self.V, self.d, out_dtype=torch.float32, bias_name="token_ff.bias")
        hf32, _ = self._ln_fwd(x, "decoder.norm.", saved, "dec_norm", out_dtype=torch.float32, pend=pend)
        rows = hf32.shape[0]
        hx = ops.convert(hf32, X2.empty(rows, self.d, self.dev))
        logits = torch.empty(rows, self.V, dtype=torch.float32, device=self.dev)
        # variant 33: ONE kernel form (the small-tile split-pair kernel) whatever the row count, so a sample's logits do not depend on the
        # size of the batch it is in (tests/test_gpu_model.py: a micro-batch equals its two halves, bit for bit)
        ops.gemm(hx, self._head_w, logits, trans_b=True, bias=self.ps.vec_span(self.ps.flat, "token_ff.bias", 0, self.V), algo=self.algo, variant=33)
        hf = ops.convert(hf32, self._empty(rows, self.d)) if saved is not None else None
        return hf, logits

    # ------------------------------------------------------------------ incremental decode (KV cache)
    def _mem_rows(self, mem, B, S):
        """Encoder memory as a (B*S, d) operand of the compute dtype."""
        mem2 = mem if isinstance(mem, X2) else mem.reshape(B * S, self.d)
        if mem2.dtype != self.cd:
            src = mem2 if isinstance(mem2, X2) else mem2.contiguous()
            mem2 = ops.convert(src, self._empty(B * S, self.d))
        return mem2

    def decode_init(self, mem, attention_mask: torch.Tensor, beams: int = 1, max_len: int = 128):
        """State for token-by-token decoding against fixed encoder memory (eval semantics): the
        cross-attention K/V of every decoder layer are projected ONCE, self-attention K/V are
        appended to a (B*beams, max_len, 2d) cache per layer -- instead of the reference's
        `use_cache=False` full-prefix recompute (wrapper.py:443-451, custom_modeling.py:279-281)."""
        B, S = attention_mask.shape
        d, Ld = self.d, self.cfg["decoder_layers"]
        st = {"B": B, "S": S, "k": int(beams), "Tmax": int(max_len), "t": 0,
              "mem_pad": (attention_mask == 0).to(torch.uint8).contiguous()}
        mem2 = self._mem_rows(mem, B, S)
        st["xkv"] = [self._linear(mem2, f"decoder.layers.{i}.multihead_attn.in_proj_weight", 3 * d, d, d, 3 * d,
                                  bias_name=f"decoder.layers.{i}.multihead_attn.in_proj_bias") for i in range(Ld)]
        # self-attention K|V cache, one (B*beams * max_len, 2d) matrix per layer: row (b, t) = b * max_len + t
        st["cache"] = [X2.zeros(B * beams * max_len, 2 * d, self.dev) if self.x3 else
                       torch.zeros(B * beams * max_len, 2 * d, dtype=self.cd, device=self.dev) for _ in range(Ld)]
        st["pe"] = self._pos_rows(max_len, None).contiguous()
        return st

    def decode_init_graphed(self, mem, attention_mask: torch.Tensor, max_len: int = 128):
        """Greedy decoding with one captured HIP graph per position: a decode step is ~70 launches of a few
        microseconds each, so eager stepping is bound by the host (1.4 ms per token at B = 128); the graph of
        position t (its cache slot, positional row and key count baked in) replays as ONE launch.  The state
        (KV caches, projected memory, token buffer, graphs) is kept per (B, S, max_len) and reused by later
        calls; only the cross-attention K/V projection and the pad mask are refreshed, in place."""
        B, S = attention_mask.shape
        key = (B, S, int(max_len))
        st = self._graph_states.get(key)
        d, Ld = self.d, self.cfg["decoder_layers"]
        mem2 = self._mem_rows(mem, B, S)
        if st is None:
            st = self.decode_init(mem, attention_mask, 1, max_len)
            st["ids_static"] = torch.zeros(B, dtype=torch.int64, device=self.dev)
            st["graphs"] = {}
            st["pool"] = torch.cuda.graph_pool_handle()
            self.decode_step(st, st["ids_static"])      # eager warm-up: one-time function attributes, allocator
            torch.cuda.synchronize()
            while len(self._graph_states) >= 2:          # at most two decode shapes stay resident (caches + graphs)
                self._graph_states.pop(next(iter(self._graph_states)))
            self._graph_states[key] = st
        else:
            st["mem_pad"].copy_((attention_mask == 0).to(torch.uint8))
            if self.pos_enc is None:   # learned positions are parameters: refresh the rows the graphs read, in place
                tab = self.ps.p("embedding.positional_encodings.pos_encodings.weight")[:int(max_len)]
                ops.layernorm_fwd(tab, self.ps.p("embedding.positional_encodings.norm.weight"),
                                  self.ps.p("embedding.positional_encodings.norm.bias"), st["pe"])
            for i in range(Ld):
                self._linear(mem2, f"decoder.layers.{i}.multihead_attn.in_proj_weight", 3 * d, d, d, 3 * d,
                             bias_name=f"decoder.layers.{i}.multihead_attn.in_proj_bias", out=st["xkv"][i])
        st["t"] = 0
        return st

    def decode_step_graphed(self, st, ids: torch.Tensor) -> torch.Tensor:
        """decode_step through the captured graph of position st['t'] (captured on first use).  The returned
        logits live in the graph's memory: consume them before replaying the same position again."""
        t = st["t"]
        st["ids_static"].copy_(ids.view(-1))
        ent = st["graphs"].get(t)
        if ent is None:     # graphs read weights / positional rows / caches through fixed pointers that are updated in place
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=st["pool"]):
                out = self.decode_step(st, st["ids_static"])
            ent = (g, out)
            st["graphs"][t] = ent
        ent[0].replay()
        st["t"] = t + 1
        return ent[1]

    def decode_reorder(self, st, beam_idx: torch.Tensor) -> None:
        """Beam search bookkeeping: row r of every cache continues beam beam_idx[r].  One afm_cache_reorder launch per
        layer copies the st['t'] filled positions of every beam row into the alternate buffer (the buffers swap)."""
        Bk, Tmax = st["B"] * st["k"], st["Tmax"]
        idx = beam_idx if beam_idx.dtype == torch.int32 else beam_idx.to(torch.int32)
        if "cache_alt" not in st:
            st["cache_alt"] = [(X2.empty(c.shape[0], c.shape[1], self.dev) if self.x3 else torch.empty_like(c)) for c in st["cache"]]
        for i, (c, alt) in enumerate(zip(st["cache"], st["cache_alt"])):
            row_bytes = ops._ld(c) * (2 if self.lowp else 4)          # physical bytes of one position (both planes of a pair)
            ops.cache_reorder(c, alt, idx, Bk, Tmax * row_bytes, st["t"] * row_bytes)
        st["cache"], st["cache_alt"] = st["cache_alt"], st["cache"]

    def decode_step(self, st, ids: torch.Tensor) -> torch.Tensor:
        """Feed token ids (B*beams,) at position st['t']; returns fp32 logits (B*beams, V)."""
        assert not self.training, "decode runs with eval semantics (no dropout)"
        B, S, k, Tmax, t = st["B"], st["S"], st["k"], st["Tmax"], st["t"]
        if t >= Tmax:
            raise ValueError("decode past max_len")
        d, Bk = self.d, B * k
        H, dh = self.cfg["decoder_attention_heads"], self.d // self.cfg["decoder_attention_heads"]
        m = self.tm
        e = torch.empty(Bk, d, dtype=torch.float32, device=self.dev)
        ops.gather_rows(ids.contiguous().view(-1), self.ps.p(f"embedding.embedding_layer_dict.{m}.weight"), e)
        x = torch.empty(Bk, d, dtype=torch.float32, device=self.dev)
        if self.norm:
            ops.layernorm_fwd(e, self.ps.p(f"embedding.embedding_norm_dict.{m}.weight"),
                              self.ps.p(f"embedding.embedding_norm_dict.{m}.bias"), x, pos=st["pe"][t:t + 1],
                              seg_len=1, out_seg_stride=1, out_off=0)
        else:
            ops.place_rows(e, x, pos=st["pe"][t:t + 1], seg_len=1, out_seg_stride=1, out_off=0)
        pend = None
        pre = self.pre_ln
        h = None if pre else self._operand(x)
        for i in range(self.cfg["decoder_layers"]):
            p = f"decoder.layers.{i}."
            # causal self-attention over the cache: the new token sees positions 0..t
            if pre:
                h, x = self._ln_fwd(x, p + "norm1.", None, None, pend=pend)
            qkv = self._linear(h, p + "self_attn.in_proj_weight", 3 * d, d, bias_name=p + "self_attn.in_proj_bias")
            c2 = st["cache"][i]                      # (Bk*Tmax, 2d): append this position's K|V
            kv_new = qkv[:, d:]
            for dst, src in (((c2.hi, kv_new.hi), (c2.lo, kv_new.lo)) if self.x3 else ((c2, kv_new),)):
                dst.view(Bk, Tmax, 2 * d)[:, t, :].copy_(src)
            a = self._empty(Bk, d)
            lse = torch.empty(Bk * H, dtype=torch.float32, device=self.dev)
            lq, lc, la = ops._ld(qkv), ops._ld(c2), ops._ld(a)
            shp = ops.attn_shape(Bk, H, 1, t + 1, dh, self.cd, lq, lc, lc, la, None, False, ops.NO_DROP,
                                 self.algo, batch_strides=(lq, Tmax * lc, Tmax * lc, la))
            ops.attn_fwd(shp, qkv[:, :d], c2[:, :d], c2[:, d:], a, lse)
            pend = self._linear(a, p + "self_attn.out_proj.weight", d, d, bias_name=p + "self_attn.out_proj.bias",
                                out_dtype=self.branch_dtype)
            # cross-attention: the k beams of a sample are k query rows against that sample's memory
            if pre:
                h, x = self._ln_fwd(x, p + "norm2.", None, None, pend=pend)
            else:
                x, h = self._post_norm(x, pend, p + "norm1.", None, None)
            w, bname = p + "multihead_attn.in_proj_weight", p + "multihead_attn.in_proj_bias"
            q = self._linear(h, w, 3 * d, d, 0, d, bias_name=bname)
            kv = st["xkv"][i]
            a = self._empty(Bk, d)
            lse = torch.empty(Bk * H, dtype=torch.float32, device=self.dev)
            shp = ops.attn_shape(B, H, k, S, dh, self.cd, ops._ld(q), ops._ld(kv), ops._ld(kv), ops._ld(a), st["mem_pad"],
                                 False, ops.NO_DROP, self.algo)
            ops.attn_fwd(shp, q, kv[:, :d], kv[:, d:], a, lse)
            pend = self._linear(a, p + "multihead_attn.out_proj.weight", d, d,
                                bias_name=p + "multihead_attn.out_proj.bias", out_dtype=self.branch_dtype)
            if pre:
                x, pend = self._ffn_fwd(x, pend, p, self.cfg["decoder_ffn_dim"], "norm3.", None, f"d{i}")
            else:
                x, h = self._post_norm(x, pend, p + "norm2.", None, None)
                _, pend = self._ffn_fwd(None, None, p, self.cfg["decoder_ffn_dim"], "norm3.", None, f"d{i}", h=h)
                x, h = self._post_norm(x, pend, p + "norm3.", None, None)
                pend = None
        _, logits = self._head(x, pend, None)
        st["t"] = t + 1
        return logits

    def align_head(self, mem, mem_pad, B, S, target, backward: bool, loss_scale: float):
        """Encoder alignment loss (custom_modeling.py:453-475): masked mean of the encoder output ->
        Linear/ReLU stack (the two Conv1d layers act on a length-1 sequence: with padding k//2 only the
        centre tap of the first one sees data) -> sigmoid -> mse / mae / sid.  With `backward` the head's
        parameter gradients are accumulated (scaled by lambda * loss_scale) and the gradient w.r.t. the
        encoder output is returned as the INITIAL value of the (B*S, d) fp32 memory gradient."""
        ac, d = self.align, self.d
        hid, n_out = int(ac["hidden_dimension"]), int(ac["output_dimension"])
        conv = ac["align_network"] == "convolutional"
        P, G = self.ps.p, self.ps.g
        f32 = dict(dtype=torch.float32, device=self.dev)
        pooled = torch.empty(B, d, **f32)
        ops.masked_mean_fwd(mem, mem_pad, B, S, pooled)
        a0 = torch.empty(B, hid, **f32)
        ops.gemm(pooled, P("align_network.0.weight"), a0, trans_b=True, bias=P("align_network.0.bias"), act=ACT_RELU)
        z = torch.empty(B, n_out, **f32)
        if not conv:
            ops.gemm(a0, P("align_network.2.weight"), z, trans_b=True, bias=P("align_network.2.bias"))
        else:
            C, k = int(ac["conv_channels"]), int(ac["kernel_size"])
            a1 = torch.empty(B, hid, **f32)
            ops.gemm(a0, P("align_network.2.weight"), a1, trans_b=True, bias=P("align_network.2.bias"))
            wc = P("align_network.4.weight").view(C, hid, k)[:, :, k // 2].contiguous()     # centre tap (C, hid)
            a2 = torch.empty(B, C, **f32)
            ops.gemm(a1, wc, a2, trans_b=True, bias=P("align_network.4.bias"), act=ACT_RELU)
            w6 = P("align_network.6.weight").view(n_out, C)
            ops.gemm(a2, w6, z, trans_b=True, bias=P("align_network.6.bias"))
        stats = torch.zeros(1, **f32)
        dz = torch.empty_like(z) if backward else None
        tgt = target.to(device=self.dev, dtype=torch.float32).contiguous()
        ops.align_loss(z, tgt, ac["loss_function"], float(ac["loss_lambda"]) * loss_scale, stats, dz,
                       scale_dev=self.scaler if backward else None)
        if not backward:
            return stats[0], None

        def layer_bwd(g, x_in, wname, w, gw_sink=None):
            """g = dL/d(layer output): accumulates dW, db; returns dL/d(layer input)."""
            gw = G(wname + "weight") if gw_sink is None else gw_sink
            ops.gemm(g, x_in, gw.view(g.shape[1], x_in.shape[1]), trans_a=True, trans_b=False, accumulate=True)
            ops.colsum(g, G(wname + "bias"), accumulate=True)
            gi = torch.empty(g.shape[0], x_in.shape[1], **f32)
            ops.gemm(g, w, gi, trans_b=False)
            return gi

        if not conv:
            g = layer_bwd(dz, a0, "align_network.2.", P("align_network.2.weight"))
        else:
            g = ops.relu_bwd(layer_bwd(dz, a2, "align_network.6.", w6), a2)
            gwc = torch.zeros(C, hid, **f32)
            g = layer_bwd(g, a1, "align_network.4.", wc, gw_sink=gwc)
            G("align_network.4.weight").view(C, hid, k)[:, :, k // 2] += gwc     # the other taps only ever see padding
            g = layer_bwd(g, a0, "align_network.2.", P("align_network.2.weight"))
        g = ops.relu_bwd(g, a0)
        dpooled = layer_bwd(g, pooled, "align_network.0.", P("align_network.0.weight"))
        dmem = torch.empty(B * S, d, **f32)
        ops.masked_mean_bwd(dpooled, mem_pad, B, S, dmem, accumulate=False)
        return stats[0], dmem

    def forward(self, enc_inputs, attention_mask, dec_ids, dec_attention_mask=None, labels=None,
                backward: bool = False, loss_scale: float = 1.0, memory=None, encoder_align_target=None):
        """One pass.  With `backward` the parameter gradients are ACCUMULATED into the flat
        gradient buffer (scaled by loss_scale, e.g. 1/acc_batches).  Returns a dict with fp32
        `logits` (B,T,V), `loss` (0-dim device tensor, mean over labels != -100), `argmax`
        (B,T) int64, `encoder_hidden_states`."""
        saved = {} if backward else None
        B, T = dec_ids.shape
        dec_ids = dec_ids.contiguous()
        self._fwd_live, self._frole = {}, None
        if memory is None:
            mem, mem_pad = self.encode(enc_inputs, attention_mask, saved, dec_len=T)
        else:
            mem = memory
            mem_pad = (attention_mask == 0).to(torch.uint8).contiguous()
        S = attention_mask.shape[1]
        align_loss = None
        if self.align and memory is None and encoder_align_target is not None and (labels is not None or backward):
            align_loss, dmem0 = self.align_head(mem, mem_pad, B, S, encoder_align_target, backward, loss_scale)
            if backward:
                saved["dmem_init"] = dmem0
        logits = self.decode(dec_ids, mem, mem_pad, dec_attention_mask, S, saved)
        self._fwd_live, self._enc_off, self._arena_rows = {}, None, 0
        # (split-pair memory stays a 2-D X2 object: .float() / .cpu() materialise it on demand)
        out = {"logits": logits.view(B, T, self.V),
               "encoder_hidden_states": mem if isinstance(mem, X2) else mem.view(B, S, self.d)}
        if saved is not None and saved.get("plan") is not None:
            # a training step with the padded rows out of the forward: position s of sample b is ROW encoder_row_map[b, s] of
            # encoder_hidden_states seen as a (B*S, d) matrix, and rows of nothing but padding hold zeros (AFM_FWD_ROW_SKIP=0: the
            # reference's layout)
            out["encoder_row_map"] = saved["plan"].dest.view(B, S)
        rows = B * T
        if labels is not None or backward:
            lab = labels.contiguous().view(-1)
            row_lse = torch.empty(rows, dtype=torch.float32, device=self.dev)
            argmax = torch.empty(rows, dtype=torch.int64, device=self.dev)
            stats = torch.zeros(2, dtype=torch.float32, device=self.dev)
            ops.ce_fwd(logits, lab, row_lse, argmax, stats)
            out["loss"] = stats[0] / stats[1]
            if align_loss is not None:     # total = lm + lambda * align (custom_modeling.py:492-497)
                out["loss_dict"] = {"model_only_loss": out["loss"], "alignment_loss": align_loss}
                out["loss"] = out["loss"] + float(self.align["loss_lambda"]) * align_loss
            out["argmax"] = argmax.view(B, T)
            out["loss_stats"] = stats
            if backward:
                self._backward(saved, logits, lab, row_lse, stats, loss_scale, mem)
                self.micro_step += 1
        return out

    def _backward_post(self, saved, dhf, mem):
        """_backward for post-LN layers (post_layer_normalisation=False): every sublayer is y = LN(x + dropout(block(x))), so the
        stream gradient passes THROUGH each LayerNorm (no bypass) and the block's input gradient is added behind it."""
        d = self.d
        B, S = saved["B"], saved["S"]
        Ld, Le = self.cfg["decoder_layers"], self.cfg["encoder_layers"]
        fd, fe = self.cfg["decoder_ffn_dim"], self.cfg["encoder_ffn_dim"]
        dx, _ = self._ln_bwd(dhf, "decoder.norm.", saved, "dec_norm", dres=None, next_site=None)
        dmem = saved.pop("dmem_init", None)
        had_init = dmem is not None
        if dmem is None and not self.lowp:
            dmem = torch.zeros(B * S, d, dtype=torch.float32, device=self.dev)
        dkv_all = self._empty_dkv_all(B * S, Ld * 2 * d) if self.lowp and Ld > 0 else None
        for i in range(Ld - 1, -1, -1):
            p, sv = f"decoder.layers.{i}.", saved["dec_layers"][i]
            dx = self._post_sub_bwd(dx, p + "norm3.", sv, "lnf", f"d{i}res2",
                                    lambda dy, acc: self._ffn_bwd(None, dy, p, fd, "norm3.", sv, None, acc=acc))
            dx = self._post_sub_bwd(dx, p + "norm2.", sv, "ln2", f"d{i}xres",
                                    lambda dy, acc: self._cross_attn_bwd(None, dy, mem, dmem, p, sv, None, dkv_all, i, acc=acc))
            dx = self._post_sub_bwd(dx, p + "norm1.", sv, "ln1", f"d{i}res",
                                    lambda dy, acc: self._self_attn_bwd(None, dy, p, sv, None, acc=acc))
            self._grads_final_from(p + "self_attn.in_proj_weight")
        if dkv_all is not None:
            if dmem is None:
                dmem = torch.empty(B * S, d, dtype=torch.float32, device=self.dev)
            ops.gemm(dkv_all, self._hb(self.wt_kv_all), dmem, trans_b=True, accumulate=had_init, algo=self.algo)
        self._embed_bwd_unhinted(dx, saved["emb_dec"])
        self._role = "enc"
        dx, _ = self._ln_bwd(self._operand(dmem, backward=True), "encoder.norm.", saved, "enc_norm", dres=None, next_site=None)
        for i in range(Le - 1, -1, -1):
            p, sv = f"encoder.layers.{i}.", saved["enc_layers"][i]
            dx = self._post_sub_bwd(dx, p + "norm2.", sv, "lnf", f"e{i}res2",
                                    lambda dy, acc: self._ffn_bwd(None, dy, p, fe, "norm2.", sv, None, acc=acc))
            dx = self._post_sub_bwd(dx, p + "norm1.", sv, "ln1", f"e{i}res",
                                    lambda dy, acc: self._self_attn_bwd(None, dy, p, sv, None, acc=acc))
            self._grads_final_from(p + "self_attn.in_proj_weight")
        self._embed_bwd_unhinted(dx, saved["emb_enc"])
        self._wgrad_flush()
        self._live, self._role = {}, None
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)

    def _grads_final_from(self, first_name: str) -> None:
        """Parameters are laid out in forward order, so once a layer's backward is done every
        gradient from its first tensor to the end of the flat buffer is final."""
        # AFM_WGRAD_LAYERS=n: the weight gradients of n consecutive layers in one grouped launch (A/B probe; only without a gradient hook,
        # whose buckets want every layer's gradients as soon as they are final)
        if self.wgrad_layers > 1 and self.grad_ready_hook is None:
            self._wg_layers_pending += 1
            if self._wg_layers_pending < self.wgrad_layers:
                return
            self._wg_layers_pending = 0
        self._wgrad_flush()
        if self.grad_ready_hook is not None:
            if self.wgrad_stream is not None:
                torch.cuda.current_stream().wait_stream(self.wgrad_stream)
            self.grad_ready_hook(self.ps.specs[first_name].offset)

    def _embed_bwd_unhinted(self, dx, saved_emb):
        """The padded-row hints (`_live`) belong to the layer stacks (role "enc": B * S encoder rows, "dec": B * T decoder rows).  A
        modality's rows are neither (its own padding differs): the embedder backward runs with no role, hence without hints."""
        keep, self._role = self._role, None
        try:
            self.embed_bwd(dx, saved_emb)
        finally:
            self._role = keep

    def _live_hint(self, t, role=None):
        """Padded-row hint for a backward operand `t` (one byte per 64-row block; None: no hint), keyed by ROLE: `role` where the
        caller names it (the memory-side operands of the decoder's cross-attention are encoder rows), else the stack `_backward` is
        in (`_role`).  The row count is only a consistency check -- an operand whose rows are not the role's gets no hint, never another
        role's.  AFM_DEBUG_LIVE=1 checks, with a host synchronisation, that every row the hint calls dead really is zero."""
        role = role or self._role
        h = self._live.get(role) if (role is not None and torch.is_tensor(t)) else None
        if h is not None and h.numel() * 64 != t.shape[0]:
            h = None
        if h is not None and _DEBUG_LIVE:
            rows = t.hi if hasattr(t, "hi") else t
            dead = ((h.t if isinstance(h, ops.RowFlags) else h) == 0).repeat_interleave(64)
            assert float(rows[dead].float().abs().max() if bool(dead.any()) else 0.0) == 0.0, "a row marked dead by the padded-row hint is not zero"
        return h

    def _backward(self, saved, logits, lab, row_lse, stats, loss_scale, mem):
        d = self.d
        B, S, T = saved["B"], saved["S"], saved["T"]
        self._wg_pending = []          # (a backward pass that raised may have left entries behind)
        if self._xattn_pending:
            torch.cuda.current_stream().wait_stream(self.xattn_stream)
            self._xattn_pending = False
        self._wg_layers_pending = 0
        # Padded positions are masked as keys everywhere and take no part in the loss: their rows of every activation gradient are
        # exact zeros.  One byte per 64-row block tells the weight-gradient kernels (token axis) and the LayerNorm backward which
        # blocks hold nothing else (include/afm_hip.h: afm_gemm_desc.k_live, afm_ln_shape.row_live).
        self._live, self._role = {}, None
        if self.row_skip:
            tgt_pad = saved.get("tgt_pad")
            if tgt_pad is not None:      # a padded decoder row is dead only if it has no label either (the caller's labels are its own)
                tgt_pad = tgt_pad.view(B, T).bool() & (lab.view(B, T) == -100)
            for role, L, pad in (("enc", S, saved.get("key_pad")), ("dec", T, tgt_pad)):
                if role == "enc" and saved.get("plan") is not None:
                    pl = saved["plan"]
                    nofill = pl.packed and self.bwd_nofill and self._bwd_verified.get((B, S, T)) is True
                    self._live[role] = ops.RowFlags(pl.live64, pl.packed, nofill, "enc")      # (afm_compact_plan already made the flags)
                elif pad is not None and L % 64 == 0:
                    self._live[role] = ops.RowFlags((pad.view(B, L // 64, 64) == 0).any(-1).to(torch.uint8).reshape(-1).contiguous(), False, False, role)
        verify = (self.bwd_nofill and saved.get("plan") is not None and saved["plan"].packed and (B, S, T) not in self._bwd_verified
                  and self.wgrad_stream is None)
        if verify:
            ops.reset_hint_log()
        self._verify_key = (B, S, T) if verify else None
        self._role = "dec"             # head, final decoder norm and the decoder stack: B * T rows
        dlog = self._empty_b(B * T, self.V)
        ops.ce_bwd(logits, lab, row_lse, stats, loss_scale, dlog, scale_dev=self.scaler)
        hf = saved["hf"]
        self._wgrad(dlog, hf, "token_ff.weight", self.V, d, bias_name="token_ff.bias")
        dhf = self._dgrad(dlog, "token_ff.weight", self.V, d)
        Ld, Le = self.cfg["decoder_layers"], self.cfg["encoder_layers"]
        if not self.pre_ln:
            return self._backward_post(saved, dhf, mem)
        dx, dy = self._ln_bwd(dhf, "decoder.norm.", saved, "dec_norm", dres=None, next_site=f"d{Ld - 1}res2")
        dmem = saved.pop("dmem_init", None)     # alignment head's gradient w.r.t. the encoder output, if any
        had_init = dmem is not None
        if dmem is None and not self.lowp:
            dmem = torch.zeros(B * S, d, dtype=torch.float32, device=self.dev)
        dkv_all = None
        if self.lowp and Ld > 0:
            dkv_all = self._empty_dkv_all(B * S, Ld * 2 * d)
        for i in range(Ld - 1, -1, -1):
            p, sv = f"decoder.layers.{i}.", saved["dec_layers"][i]
            dx, dy = self._ffn_bwd(dx, dy, p, self.cfg["decoder_ffn_dim"], "norm3.", sv, f"d{i}xres")
            dx, dy = self._cross_attn_bwd(dx, dy, mem, dmem, p, sv, f"d{i}res", dkv_all, i)
            dx, dy = self._self_attn_bwd(dx, dy, p, sv, f"d{i - 1}res2" if i > 0 else None)
            self._grads_final_from(p + "self_attn.in_proj_weight")
        dmem_c = None
        if dkv_all is not None:      # d(encoder output) = [dK|dV of every layer] @ [their projection weights], K = Ld*2d
            if dmem is None and self.single16:
                # nothing to add to (no alignment head): the GEMM writes the 16-bit operand of the encoder's final LayerNorm
                # backward itself, instead of an fp32 matrix and a cast kernel behind it (the same rounding, once)
                dmem_c = self._empty_b(B * S, d)
                ops.gemm(dkv_all, self._hb(self.wt_kv_all), dmem_c, trans_b=True, algo=self.algo, k_live=self._live_hint(dkv_all, "enc"))
            else:
                if dmem is None:
                    dmem = torch.empty(B * S, d, dtype=torch.float32, device=self.dev)
                ops.gemm(dkv_all, self._hb(self.wt_kv_all), dmem, trans_b=True, accumulate=had_init, algo=self.algo)
        self._embed_bwd_unhinted(dx, saved["emb_dec"])
        if dmem_c is None:
            dmem_c = dmem
            if self.lowp:
                dmem_c = self._empty_b(B * S, d)
                ops.dropout_cast(dmem, dmem_c)
        self._role = "enc"             # final encoder norm and the encoder stack: B * S rows
        dx, dy = self._ln_bwd(dmem_c, "encoder.norm.", saved, "enc_norm", dres=None, next_site=f"e{Le - 1}res2")
        for i in range(Le - 1, -1, -1):
            p, sv = f"encoder.layers.{i}.", saved["enc_layers"][i]
            dx, dy = self._ffn_bwd(dx, dy, p, self.cfg["encoder_ffn_dim"], "norm2.", sv, f"e{i}res")
            dx, dy = self._self_attn_bwd(dx, dy, p, sv, f"e{i - 1}res2" if i > 0 else None)
            self._grads_final_from(p + "self_attn.in_proj_weight")
        self._embed_bwd_unhinted(dx, saved["emb_enc"])
        self._wgrad_flush()
        if self._verify_key is not None:      # this backward ran WITH its zero fills: did every hinted call take its hint?
            took, ignored = ops.hint_log("enc")      # (the encoder rows' hints: the decoder rows keep their fills, ignored or not)
            self._bwd_verified[self._verify_key] = ignored == 0 and took > 0
            self._verify_key = None
        self._live, self._role = {}, None
        if self.wgrad_stream is not None:   # every weight gradient is in before the caller reads the buffer
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
