"""Torch-tensor front end of the C ABI: every function here is one libafm_hip.so call.

Tensors are only pointer + shape carriers.  All calls enqueue on torch's CURRENT stream, so
they compose with torch.cuda.graph capture and with side streams used for the gradient
all-reduce.  Nothing here computes on the CPU or through torch ops.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import lib as L
from .lib import AFM_BF16, AFM_BF16X2, AFM_F16, AFM_F32, ALGO_AUTO, ACT_GELU, ACT_NONE, ACT_RELU, AttnShape, Dropout, GemmDesc, LnShape
from .x2 import X2

_DT = {torch.float32: AFM_F32, torch.bfloat16: AFM_BF16, X2.dtype: AFM_BF16X2, torch.float16: AFM_F16}


def _dt(t) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise L.AfmError(f"unsupported dtype {t.dtype}") from None


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, X2):
        t = t.hi
    if not t.is_cuda:
        raise L.AfmError("libafm_hip operates on device memory only (got a CPU tensor)")
    return t.data_ptr()


def empty(rows: int, cols: int, dtype, device):
    """(rows x cols) activation buffer of a compute dtype (torch dtype or X2.dtype)."""
    if dtype == X2.dtype:
        return X2.empty(rows, cols, device)
    return torch.empty(rows, cols, dtype=dtype, device=device)


def is_contig(t) -> bool:
    """Rows back to back (an X2 tensor: hi | lo planes of a row back to back, ld = 2 n)."""
    if isinstance(t, X2):
        return t.ld == 2 * t.shape[1]
    return t.is_contiguous()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ld(t) -> int:
    """Row stride (elements) of a 2-D view whose last dim is contiguous."""
    if isinstance(t, X2):
        return t.ld
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise L.AfmError(f"expected a row-major 2-D view, got shape {tuple(t.shape)} stride {t.stride()}")
    return int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))


def drop(p: float = 0.0, seed: int = 0, site: int = 0) -> Dropout:
    return Dropout(float(p), int(site) & 0xFFFFFFFF, int(seed) & 0xFFFFFFFFFFFFFFFF)


NO_DROP = drop()


def last_algo() -> str:
    return L.load().afm_last_algo().decode()


_ALGO_LOG = None   # tests: names of the kernel families dispatched by gemm / attention calls since reset_algo_log()


def reset_algo_log() -> None:
    global _ALGO_LOG
    _ALGO_LOG = []


def algo_log():
    global _ALGO_LOG
    out, _ALGO_LOG = _ALGO_LOG or [], None
    return out


def _log_algo() -> None:
    if _ALGO_LOG is not None:
        _ALGO_LOG.append(last_algo())


_DEBUG_SYNC = bool(os.environ.get("AFM_DEBUG_SYNC"))   # debugging aid: synchronise and name the attention call that faulted



class RowFlags:
    """A padded-row hint with its layout: `t` = one byte per 64-row block (0 = nothing but padding), `dealt` = the live rows are PACKED
    to the front of the matrix (afm_compact_plan mode 2), so the GEMM kernels deal row panels / k-steps round-robin to XCDs / split-K units
    instead of cutting contiguous bands (afm_gemm_desc.reserved2 bit 2).  Every hint argument below takes a plain uint8 tensor too."""
    __slots__ = ("t", "dealt", "nofill", "tag")

    def __init__(self, t, dealt=False, nofill=False, tag=""):
        self.tag = tag      # whose rows (the engine's role): the hint log below is kept per tag
        # nofill (forward sense only): dead rows are not written at all -- the outputs are persistent buffers that already hold finite
        # values there (afm_gemm_desc.reserved2 bit 3, afm_ln_shape.flags bit 0)
        self.t, self.dealt, self.nofill = t, bool(dealt), bool(nofill)

    def numel(self):
        return self.t.numel()


def _flags(h):
    return (h.t, h.dealt) if isinstance(h, RowFlags) else (h, False)


# Padded-row hints of the calls since reset_hint_log(): [honoured, ignored] (afm_last_hint after every hinted afm_gemm / afm_gemm_group /
# afm_layernorm_bwd).  The engine runs a backward WITH the zero fills, reads this, and only where nothing was ignored lets the next steps'
# backward kernels leave dead rows unwritten (RowFlags.nofill): from then on an ignored hint is an error, not a slower path.
_HINT_LOG: dict = {}


def reset_hint_log() -> None:
    _HINT_LOG.clear()


def hint_log(tag=""):
    return tuple(_HINT_LOG.get(tag, (0, 0)))


def _note_hint(tags, nofill: bool, what: str) -> None:
    """`tags`: the tags of the hints the call was given (empty: none)."""
    if not tags:
        return
    st = L.load().afm_last_hint()
    for tag in set(tags):
        e = _HINT_LOG.setdefault(tag, [0, 0])
        e[0 if st > 0 else 1] += 1
    if st <= 0:
        if os.environ.get("AFM_DEBUG_HINTS"):
            print(f"[afm] hint ignored: {what} ({last_algo()})", flush=True)
        if nofill:
            raise L.AfmError(f"{what}: a padded-row hint was ignored while dead rows are left unwritten (nofill): results would be undefined")


def gemm_desc(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, *, trans_a=False, trans_b=True,
              bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
              pre_act: Optional[torch.Tensor] = None, act: int = ACT_NONE, accumulate: bool = False,
              dropout: Dropout = NO_DROP, algo: int = ALGO_AUTO, a_colsum: Optional[torch.Tensor] = None,
              variant: int = 0, glu_rows: int = 0, sg_hi_only: bool = False, k_live: Optional[torch.Tensor] = None,
              rows_unread: Optional[torch.Tensor] = None) -> GemmDesc:
    """The afm_gemm_desc of c = epilogue(op(a) @ op(b)) (arguments as `gemm`); the caller keeps the tensors alive until launch."""
    M, N = c.shape
    if act in (L.ACT_GLU, L.ACT_GLU_SAVE):
        N *= 2
    elif act == L.ACT_GLU_BWD:
        N //= 2
    K = a.shape[0] if trans_a else a.shape[1]
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.transA, d.transB = int(trans_a), int(trans_b)
    d.lda, d.ldb, d.ldc = _ld(a), _ld(b), _ld(c)
    d.a_dtype, d.b_dtype, d.c_dtype = _dt(a), _dt(b), _dt(c)
    d.A, d.B, d.C = _ptr(a), _ptr(b), _ptr(c)
    exp_a = (K, M) if trans_a else (M, K)
    exp_b = (N, K) if trans_b else (K, N)
    if tuple(a.shape) != exp_a or tuple(b.shape) != exp_b:
        raise L.AfmError(f"gemm shape mismatch: a {tuple(a.shape)} b {tuple(b.shape)} c {tuple(c.shape)}")
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    d.bias = _ptr(bias)
    if residual is not None:
        assert residual.dtype == c.dtype and _ld(residual) == d.ldc and residual.shape == c.shape
    d.residual = _ptr(residual)
    if pre_act is not None and act in (L.ACT_GLU_SAVE, L.ACT_GLU_BWD):
        assert pre_act.dtype == c.dtype and tuple(pre_act.shape) == (M, max(N, c.shape[1]))
        if act == L.ACT_GLU_SAVE:     # the saved factors are written with their own dense row stride
            assert is_contig(pre_act) and is_contig(c)
        else:                         # read back with C's row stride (hi-plane views in mixed mode: equal strides, not dense)
            assert _ld(pre_act) == _ld(c)
    elif pre_act is not None:
        assert pre_act.dtype == c.dtype and _ld(pre_act) == d.ldc and pre_act.shape == c.shape
    d.pre_act = _ptr(pre_act)
    if a_colsum is not None:
        assert trans_a and a_colsum.dtype == torch.float32 and a_colsum.numel() == M and a_colsum.is_contiguous()
    d.a_colsum = _ptr(a_colsum)
    d.act, d.accumulate, d.algo = int(act), int(accumulate), int(algo)
    d.reserved = int(variant)
    d.glu_rows = int(glu_rows)
    d.reserved2 = 1 if sg_hi_only else 0
    d.drop = dropout
    nofill = (isinstance(rows_unread, RowFlags) and rows_unread.nofill) or (isinstance(k_live, RowFlags) and k_live.nofill and not trans_a)
    d._hint_tag = k_live.tag if isinstance(k_live, RowFlags) else (rows_unread.tag if isinstance(rows_unread, RowFlags) else "")
    k_live, dealt = _flags(k_live)
    rows_unread, dealt_u = _flags(rows_unread)
    if dealt or dealt_u:
        d.reserved2 |= 4
    if nofill:
        d.reserved2 |= 8
    if k_live is not None:      # one byte per 64 stored rows of `a` (token positions), 0 = all of them zero (padding)
        nrows = K if trans_a else M
        assert (trans_a or trans_b) and k_live.dtype == torch.uint8 and k_live.is_contiguous()
        assert nrows % 64 == 0 and k_live.numel() == nrows // 64
    d.k_live = _ptr(k_live)
    if rows_unread is not None:     # k_live in the FORWARD sense: blocks of 64 output rows nobody reads (0) are written as zeros, not computed
        assert k_live is None and not trans_a and trans_b and rows_unread.dtype == torch.uint8 and rows_unread.is_contiguous()
        assert M % 64 == 0 and rows_unread.numel() == M // 64
        d.k_live = _ptr(rows_unread)
        d.reserved2 |= 2
    return d


def gemm(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, **kw) -> torch.Tensor:
    """c = epilogue(op(a) @ op(b)); default is the nn.Linear form c = a @ b^T + bias.
    a_colsum (trans_a only): a_colsum[m] += sum_k a[k, m], the bias gradient of the wgrad form.
    Gated-FFN forms (act 6 / 7 / 8, include/afm_hip.h): c is (M, N/2) resp. (M, 2N); glu_rows = f."""
    d = gemm_desc(a, b, c, **kw)
    L.check(L.load().afm_gemm(C.byref(d), _stream()), "afm_gemm")
    _log_algo()
    _note_hint([d._hint_tag] if d.k_live else [], bool(d.reserved2 & 8) and not (d.reserved2 & 2), f"afm_gemm M={d.M} N={d.N} K={d.K} ta={d.transA} act={d.act}")
    if _DEBUG_SYNC:
        print(f"[afm] gemm {last_algo()} M={d.M} N={d.N} K={d.K} ta={d.transA} tb={d.transB} ld=({d.lda},{d.ldb},{d.ldc}) dt=({d.a_dtype},{d.b_dtype},{d.c_dtype}) "
              f"act={d.act} glu={d.glu_rows}", flush=True)
        torch.cuda.synchronize()
    return c


def gemm_group(descs) -> None:
    """afm_gemm_group: the GEMMs of `descs` (gemm_desc objects), weight-gradient problems fused into shared launches."""
    if not descs:
        return
    arr = (GemmDesc * len(descs))(*descs)
    L.check(L.load().afm_gemm_group(C.cast(arr, C.c_void_p), len(descs), _stream()), "afm_gemm_group")
    _log_algo()
    _note_hint([d._hint_tag for d in descs if d.k_live], False, "afm_gemm_group " + " ".join(f"{d.M}x{d.N}x{d.K}" for d in descs))


def gather_rows(ids, table, out, scale=None):
    n = ids.numel()
    assert ids.dtype == torch.int64 and ids.is_contiguous() and out.is_contiguous()
    V, d = table.shape
    L.check(L.load().afm_gather_rows(_ptr(ids), _ptr(scale), _ptr(table), _ptr(out), n, d, V, _stream()),
            "afm_gather_rows")
    return out


def scatter_add_rows(ids, dout, dtable, scale=None, padding_idx=-1):
    n = ids.numel()
    V, d = dtable.shape
    assert ids.dtype == torch.int64 and ids.is_contiguous() and dout.is_contiguous()
    L.check(L.load().afm_scatter_add_rows(_ptr(ids), _ptr(scale), _ptr(dout), _ptr(dtable), n, d, V,
                                          int(padding_idx), _stream()), "afm_scatter_add_rows")


def ln_shape(rows, d, y_dtype, seg_len=0, out_seg_stride=0, out_off=0, eps=1e-5) -> LnShape:
    return LnShape(int(rows), int(d), _DT[y_dtype], int(seg_len), int(out_seg_stride), int(out_off),
                   float(eps), 0)


def layernorm_fwd(x, gamma, beta, y, mean=None, rstd=None, pos=None, seg_len=0, out_seg_stride=0,
                  out_off=0, eps=1e-5, add=None, x_sum=None, add_dropout: Dropout = NO_DROP, row_live=None, row_map=None):
    """y = LN(x [+ dropout(add)]); with `add` the summed stream is also written to x_sum (fp32, may be x).
    row_live (identity row mapping): one byte per 64 rows, 0 = nobody reads those rows of the outputs (zeros written, nothing loaded).
    row_map (placement form): int32 per position of the concatenated sequence, its row inside the sample's slot (afm_compact_plan)."""
    rows, d = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and is_contig(y)
    s = ln_shape(rows, d, y.dtype, seg_len, out_seg_stride, out_off, eps)
    if isinstance(row_live, RowFlags) and row_live.nofill:
        s.flags |= 1
    row_live, _ = _flags(row_live)
    if row_live is not None:
        assert row_live.dtype == torch.uint8 and row_live.is_contiguous() and rows % 64 == 0 and row_live.numel() == rows // 64 and seg_len == 0
        s.row_live = _ptr(row_live)
    if row_map is not None:
        assert row_map.dtype == torch.int32 and row_map.is_contiguous() and seg_len > 0 and row_map.numel() == (rows // seg_len) * out_seg_stride
        s.row_map = _ptr(row_map)
    if add is not None:
        assert add.shape == x.shape and is_contig(add) and x_sum is not None and x_sum.dtype == torch.float32
        s.add_dtype = _dt(add)
        s.add_drop = add_dropout
    L.check(L.load().afm_layernorm_fwd(C.byref(s), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(pos), _ptr(y),
                                       _ptr(mean), _ptr(rstd), _ptr(add), _ptr(x_sum), _stream()),
            "afm_layernorm_fwd")
    return y


def layernorm_bwd_ws(rows, d) -> int:
    s = ln_shape(rows, d, torch.float32)
    return int(L.load().afm_layernorm_bwd_ws_floats(C.byref(s)))


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, ws, dres=None, seg_len=0,
                  out_seg_stride=0, out_off=0, dx_drop=None, dropout: Dropout = NO_DROP, row_live=None, row_map=None):
    rows, d = x.shape
    s = ln_shape(rows, d, dy.dtype, seg_len, out_seg_stride, out_off)
    if row_map is not None:      # the embedder's placement through afm_compact_plan's map (as layernorm_fwd)
        assert row_map.dtype == torch.int32 and row_map.is_contiguous() and seg_len > 0 and row_map.numel() == (rows // seg_len) * out_seg_stride
        s.row_map = _ptr(row_map)
    ln_nofill = isinstance(row_live, RowFlags) and row_live.nofill
    ln_tag = row_live.tag if isinstance(row_live, RowFlags) else ""
    if ln_nofill:
        s.flags |= 1
    row_live, _ = _flags(row_live)
    if row_live is not None:     # one byte per 64 rows, 0 = dy (and dres) are zero there: skipped, zeros written (nofill: left as they are)
        assert row_live.dtype == torch.uint8 and row_live.is_contiguous() and rows % 64 == 0 and row_live.numel() == rows // 64
        assert seg_len == 0
        s.row_live = _ptr(row_live)
    assert ws.numel() >= layernorm_bwd_ws(rows, d)
    if dx_drop is not None:
        assert dx_drop.dtype == dy.dtype and dx_drop.shape == x.shape and is_contig(dx_drop) and is_contig(dy)
    L.check(L.load().afm_layernorm_bwd(C.byref(s), _ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd),
                                       _ptr(dres), _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws),
                                       _ptr(dx_drop), C.byref(dropout), _stream()),
            "afm_layernorm_bwd")
    _note_hint([ln_tag] if row_live is not None else [], ln_nofill, f"afm_layernorm_bwd rows={rows} d={d}")
    return dx


def place_rows(x, y, pos=None, seg_len=0, out_seg_stride=0, out_off=0, gather=False):
    """multimodal_norm = False: rows of x into their slice of the concatenated sequence (+ positional rows), or back (gather)."""
    assert x.dtype == torch.float32 and y.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous()
    rows, d = (y.shape if gather else x.shape)
    L.check(L.load().afm_place_rows(_ptr(x), _ptr(pos), _ptr(y), rows, d, int(seg_len), int(out_seg_stride), int(out_off),
                                    int(gather), _stream()), "afm_place_rows")
    return y


class CompactPlan:
    """afm_compact_plan's outputs for a (B, S) key-padding mask: `dest` (B*S int32: the ROW every position moves to), `seq_off` (B + 1 int32:
    first row of every sample; its last entry = rows in use), `pad` (B, S uint8, the mask over each sample's positions in their new
    order), `live64` / `live_tile` (B*S/64 uint8: exact 64-row blocks / whole `tile_rows` groups holding a live position), `n_live` (B
    int32).  mode 0: flags only, 1: live positions to the front of each sample's own S rows, 2: the batch packed (slots of ceil128(live) rows)."""
    __slots__ = ("dest", "seq_off", "pad", "live64", "live_tile", "n_live", "mode", "B", "S")

    @property
    def compact(self):
        return self.mode != 0

    @property
    def packed(self):
        return self.mode == 2


def compact_plan(key_pad, B, S, tile_rows=256, compact=True) -> CompactPlan:
    """compact: False / 0 = flags only, True / 1 = per-sample compaction, 2 = packed rows (include/afm_hip.h, afm_compact_plan)."""
    assert key_pad.dtype == torch.uint8 and key_pad.is_contiguous() and key_pad.numel() == B * S and S % tile_rows == 0
    dev = key_pad.device
    p = CompactPlan()
    p.B, p.S, p.mode = B, S, int(compact)
    p.dest = torch.empty(B * S, dtype=torch.int32, device=dev)
    p.seq_off = torch.empty(B + 1, dtype=torch.int32, device=dev)
    p.pad = torch.empty(B, S, dtype=torch.uint8, device=dev)
    p.live64 = torch.empty(B * S // 64, dtype=torch.uint8, device=dev)
    p.live_tile = torch.empty(B * S // 64, dtype=torch.uint8, device=dev)
    p.n_live = torch.empty(B, dtype=torch.int32, device=dev)
    L.check(L.load().afm_compact_plan(_ptr(key_pad), B, S, int(tile_rows), p.mode, _ptr(p.dest), _ptr(p.seq_off), _ptr(p.pad), _ptr(p.live64),
                                      _ptr(p.live_tile), _ptr(p.n_live), _stream()), "afm_compact_plan")
    return p


def permute_rows(x, y, row_map, B, S, gather=False):
    """fp32 rows through a compact_plan map (absolute rows): y[map[i]] = x[i], or the reverse (gather)."""
    assert x.dtype == torch.float32 and y.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous() and x.shape == y.shape
    assert row_map.dtype == torch.int32 and row_map.numel() == B * S == x.shape[0]
    L.check(L.load().afm_permute_rows(_ptr(x), _ptr(y), _ptr(row_map), B, S, x.shape[1], int(gather), _stream()), "afm_permute_rows")
    return y


def attn_shape(B, H, Tq, Tk, dh, dtype, ldq, ldk, ldv, ldo, key_pad=None, causal=False,
               dropout: Dropout = NO_DROP, algo=ALGO_AUTO, scale=None, batch_strides=None, q_off=None, k_off=None) -> AttnShape:
    s = AttnShape()
    s.B, s.H, s.Tq, s.Tk, s.dh = B, H, Tq, Tk, dh
    s.dtype = _DT[dtype]
    s.ldq, s.ldk, s.ldv, s.ldo = ldq, ldk, ldv, ldo
    s.causal, s.algo = int(causal), int(algo)
    s.scale = float(scale if scale is not None else dh ** -0.5)
    if key_pad is not None:
        assert key_pad.dtype in (torch.uint8, torch.bool) and key_pad.is_contiguous()
        assert key_pad.numel() == B * Tk
    if batch_strides is not None:   # (sqb, skb, svb, sob) in elements; KV-cache decode
        s.sqb, s.skb, s.svb, s.sob = (int(v) for v in batch_strides)
    s.key_pad = _ptr(key_pad)
    s._keepalive = key_pad  # the struct only holds a raw pointer: keep the mask tensor alive with it
    for name, off in (("q_off", q_off), ("k_off", k_off)):      # packed rows (B + 1 int32: ops.compact_plan(...).seq_off)
        if off is not None:
            assert off.dtype == torch.int32 and off.is_contiguous() and off.numel() == B + 1
            setattr(s, name, _ptr(off))
    s._off_keepalive = (q_off, k_off)
    s.drop = dropout
    return s


def attn_drop_bits_words(B, H, Tq, Tk) -> int:
    """uint64 words of the keep-bit tensor of one attention call (afm_attn_shape.drop_bits)."""
    return B * H * (((Tq + 127) // 128) * 4) * (((Tk + 63) // 64) * 2) * 16


def attn_set_drop_bits(s: AttnShape, bits) -> AttnShape:
    if bits is not None:
        assert bits.dtype == torch.int64 and bits.is_contiguous() and bits.numel() >= attn_drop_bits_words(s.B, s.H, s.Tq, s.Tk)
    s.drop_bits = _ptr(bits)
    s._bits_keepalive = bits
    return s


def attn_fill_drop_bits(s: AttnShape) -> bool:
    """afm_attn_drop_bits_fill on the current stream; True (and the shape's "forward reads the bits" flag set) when the kernels
    of this shape take the tensor, False otherwise (the forward then hashes and writes it as before)."""
    r = L.load().afm_attn_drop_bits_fill(C.byref(s), _stream())
    if r == L.ERR_UNSUPPORTED:
        return False
    L.check(r, "afm_attn_drop_bits_fill")
    s.reserved |= 32
    return True


def _debug_sync(what, s):
    if _DEBUG_SYNC:
        print(f"[afm] {what} B={s.B} H={s.H} Tq={s.Tq} Tk={s.Tk} causal={s.causal} bits={bool(s.drop_bits)} ld=({s.ldq},{s.ldk},{s.ldv},{s.ldo})",
              flush=True)
        torch.cuda.synchronize()


def attn_fwd(s: AttnShape, q, k, v, o, lse):
    _debug_sync("attn_fwd ->", s)
    L.check(L.load().afm_attn_fwd(C.byref(s), _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), _stream()),
            "afm_attn_fwd")
    _log_algo()
    _debug_sync("attn_fwd ok " + last_algo(), s)
    return o


def attn_bwd(s: AttnShape, q, k, v, o, do, lse, delta, dq, dk, dv, lddq, lddk, lddv):
    L.check(L.load().afm_attn_bwd(C.byref(s), _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(do), _ptr(lse),
                                  _ptr(delta), _ptr(dq), _ptr(dk), _ptr(dv), lddq, lddk, lddv, _stream()),
            "afm_attn_bwd")
    _log_algo()
    _debug_sync("attn_bwd ok " + last_algo(), s)


def glu_fwd(u, v, g, dropout: Dropout = NO_DROP, act: int = ACT_GELU):
    """g = dropout(act(u) * (v if v is not None else 1)); act = ACT_GELU or ACT_RELU."""
    rows, f = g.shape
    L.check(L.load().afm_glu_fwd(_ptr(u), _ptr(v), _ptr(g), rows, f, _ld(u), _ld(v) if v is not None else 0,
                                 _ld(g), _dt(g), int(act), C.byref(dropout), _stream()), "afm_glu_fwd")
    return g


def glu_bwd(u, v, dg, du, dv, dropout: Dropout = NO_DROP, act: int = ACT_GELU):
    rows, f = dg.shape
    L.check(L.load().afm_glu_bwd(_ptr(u), _ptr(v), _ptr(dg), _ptr(du), _ptr(dv), rows, f, _ld(u),
                                 _ld(v) if v is not None else 0, _ld(dg), _ld(du),
                                 _ld(dv) if dv is not None else 0, _dt(dg), int(act), C.byref(dropout), _stream()),
            "afm_glu_bwd")


def dropout_cast(x, y, dropout: Dropout = NO_DROP):
    rows, n = x.shape
    assert x.dtype == torch.float32
    L.check(L.load().afm_dropout_cast(_ptr(x), _ptr(y), rows, n, _ld(x), _ld(y), _dt(y), C.byref(dropout),
                                      _stream()), "afm_dropout_cast")
    return y


def colsum(x, out, accumulate=True):
    rows, n = x.shape
    assert out.dtype == torch.float32 and out.numel() == n
    L.check(L.load().afm_colsum(_ptr(x), _ptr(out), rows, n, _ld(x), _dt(x), int(accumulate), _stream()),
            "afm_colsum")
    return out


def add_inplace(y, x):
    assert y.dtype == torch.float32 and x.dtype == torch.float32 and y.numel() == x.numel()
    L.check(L.load().afm_add_inplace(_ptr(y), _ptr(x), y.numel(), _stream()), "afm_add_inplace")


def batch_sum(x, out, B, S, d, accumulate=True):
    L.check(L.load().afm_batch_sum(_ptr(x), _ptr(out), B, S, d, int(accumulate), _stream()), "afm_batch_sum")


def relu_bwd(dy, act, dx=None):
    """dx = dy * (act > 0), fp32, contiguous (dx defaults to dy: in place)."""
    dx = dy if dx is None else dx
    assert dy.dtype == torch.float32 and act.dtype == torch.float32 and dy.is_contiguous() and act.is_contiguous()
    assert dx.is_contiguous() and dy.numel() == act.numel() == dx.numel()
    L.check(L.load().afm_relu_bwd(_ptr(dy), _ptr(act), _ptr(dx), dy.numel(), _stream()), "afm_relu_bwd")
    return dx


def convert(src, dst):
    """dst = cast(src) between fp32 / bf16 / split-pair 2-D views."""
    rows, n = src.shape
    assert tuple(dst.shape) == (rows, n)
    L.check(L.load().afm_convert(_ptr(src), _dt(src), _ld(src), _ptr(dst), _dt(dst), _ld(dst), rows, n, _stream()),
            "afm_convert")
    return dst


def cast_x2(src, dst=None, dst_t=None):
    rows, cols = src.shape
    assert src.dtype == torch.float32 and src.is_contiguous()
    for t, (r, c) in ((dst, (rows, cols)), (dst_t, (cols, rows))):
        assert t is None or (isinstance(t, X2) and tuple(t.shape) == (r, c) and t.ld == 2 * c)
    L.check(L.load().afm_cast_x2(_ptr(src), _ptr(dst), _ptr(dst_t), rows, cols, _stream()), "afm_cast_x2")


def cast_weights(src, dst=None, dst_t=None, glu_rows: int = 0):
    """fp32 (rows x cols) -> bf16 / fp16 / split-pair shadow `dst` and its transpose `dst_t`, optionally with the gated-FFN row
    interleave (glu_rows = f, rows = 2f)."""
    rows, cols = src.shape
    assert src.dtype == torch.float32 and src.is_contiguous()
    t = dst if dst is not None else dst_t
    assert (dst is None or is_contig(dst)) and (dst_t is None or is_contig(dst_t))
    L.check(L.load().afm_cast_weights(_ptr(src), _ptr(dst), _ptr(dst_t), rows, cols, _dt(t), int(glu_rows), _stream()),
            "afm_cast_weights")


class CastBatch:
    """A fixed list of (src fp32, dst, dst_t, glu_rows) matrices cast by ONE launch (afm_cast_weights_batch).  The item table lives on
    the device and holds raw pointers: the tensors must stay where they are (parameter spans and shadow buffers do)."""

    def __init__(self, entries, dtype):
        import numpy as np
        items = (L.CastItem * len(entries))()
        tile0 = 0
        self._keep = []
        for i, (src, dst, dst_t, glu_rows) in enumerate(entries):
            rows, cols = src.shape
            assert src.dtype == torch.float32 and src.is_contiguous()
            assert (dst is None or is_contig(dst)) and (dst_t is None or is_contig(dst_t)) and (dst is not None or dst_t is not None)
            items[i].src, items[i].dst, items[i].dst_t = _ptr(src), _ptr(dst), _ptr(dst_t)
            items[i].rows, items[i].cols, items[i].glu_rows, items[i].tile0 = rows, cols, int(glu_rows), tile0
            tile0 += ((rows + 63) // 64) * ((cols + 63) // 64)
            self._keep.append((src, dst, dst_t))
        self.n, self.tiles, self.dt = len(entries), tile0, _DT[dtype]
        raw = np.frombuffer(bytes(items), dtype=np.uint8).copy()
        self.table = torch.from_numpy(raw).to(entries[0][0].device)

    def run(self):
        L.check(L.load().afm_cast_weights_batch(_ptr(self.table), self.n, self.tiles, self.dt, _stream()), "afm_cast_weights_batch")


def cast_bf16(src, dst=None, dst_t=None):
    rows, cols = src.shape
    assert src.dtype == torch.float32 and src.is_contiguous()
    L.check(L.load().afm_cast_bf16(_ptr(src), _ptr(dst), _ptr(dst_t), rows, cols, _stream()), "afm_cast_bf16")


ALIGN_KINDS = {"mse": 0, "mae": 1, "sid": 2}


def masked_mean_fwd(x, key_pad, B, S, out):
    """out (B, d) fp32 = mean over the kept rows of x (B*S, d); key_pad (B, S) uint8, 1 = pad."""
    d = x.shape[-1]
    assert is_contig(x) and key_pad.dtype == torch.uint8 and out.dtype == torch.float32
    L.check(L.load().afm_masked_mean_fwd(_ptr(x), _dt(x), _ptr(key_pad), B, S, d, _ptr(out), _stream()), "afm_masked_mean_fwd")


def masked_mean_bwd(dy, key_pad, B, S, dx, accumulate=False):
    d = dy.shape[-1]
    assert dy.dtype == torch.float32 and dx.dtype == torch.float32 and dx.is_contiguous()
    L.check(L.load().afm_masked_mean_bwd(_ptr(dy), _ptr(key_pad), B, S, d, _ptr(dx), int(accumulate), _stream()), "afm_masked_mean_bwd")


def align_loss(z, target, kind: str, grad_scale, stats, dz=None, scale_dev=None):
    B, n = z.shape
    assert z.dtype == torch.float32 and target.dtype == torch.float32 and z.is_contiguous() and target.is_contiguous()
    L.check(L.load().afm_align_loss(_ptr(z), _ptr(target), ALIGN_KINDS[kind], B, n, float(grad_scale), _ptr(scale_dev),
                                    _ptr(stats), _ptr(dz), _stream()), "afm_align_loss")


def mix_spectra(table, idx, ratio, normalize=False, out_len=1800):
    """Weighted mixtures of rows of `table` (N, L) fp32: idx (n, c) int64, ratio (c,) float64 -> (n, out_len) fp32."""
    N, Ln = table.shape
    n, c = idx.shape
    assert table.dtype == torch.float32 and table.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous()
    ratio = torch.as_tensor(ratio, dtype=torch.float64, device=table.device).contiguous()
    assert ratio.numel() == c
    out = torch.empty(n, out_len, dtype=torch.float32, device=table.device)
    L.check(L.load().afm_mix_spectra(_ptr(table), N, Ln, _ptr(idx), n, c, _ptr(ratio), int(normalize), out_len, _ptr(out),
                                     _stream()), "afm_mix_spectra")
    return out


def patch_preprocess(spectra, present, mean, std, patch_size, masking=False, interpolation=False, overlap=1,
                     derivative=False, seq_first=False):
    """PatchPreprocessor.__call__ on the device: spectra (B, L) fp32, present (B,) bool/uint8 or None.
    Returns (patches fp32 (B, P, ps) | (P, B, ps), mask bool, True = pad)."""
    B, Ln = spectra.shape
    assert spectra.dtype == torch.float32 and spectra.is_contiguous()
    d = L.PatchDesc(B=B, L=Ln, patch_size=patch_size, step=patch_size // overlap, interpolation=int(interpolation),
                    derivative=int(derivative), masking=int(masking), seq_first=int(seq_first),
                    mean=float(mean), std=float(std))
    lib = L.load()
    P = lib.afm_patch_count(C.byref(d))
    if P <= 0:
        raise L.AfmError("afm_patch_preprocess: invalid descriptor")
    pr = None if present is None else present.to(torch.uint8).contiguous()
    shape = (P, B, patch_size) if seq_first else (B, P, patch_size)
    patches = torch.empty(shape, dtype=torch.float32, device=spectra.device)
    mask = torch.empty(shape[:2], dtype=torch.uint8, device=spectra.device)
    L.check(lib.afm_patch_preprocess(C.byref(d), _ptr(spectra), _ptr(pr), _ptr(patches), _ptr(mask), _stream()),
            "afm_patch_preprocess")
    return patches, mask.bool()


def beam_desc(B, k, V, cur_len, max_length, eos, pad, stop_rule, logits, seq_in, seq_out, beam_scores, beam_idx,
              hyp_seq, hyp_score, hyp_len, hyp_count, done, n_open) -> "L.BeamDesc":
    d = L.BeamDesc()
    d.B, d.k, d.V, d.ldl = B, k, V, (_ld(logits) if logits is not None else V)
    d.cur_len, d.max_length, d.eos, d.pad, d.stop_rule, d.Lmax = cur_len, max_length, eos, pad, stop_rule, int(seq_in.shape[1])
    assert seq_in.dtype == torch.int64 and seq_in.is_contiguous() and hyp_seq.is_contiguous() and beam_scores.dtype == torch.float32
    assert beam_idx is None or beam_idx.dtype == torch.int32
    d.logits, d.seq_in, d.seq_out, d.beam_scores, d.beam_idx = _ptr(logits), _ptr(seq_in), _ptr(seq_out), _ptr(beam_scores), _ptr(beam_idx)
    d.hyp_seq, d.hyp_score, d.hyp_len, d.hyp_count = _ptr(hyp_seq), _ptr(hyp_score), _ptr(hyp_len), _ptr(hyp_count)
    d.done, d.n_open = _ptr(done), _ptr(n_open)
    d._keep = (logits, seq_in, seq_out, beam_scores, beam_idx, hyp_seq, hyp_score, hyp_len, hyp_count, done, n_open)
    return d


def beam_step(d) -> None:
    L.check(L.load().afm_beam_step(C.byref(d), _stream()), "afm_beam_step")


def beam_finalize(d, out, out_scores, out_len) -> None:
    L.check(L.load().afm_beam_finalize(C.byref(d), _ptr(out), _ptr(out_scores), _ptr(out_len), _stream()), "afm_beam_finalize")


def cache_reorder(src, dst, beam_idx, rows: int, block_bytes: int, used_bytes: int) -> None:
    assert beam_idx.dtype == torch.int32 and beam_idx.numel() == rows
    L.check(L.load().afm_cache_reorder(_ptr(src), _ptr(dst), _ptr(beam_idx), rows, block_bytes, used_bytes, _stream()),
            "afm_cache_reorder")


def ce_fwd(logits, labels, row_lse, argmax, stats):
    rows, V = logits.shape
    assert logits.dtype == torch.float32 and labels.dtype == torch.int64
    L.check(L.load().afm_ce_fwd(_ptr(logits), _ptr(labels), rows, V, _ld(logits), _ptr(row_lse), _ptr(argmax),
                                _ptr(stats), _stream()), "afm_ce_fwd")


def ce_bwd(logits, labels, row_lse, stats, grad_scale, dlogits, scale_dev=None):
    """scale_dev: device float (word 0 of the loss-scaler state) multiplied into the gradient (fp16 mode)."""
    rows, V = logits.shape
    L.check(L.load().afm_ce_bwd(_ptr(logits), _ptr(labels), _ptr(row_lse), _ptr(stats), float(grad_scale), _ptr(scale_dev),
                                _ptr(dlogits), _dt(dlogits), _ld(dlogits), rows, V, _ld(logits), _stream()),
            "afm_ce_bwd")


_SUMSQ_WS = {}


def sumsq(g, out, partial=None):
    """out[0] += sum g^2, bit-reproducible (two stages through a 2048-float workspace, kept per device when not given)."""
    if partial is None:
        partial = _SUMSQ_WS.get(g.device)
        if partial is None:
            partial = _SUMSQ_WS[g.device] = torch.empty(2048, dtype=torch.float32, device=g.device)
    L.check(L.load().afm_sumsq(_ptr(g), g.numel(), _ptr(out), _ptr(partial), _stream()), "afm_sumsq")


def adam_step(p, g, m, v, hyper, sumsq_buf, p_lowp=None, zero_grad=True, scaler=None):
    """p_lowp: flat bf16 / fp16 shadow of p written by the same kernel; scaler: the 4-float loss-scaler state (fp16 mode)."""
    L.check(L.load().afm_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(hyper), _ptr(sumsq_buf),
                                   _ptr(p_lowp), _dt(p_lowp) if p_lowp is not None else AFM_BF16, int(zero_grad),
                                   _ptr(scaler), _stream()), "afm_adam_step")


def scaler_update(state, sumsq_buf, growth=2.0, backoff=0.5, interval=2000):
    """torch.amp.GradScaler.update() on the device (include/afm_hip.h, afm_scaler_update)."""
    assert state.dtype == torch.float32 and state.numel() >= 4
    L.check(L.load().afm_scaler_update(_ptr(state), _ptr(sumsq_buf), float(growth), float(backoff), int(interval), _stream()),
            "afm_scaler_update")
