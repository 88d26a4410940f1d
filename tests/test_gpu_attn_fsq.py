"""GPU: the fused short-query attention backward (csrc/afm_attn_fsq_impl.h; afm_attn_shape.reserved bit 18) against the two general
backward kernels on the same inputs and against a torch fp32 restatement of the attention backward.

The shape is the decoder's cross-attention (reference: custom_modeling.py:312-318 -- nn.MultiheadAttention over the encoder memory with
its key padding mask): Tq <= 128 queries against the padded memory of each sample, dense and packed (afm_compact_plan mode 2) key rows.
delta is the dQ kernel's bit for bit; dQ follows its chain of products (without dropout equal up to the compiler's choice of a fused
scale-and-round for some elements; with dropout the fused kernel keeps dO unrounded, see the kernel's header); dK / dV sum the queries in
another order.  The torch reference takes the dropout mask out of the keep-bit tensor the forward kernel wrote.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
FSQ = 262144


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodalanalytical_amd import ops as _ops
    return _ops


def _keep_from_bits(bits, B, H, Tq, Tk):
    """The keep mask (B, H, Tq, Tk) out of the keep-bit tensor the forward kernel wrote (include/afm_hip.h, afm_attn_shape.drop_bits):
    [b h][32-query block][32-key block][16 words]; word r, bit l <-> query l & 31, key (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the block."""
    nq, nk = ((Tq + 127) // 128) * 4, ((Tk + 63) // 64) * 2
    w = bits[:B * H * nq * nk * 16].view(B * H, nq, nk, 16, 1)
    lane = torch.arange(64, device=bits.device)
    bit = ((w >> lane) & 1).bool()                                   # (BH, nq, nk, 16, 64)
    r = torch.arange(16, device=bits.device)
    key = ((r & 3) + 8 * (r >> 2))[:, None] + 4 * (lane >> 5)[None, :]   # (16, 64)
    qq = (lane & 31)[None, :].expand(16, 64)
    keep = torch.zeros(B * H, nq, nk, 32, 32, dtype=torch.bool, device=bits.device)
    keep[:, :, :, qq, key] = bit
    keep = keep.permute(0, 1, 3, 2, 4).reshape(B, H, nq * 32, nk * 32)
    return keep[:, :, :Tq, :Tk]


def _torch_ref(q, k, v, do, pad, H, scale, keep=None, p_drop=0.0, causal=False):
    """fp32 attention forward + backward, (B, T, H dh) tensors, pad (B, Tk) bool; dropout through a given keep mask."""
    B, Tq, d = q.shape
    Tk = k.shape[1]
    dh = d // H
    qh, kh, vh, doh = (x.float().view(B, -1, H, dh).transpose(1, 2) for x in (q, k, v, do))
    qh.requires_grad_(True); kh.requires_grad_(True); vh.requires_grad_(True)
    s = (qh @ kh.transpose(-1, -2)) * scale
    s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    if causal:
        s = s.masked_fill(torch.ones(Tq, Tk, dtype=torch.bool, device=s.device).triu(1)[None, None], float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)          # a sample with no live key: zeros, as the kernels define it
    if keep is not None:
        p = p * keep / (1.0 - p_drop)
    o = p @ vh
    o.backward(doh)
    back = lambda x: x.transpose(1, 2).reshape(B, -1, d)
    return back(o.detach()), back(qh.grad), back(kh.grad), back(vh.grad)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("Tq,Tk,lens", [(128, 1024, (1024, 700, 130, 64, 5, 0, 333, 960)), (100, 1024, (1024, 2, 63, 65, 512, 200, 999, 128)),
                                        (128, 56, (56, 3, 17, 40, 0, 56, 1, 33)), (37, 200, (200, 64, 129, 7, 0, 199, 128, 100)),
                                        (64, 320, (320, 100, 31)),
                                        # the decoder's self-attention: causal, Tq == Tk, padded tails
                                        (-128, 128, (128, 90, 17, 64, 2, 0, 128, 33)), (-100, 100, (100, 2, 63, 65, 50, 99, 7, 100)), (-38, 38, (38, 5, 30))])
def test_fused_equals_two_kernels(ops, dtype, p, Tq, Tk, lens):
    causal = Tq < 0
    Tq = abs(Tq)
    B, H, dh = len(lens), (8 if len(lens) == 8 else 5), 64          # (3 x 5 heads: not a multiple of 8, the block map's other branch)
    d = H * dh
    g = torch.Generator().manual_seed(3)
    n = torch.tensor(lens)
    pad = (torch.arange(Tk)[None, :] >= n[:, None]).to(torch.uint8).to(DEV)
    q = (torch.randn(B * Tq, d, generator=g) * 0.5).to(DEV).to(dtype)
    kv = (torch.randn(B * Tk, 2 * d, generator=g) * 0.5).to(DEV).to(dtype)
    do = (torch.randn(B * Tq, d, generator=g) * 0.1).to(DEV).to(dtype)
    drop = ops.drop(p, 4, 1) if p else ops.NO_DROP

    def run(flag, bits=True):
        shp = ops.attn_shape(B, H, Tq, Tk, dh, dtype, d, 2 * d, 2 * d, d, pad, causal, drop)
        if p and bits:
            ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=DEV))
        o = torch.full((B * Tq, d), 3.0, dtype=dtype, device=DEV)
        lse = torch.full((B * H * Tq,), 3.0, device=DEV)
        ops.attn_fwd(shp, q, kv[:, :d], kv[:, d:], o, lse)
        if p and bits:
            run.bits = shp._bits_keepalive
        shp.reserved |= flag if flag else 32768          # (baseline: the 32 x 32 x 16 dQ kernel in every case -- the chain the fused kernel runs)
        dq = torch.full((B * Tq, d), 3.0, dtype=dtype, device=DEV)
        dkv = torch.full((B * Tk, 2 * d), float("nan"), dtype=dtype, device=DEV)
        delta = torch.full_like(lse, 7.0)
        ops.attn_bwd(shp, q, kv[:, :d], kv[:, d:], o, do, lse, delta, dq, dkv[:, :d], dkv[:, d:], d, 2 * d, 2 * d)
        assert ops.last_algo() == ("attn_fsq" if flag else "attn_mfma")
        return o, dq, dkv, delta

    o0, dq0, dkv0, dl0 = run(0)
    o1, dq1, dkv1, dl1 = run(FSQ)
    assert torch.equal(dl0, dl1)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    assert bool(torch.isfinite(dkv1.float()).all())
    errs = {}
    for name, a, b in (("dQ", dq0, dq1), ("dK", dkv0[:, :d], dkv1[:, :d]), ("dV", dkv0[:, d:], dkv1[:, d:])):
        errs[name] = float((a.float() - b.float()).abs().max() / a.float().abs().max().clamp_min(1e-6))
    if p:      # the same dropout stream without the keep-bit tensor (the kernels hash again): a third opinion
        _, dq2, dkv2, _ = run(0, bits=False)
        for name, a, b in (("dQ~hash", dq2, dq1), ("dK~hash", dkv2[:, :d], dkv1[:, :d]), ("dV~hash", dkv2[:, d:], dkv1[:, d:])):
            errs[name] = float((a.float() - b.float()).abs().max() / a.float().abs().max().clamp_min(1e-6))
    padded = (pad != 0).view(-1)
    assert float(dkv1[padded].float().abs().max() if bool(padded.any()) else 0.0) == 0.0      # padded keys: exact zeros
    if True:      # (block kept flat for the diff's sake)
        keep = _keep_from_bits(run.bits, B, H, Tq, Tk) if p else None
        ro, rdq, rdk, rdv = _torch_ref(q.view(B, Tq, d), kv[:, :d].reshape(B, Tk, d), kv[:, d:].reshape(B, Tk, d), do.view(B, Tq, d),
                                       pad.bool(), H, dh ** -0.5, keep, p, causal)
        if p:
            seen = torch.ones(Tq, Tk, dtype=torch.bool, device=keep.device).tril() if causal else torch.ones(Tq, Tk, dtype=torch.bool, device=keep.device)
            assert 0.85 < float(keep[0][:, seen].float().mean()) < 0.95          # (sample 0 has no padded key in any of the cases: every block at or below the diagonal was written)
            assert float((o1.view(B, Tq, d).float() - ro).abs().max() / ro.abs().max()) < (4e-3 if dtype == torch.float16 else 3e-2)
        for name, a, b in (("dQ", dq1.view(B, Tq, d), rdq), ("dK", dkv1[:, :d].reshape(B, Tk, d), rdk), ("dV", dkv1[:, d:].reshape(B, Tk, d), rdv)):
            err = float((a.float() - b).abs().max() / b.abs().max())
            errs[name + "~torch"] = err
            assert err < (4e-3 if dtype == torch.float16 else 3e-2), (name, errs)
    # (a sample with ONE live key under dropout is left out of the long-memory case on purpose: its dS cancels to nothing in exact
    # arithmetic, and what every kernel here computes instead is dO . (O - round16(O)) -- delta is taken from the ROUNDED output -- which
    # the dK/dV kernel happens to cancel exactly (it rounds scale * V the same way) and the dQ chain does not: noise of 2^-11 |dP|,
    # 1e-2 of this test's small max |dK|, in the fused kernel as in the dQ kernel's dQ since round 3)
    # (causal rows are the same case in small: the first queries see one or two keys.  Under dropout the two-kernel comparison gets twice the
    # tolerance there; the torch comparison above keeps its bar)
    assert all(e < (2 * tol if (causal and p) else tol) for e in errs.values()), errs
    if not p:
        assert errs["dQ"] < tol / 4, errs


@pytest.mark.parametrize("nofill", [False, True])
@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("lens", [(1024, 700, 130, 128, 5, 0, 333), (1024, 700, 130, 128, 5, 32, 333)])
def test_fused_packed_memory_rows(ops, p, lens, nofill):
    """Packed key rows (k_off): live rows bit-identical to the dense layout's, the dead tail zero-filled through the bijection -- or, with
    the no-fill flag, untouched beyond the 64-row block around the slots' end."""
    B, H, T, dh, Tq = 7, 8, 1024, 64, 128
    d = H * dh
    g = torch.Generator().manual_seed(9)
    n = torch.tensor(lens)
    pad = (torch.arange(T)[None, :] >= n[:, None]).to(torch.uint8).to(DEV)
    plan = ops.compact_plan(pad, B, T, 256, compact=2)
    off, dest = plan.seq_off, plan.dest.long()
    kv_d = (torch.randn(B * T, 2 * d, generator=g) * 0.5).to(DEV).half()
    kv_p = torch.zeros_like(kv_d)
    kv_p[dest] = kv_d
    q = (torch.randn(B * Tq, d, generator=g) * 0.5).to(DEV).half()
    do = (torch.randn(B * Tq, d, generator=g) * 0.1).to(DEV).half()
    drop = ops.drop(p, 4, 1) if p else ops.NO_DROP

    def run(packed):
        kv = kv_p if packed else kv_d
        shp = ops.attn_shape(B, H, Tq, T, dh, torch.float16, d, 2 * d, 2 * d, d, pad, False, drop, **(dict(k_off=off) if packed else {}))
        if p:
            ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, T), dtype=torch.int64, device=DEV))
        o = torch.full((B * Tq, d), 3.0, dtype=torch.float16, device=DEV)
        lse = torch.full((B * H * Tq,), 3.0, device=DEV)
        ops.attn_fwd(shp, q, kv[:, :d], kv[:, d:], o, lse)
        shp.reserved |= FSQ | (131072 if (packed and nofill) else 0)
        dq = torch.full((B * Tq, d), 3.0, dtype=torch.float16, device=DEV)
        dkv = torch.full((B * T, 2 * d), 3.0, dtype=torch.float16, device=DEV)
        delta = torch.empty_like(lse)
        ops.attn_bwd(shp, q, kv[:, :d], kv[:, d:], o, do, lse, delta, dq, dkv[:, :d], dkv[:, d:], d, 2 * d, 2 * d)
        assert ops.last_algo() == "attn_fsq"
        return dq, dkv

    dqd, dkvd = run(False)
    dqp, dkvp = run(True)
    live_k = (pad == 0).view(-1)
    assert torch.equal(dqd, dqp)
    assert torch.equal(dkvd[live_k], dkvp[dest][live_k])
    used = int(off[-1])
    end = -(-used // 64) * 64 if nofill else B * T
    assert float(dkvp[used:end].float().abs().max() if end > used else 0.0) == 0.0
    if nofill and end < B * T:
        assert float((dkvp[end:].float() - 3.0).abs().max()) == 0.0          # not written
