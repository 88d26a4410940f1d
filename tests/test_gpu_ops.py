"""GPU: every C-ABI entry point against the CPU oracle's primitives (oracle/afm_oracle.py) on the
same seeded inputs.  fp32 paths are held to ~1e-5; bf16 storage paths to bf16 rounding."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import afm_oracle as O  # noqa: E402
from tests.dropmask import keep_mask, keep_mask16  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodalanalytical_amd import ops as _ops
    return _ops


DEV = "cuda:0"


def dev(t, dtype=None):
    return t.to(DEV if dtype is None else DEV, dtype=dtype if dtype is not None else t.dtype).contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(got, ref, rtol, atol, msg=""):
    torch.testing.assert_close(got.detach().float().cpu(), ref.float(), rtol=rtol, atol=atol, msg=lambda m: f"{msg}: {m}")


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(80, 26, 64), (37, 130, 75), (256, 192, 128), (5, 7, 3), (300, 64, 2048)])
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_generic_fp32(ops, M, N, K, ta, tb):
    a = rnd(*((K, M) if ta else (M, K)), seed=1)
    b = rnd(*((N, K) if tb else (K, N)), seed=2)
    bias = rnd(N, seed=3)
    ref = (a.T if ta else a).double() @ (b.T if tb else b).double() + bias.double()
    c = torch.empty(M, N, device=DEV)
    ops.gemm(dev(a), dev(b), c, trans_a=ta, trans_b=tb, bias=dev(bias), algo=1)
    close(c, ref, 1e-5, 1e-4 * math.sqrt(K) / 8, f"gemm {M}x{N}x{K}")


def test_gemm_epilogue_and_strides(ops):
    M, N, K = 96, 48, 40
    a, w, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    p, seed, site = 0.25, 1234567890123, 7
    big = torch.zeros(M, N + 16, device=DEV)
    pre = torch.zeros(M, N + 16, device=DEV)
    resd = torch.zeros(M, N + 16, device=DEV); resd[:, :N] = dev(res)
    ops.gemm(dev(a), dev(w), big[:, :N], bias=dev(bias), residual=resd[:, :N], pre_act=pre[:, :N], act=2,
             dropout=ops.drop(p, seed, site), algo=1)
    t = a.double() @ w.double().T + bias.double()
    keep = torch.from_numpy(keep_mask(p, seed, site, M * N)).view(M, N)
    ref = O.gelu(t) * keep / (1 - p) + res.double()
    close(big[:, :N], ref, 1e-5, 1e-5)
    close(pre[:, :N], t, 1e-5, 1e-5)
    assert float(big[:, N:].abs().max()) == 0.0
    assert abs(float(keep.float().mean()) - 0.75) < 0.03
    # accumulate + bf16 operands + fp32 out (the wgrad form)
    dy, x = rnd(200, 24, seed=5), rnd(200, 40, seed=6)
    acc = rnd(24, 40, seed=7)
    c = dev(acc).clone()
    ops.gemm(dev(dy, torch.bfloat16), dev(x, torch.bfloat16), c, trans_a=True, trans_b=False, accumulate=True, algo=1)
    ref = acc.double() + dy.bfloat16().double().T @ x.bfloat16().double()
    close(c, ref, 1e-5, 1e-4)


def test_gemm_splitk(ops):
    M, N, K = 64, 48, 8192
    a, b = rnd(K, M, seed=1, scale=0.1), rnd(K, N, seed=2, scale=0.1)
    c0 = rnd(M, N, seed=3)
    c = dev(c0).clone()
    ops.gemm(dev(a), dev(b), c, trans_a=True, trans_b=False, accumulate=True, algo=1)
    assert ops.last_algo() == "generic_splitk"
    close(c, c0.double() + a.double().T @ b.double(), 1e-4, 1e-4)


# ------------------------------------------------------------------ embedding rows
def test_gather_scatter(ops):
    V, d, n = 45, 64, 300
    table = rnd(V, d, seed=1)
    ids = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(2))
    scale = rnd(n, seed=3)
    out = torch.empty(n, d, device=DEV)
    ops.gather_rows(dev(ids), dev(table), out, dev(scale))
    close(out, table[ids] * scale[:, None], 0, 0)
    dout = rnd(n, d, seed=4)
    dtab = torch.zeros(V, d, device=DEV)
    ops.scatter_add_rows(dev(ids), dev(dout), dtab, dev(scale), padding_idx=0)
    ref = torch.zeros(V, d, dtype=torch.float64)
    keep = ids != 0
    ref.index_add_(0, ids[keep], (dout * scale[:, None])[keep].double())
    close(dtab, ref, 1e-5, 1e-5)


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("d", [48, 64, 512, 768])
@pytest.mark.parametrize("ydt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Sm,S,off", [(3, 5, 12, 4), (8, 12, 20, 7)])     # 15 rows: row-per-wave scalar kernels; 96 rows: the vectorised ones
def test_layernorm_fwd_bwd(ops, d, ydt, B, Sm, S, off):
    rows = B * Sm
    x = rnd(rows, d, seed=1) * 2 + 0.5
    gam, bet = 1 + 0.1 * rnd(d, seed=2), 0.1 * rnd(d, seed=3)
    pos = rnd(S, d, seed=4)
    y = torch.zeros(B * S, d, dtype=ydt, device=DEV)
    mean = torch.empty(rows, device=DEV); rstd = torch.empty(rows, device=DEV)
    ops.layernorm_fwd(dev(x), dev(gam), dev(bet), y, mean, rstd, pos=dev(pos), seg_len=Sm, out_seg_stride=S, out_off=off)
    xr = x.double().requires_grad_(True)
    gr, br = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref_rows = O.layer_norm(xr, gr, br).view(B, Sm, d) + pos.double()[off:off + Sm]
    ref = torch.zeros(B, S, d, dtype=torch.float64)
    ref[:, off:off + Sm] = ref_rows.detach()
    tol = dict(rtol=1e-5, atol=1e-5) if ydt == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    close(y.view(B, S, d), ref, **tol)
    # backward
    dy_full = rnd(B * S, d, seed=5)
    dres = rnd(rows, d, seed=6)
    dy_dev = dev(dy_full, ydt)
    dyr = dy_dev.float().cpu().double().view(B, S, d)[:, off:off + Sm]
    ref_rows.backward(dyr)
    dx = torch.empty(rows, d, device=DEV)
    dg = dev(rnd(d, seed=7)); db = dev(rnd(d, seed=8))
    dg0, db0 = dg.clone().cpu(), db.clone().cpu()
    ws = torch.empty(ops.layernorm_bwd_ws(rows, d), device=DEV)
    ops.layernorm_bwd(dy_dev, dev(x), dev(gam), mean, rstd, dx, dg, db, ws, dres=dev(dres), seg_len=Sm,
                      out_seg_stride=S, out_off=off)
    close(dx, xr.grad + dres.double(), 1e-4, 1e-5)
    # second output: dropout(dx) in the operand dtype (identity row mapping)
    dxd = torch.empty(rows, d, dtype=ydt, device=DEV)
    dx2 = torch.empty(rows, d, device=DEV)
    dy_rows = dev(dy_full.view(B, S, d)[:, off:off + Sm].reshape(rows, d), ydt)
    ops.layernorm_bwd(dy_rows, dev(x), dev(gam), mean, rstd, dx2, dg.clone(), db.clone(), ws, dres=dev(dres),
                      dx_drop=dxd, dropout=ops.drop(0.25, 77, 2))
    keep = torch.from_numpy(keep_mask(0.25, 77, 2, rows * d)).view(rows, d)
    close(dx2, dx.cpu(), 1e-6, 1e-6)
    close(dxd, (dx2.cpu() * keep / 0.75).to(ydt), 1e-2 if ydt == torch.bfloat16 else 1e-6, 1e-6)
    close(dg, dg0.double() + gr.grad, 1e-4, 1e-5)
    close(db, db0.double() + br.grad, 1e-4, 1e-5)


# ------------------------------------------------------------------ attention
def _attn_case(B, H, Tq, Tk, dh, causal, pad, seed=0):
    q, k, v = rnd(B, Tq, H, dh, seed=seed + 1), rnd(B, Tk, H, dh, seed=seed + 2), rnd(B, Tk, H, dh, seed=seed + 3)
    key_pad = None
    if pad:
        key_pad = torch.zeros(B, Tk, dtype=torch.bool)
        for b in range(B):
            key_pad[b, Tk - 1 - (b * 3) % max(1, Tk // 2):] = True
        key_pad[:, 0] = False
    return q, k, v, key_pad


@pytest.mark.parametrize("B,H,Tq,Tk,dh,causal,pad", [
    (2, 4, 20, 20, 16, True, True), (2, 4, 16, 40, 16, False, True), (1, 2, 70, 70, 64, False, True),
    (2, 2, 33, 33, 8, True, False), (1, 3, 5, 130, 32, False, False)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_attention_generic(ops, B, H, Tq, Tk, dh, causal, pad, dt):
    q, k, v, key_pad = _attn_case(B, H, Tq, Tk, dh, causal, pad)
    if dt == torch.bfloat16:
        q, k, v = q.bfloat16().float(), k.bfloat16().float(), v.bfloat16().float()
    D = H * dh
    qd, kd, vd = (dev(t.reshape(t.shape[0], t.shape[1], D).reshape(-1, D), dt) for t in (q, k, v))
    o = torch.empty(B * Tq, D, dtype=dt, device=DEV)
    lse = torch.empty(B * H * Tq, device=DEV)
    kp = None if key_pad is None else dev(key_pad.to(torch.uint8))
    shp = ops.attn_shape(B, H, Tq, Tk, dh, dt, D, D, D, D, kp, causal, algo=1)
    ops.attn_fwd(shp, qd, kd, vd, o, lse)
    qr, kr, vr = (t.double().transpose(1, 2).requires_grad_(True) for t in (q, k, v))
    ref = O.attention(qr, kr, vr, key_pad, causal)           # (B,H,Tq,dh)
    ref_o = ref.transpose(1, 2).reshape(B * Tq, D)
    tol = dict(rtol=1e-5, atol=1e-5) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    close(o, ref_o, **tol)
    do = rnd(B * Tq, D, seed=9)
    if dt == torch.bfloat16:
        do = do.bfloat16().float()
    ref.backward(do.double().view(B, Tq, H, dh).transpose(1, 2))
    dq, dk, dv = (torch.empty(n, D, dtype=dt, device=DEV) for n in (B * Tq, B * Tk, B * Tk))
    delta = torch.empty_like(lse)
    ops.attn_bwd(shp, qd, kd, vd, o, dev(do, dt), lse, delta, dq, dk, dv, D, D, D)
    for got, r, T in ((dq, qr, Tq), (dk, kr, Tk), (dv, vr, Tk)):
        close(got, r.grad.transpose(1, 2).reshape(B * T, D), **(dict(rtol=1e-4, atol=1e-5) if dt == torch.float32 else dict(rtol=3e-2, atol=3e-2)))


def test_attention_all_masked_row_and_dropout(ops):
    B, H, Tq, Tk, dh = 2, 2, 6, 9, 16
    q, k, v, _ = _attn_case(B, H, Tq, Tk, dh, False, False)
    key_pad = torch.zeros(B, Tk, dtype=torch.bool); key_pad[1] = True   # batch 1: every key masked
    D = H * dh
    qd, kd, vd = (dev(t.reshape(-1, D)) for t in (q, k, v))
    o = torch.full((B * Tq, D), 7.0, device=DEV); lse = torch.empty(B * H * Tq, device=DEV)
    p, seed, site = 0.3, 99, 5
    shp = ops.attn_shape(B, H, Tq, Tk, dh, torch.float32, D, D, D, D, dev(key_pad.to(torch.uint8)), False,
                         ops.drop(p, seed, site), algo=1)
    ops.attn_fwd(shp, qd, kd, vd, o, lse)
    assert float(o.view(B, Tq, D)[1].abs().max()) == 0.0           # _safe_softmax zeros
    assert torch.isinf(lse.view(B, H, Tq)[1]).all()
    keep, dscale = keep_mask16(p, seed, site, B * H * Tq * Tk, Tk)
    keep = torch.from_numpy(keep).view(B, H, Tq, Tk)
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.05
    qr, kr, vr = (t.double().transpose(1, 2).requires_grad_(True) for t in (q, k, v))
    s = (qr @ kr.transpose(-1, -2)) / math.sqrt(dh)
    pr = torch.softmax(s, -1) * keep * dscale
    ref = (pr @ vr)
    close(o.view(B, Tq, H, dh)[0], ref[0].transpose(0, 1), 1e-5, 1e-5)
    do = rnd(B * Tq, D, seed=3)
    ref[0].backward(do.double().view(B, Tq, H, dh).transpose(1, 2)[0])
    dq, dk, dv = (torch.empty(n, D, device=DEV) for n in (B * Tq, B * Tk, B * Tk))
    ops.attn_bwd(shp, qd, kd, vd, o, dev(do), lse, torch.empty_like(lse), dq, dk, dv, D, D, D)
    close(dq.view(B, Tq, H, dh)[0], qr.grad[0].transpose(0, 1), 1e-4, 1e-5)
    close(dk.view(B, Tk, H, dh)[0], kr.grad[0].transpose(0, 1), 1e-4, 1e-5)
    close(dv.view(B, Tk, H, dh)[0], vr.grad[0].transpose(0, 1), 1e-4, 1e-5)
    assert float(dq.view(B, Tq, D)[1].abs().max()) == 0.0 and float(dk.view(B, Tk, D)[1].abs().max()) == 0.0


# ------------------------------------------------------------------ GLU, dropout-cast, colsum, casts
@pytest.mark.parametrize("gated", [False, True])
@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_glu(ops, gated, act):
    from multimodalanalytical_amd.lib import ACT_GELU, ACT_RELU
    rows, f = 37, 72
    uv = rnd(rows, 2 * f, seed=1)
    p, seed, site = 0.2, 42, 3
    g = torch.empty(rows, f, device=DEV)
    uvd = dev(uv)
    code = ACT_GELU if act == "gelu" else ACT_RELU
    ops.glu_fwd(uvd[:, :f], uvd[:, f:] if gated else None, g, ops.drop(p, seed, site), act=code)
    ur = uv[:, :f].double().requires_grad_(True); vr = uv[:, f:].double().requires_grad_(True)
    keep = torch.from_numpy(keep_mask(p, seed, site, rows * f)).view(rows, f)
    ref = (O.gelu(ur) if act == "gelu" else torch.relu(ur)) * (vr if gated else 1.0) * keep / (1 - p)
    close(g, ref, 1e-5, 1e-6)
    dg = rnd(rows, f, seed=2)
    ref.backward(dg.double())
    duv = torch.zeros(rows, 2 * f, device=DEV)
    ops.glu_bwd(uvd[:, :f], uvd[:, f:] if gated else None, dev(dg), duv[:, :f], duv[:, f:] if gated else None,
                ops.drop(p, seed, site), act=code)
    close(duv[:, :f], ur.grad, 1e-4, 1e-6)
    if gated:
        close(duv[:, f:], vr.grad, 1e-4, 1e-6)


def test_dropout_cast_colsum_casts(ops):
    rows, n = 50, 36
    x = rnd(rows, n, seed=1)
    y = torch.empty(rows, n, dtype=torch.bfloat16, device=DEV)
    ops.dropout_cast(dev(x), y, ops.drop(0.5, 5, 9))
    keep = torch.from_numpy(keep_mask(0.5, 5, 9, rows * n)).view(rows, n)
    close(y, (x * keep * 2).bfloat16(), 0, 0)
    out = dev(rnd(n, seed=2)); out0 = out.cpu().clone()
    ops.colsum(dev(x), out, accumulate=True)
    close(out, out0.double() + x.double().sum(0), 1e-5, 1e-5)
    big = rnd(5000, n, seed=3)
    ops.colsum(dev(big, torch.bfloat16), out, accumulate=False)
    close(out, big.bfloat16().double().sum(0), 1e-4, 1e-3)
    w = rnd(70, 130, seed=4)
    d1 = torch.empty(70, 130, dtype=torch.bfloat16, device=DEV); d2 = torch.empty(130, 70, dtype=torch.bfloat16, device=DEV)
    ops.cast_bf16(dev(w), d1, d2)
    close(d1, w.bfloat16(), 0, 0); close(d2, w.bfloat16().T, 0, 0)
    xs = rnd(4 * 6, 10, seed=5); o = torch.zeros(6, 10, device=DEV)
    ops.batch_sum(dev(xs), o, 4, 6, 10, accumulate=False)
    close(o, xs.view(4, 6, 10).double().sum(0), 1e-6, 1e-6)
    ops.add_inplace(o, o.clone())
    close(o, 2 * xs.view(4, 6, 10).double().sum(0), 1e-6, 1e-6)


# ------------------------------------------------------------------ loss, optimiser
@pytest.mark.parametrize("V", [26, 128, 300])
def test_cross_entropy(ops, V):
    rows = 90
    logits = rnd(rows, V, seed=1) * 3
    logits[3, 5] = logits[3, 9] = logits[3].max() + 1.0   # a tie: first index wins
    labels = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(2))
    labels[::4] = -100
    lse = torch.empty(rows, device=DEV); am = torch.empty(rows, dtype=torch.int64, device=DEV)
    stats = torch.zeros(2, device=DEV)
    ld = dev(logits)
    ops.ce_fwd(ld, dev(labels), lse, am, stats)
    lr = logits.double().requires_grad_(True)
    ref = O.cross_entropy(lr, labels)
    close(stats[0] / stats[1], ref.detach(), 1e-5, 1e-6)
    assert int(stats[1]) == int((labels != -100).sum())
    assert torch.equal(am.cpu(), logits.argmax(-1))
    (ref * 0.25).backward()
    Vp = (V + 7) // 8 * 8
    dl = torch.full((rows, Vp), 5.0, device=DEV)
    ops.ce_bwd(ld, dev(labels), lse, stats, 0.25, dl)
    close(dl[:, :V], lr.grad, 1e-4, 1e-7)
    assert Vp == V or float(dl[:, V:].abs().max()) == 0.0


def test_adam_against_oracle(ops):
    n = 1000 + 3
    p, g = rnd(n, seed=1), rnd(n, seed=2) * 3
    for decoupled in (0.0, 1.0):
        pd, gd = dev(p).clone(), dev(g).clone()
        m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        pr, mr, vr = p.clone().double(), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
        shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
        for t in (1, 2, 3):
            lr, b1 = O.onecycle(t - 1, 10, 1e-2)
            gt = g.double() * t
            gd.copy_(dev(g) * t)
            norm = float(gt.norm())
            coef = min(1.0, 1.0 / (norm + 1e-6))
            O.adam_step(pr, gt * coef, mr, vr, t, lr, b1, 0.999, 1e-8, 0.01, bool(decoupled))
            hyper = torch.tensor([lr, b1, 0.999, 1e-8, 0.01, 1 - b1 ** t, 1 - 0.999 ** t, 1.0, 1.0, decoupled], device=DEV)
            ss = torch.zeros(1, device=DEV)
            ops.sumsq(gd, ss)
            close(ss[0].sqrt(), torch.tensor(norm), 1e-5, 1e-5)
            ops.adam_step(pd, gd, m, v, hyper, ss, shadow, zero_grad=True)
            close(pd, pr, 1e-5, 1e-6, f"adam t={t}")
            assert float(gd.abs().max()) == 0.0
            close(shadow, pd.float().cpu().bfloat16(), 0, 0)


@pytest.mark.parametrize("n", [3, 1000 + 3, 45_000_000 + 1])
def test_sumsq_is_bit_reproducible(ops, n):
    # data-parallel replicas derive their clip coefficient from this number: it has to be the same bits on every rank and every run
    g = torch.randn(n, device=DEV) * 0.3
    got = []
    for _ in range(6):
        ss = torch.zeros(1, device=DEV)
        ops.sumsq(g, ss)
        got.append(ss.clone())
    assert all(torch.equal(got[0], x) for x in got[1:])
    close(got[0][0], g.double().pow(2).sum().float().cpu(), 2e-6, 0)
    ss = torch.full((1,), 2.0, device=DEV)          # accumulates into out[0]
    ops.sumsq(g, ss)
    close(ss[0] - 2.0, got[0][0].cpu(), 1e-5, 1e-5 * n)


# ------------------------------------------------------------------ MFMA GEMMs (bf16 operands, fp32 accumulate)
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (100, 200, 72), (1000, 1536, 512), (384, 64, 2048), (129, 24, 64)])
@pytest.mark.parametrize("cdt", [torch.bfloat16, torch.float32])
def test_gemm_mfma_nt(ops, M, N, K, cdt):
    a, w, bias = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2).bfloat16(), rnd(N, seed=3)
    c = torch.empty(M, N, dtype=cdt, device=DEV)
    ops.gemm(dev(a), dev(w), c, bias=dev(bias), algo=2)
    assert ops.last_algo() == "mfma_nt"
    ref = a.double() @ w.double().T + bias.double()
    tol = dict(rtol=1e-2, atol=1e-2 * math.sqrt(K) / 4) if cdt == torch.bfloat16 else dict(rtol=1e-4, atol=2e-4 * math.sqrt(K) / 8)
    close(c, ref, **tol)


def test_gemm_mfma_nt_identity_asymmetric(ops):
    # A = I with an asymmetric B: catches a transposed C write or swapped fragment maps exactly
    n = 128
    a = torch.eye(n).bfloat16()
    w = (torch.arange(n * n).view(n, n) % 251 - 125).float().bfloat16()
    c = torch.empty(n, n, device=DEV)
    ops.gemm(dev(a), dev(w), c, algo=2)
    close(c, w.float().T, 0, 0)


def test_gemm_mfma_nt_epilogue(ops):
    M, N, K = 200, 136, 96
    a, w, bias, res = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2).bfloat16(), rnd(N, seed=3), rnd(M, N, seed=4)
    p, seed, site = 0.1, 77, 11
    c = torch.empty(M, N, device=DEV); pre = torch.empty(M, N, device=DEV)
    ops.gemm(dev(a), dev(w), c, bias=dev(bias), residual=dev(res), pre_act=pre, act=2, dropout=ops.drop(p, seed, site), algo=2)
    t = a.double() @ w.double().T + bias.double()
    keep = torch.from_numpy(keep_mask(p, seed, site, M * N)).view(M, N)
    close(pre, t, 1e-4, 1e-4)
    close(c, O.gelu(t) * keep / (1 - p) + res.double(), 1e-4, 1e-4)
    # strided operands / output (the packed-projection slices) and accumulate
    big_w = rnd(3 * N, K, seed=5).bfloat16(); wd = dev(big_w)
    out = torch.zeros(M, 2 * N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(dev(a), wd[N:2 * N], out[:, N:], algo=2)
    close(out[:, N:], a.double() @ big_w[N:2 * N].double().T, 1e-2, 3e-2)
    assert float(out[:, :N].abs().max()) == 0.0
    acc0 = rnd(M, N, seed=6); cacc = dev(acc0).clone()
    ops.gemm(dev(a), dev(w), cacc, accumulate=True, algo=2)
    close(cacc, acc0.double() + a.double() @ w.double().T, 1e-4, 1e-4)


@pytest.mark.parametrize("R,M,N", [(512, 128, 128), (1000, 192, 64), (4096, 1536, 512), (777, 24, 64), (8192, 128, 2048), (16384, 520, 200), (4096, 256, 128)])
def test_gemm_mfma_tn_wgrad(ops, R, M, N):
    dy, x = rnd(R, M, seed=1).bfloat16(), rnd(R, N, seed=2).bfloat16()
    g0 = rnd(M, N, seed=3)
    g = dev(g0).clone()
    b0 = rnd(M, seed=4); gb = dev(b0).clone()
    ops.gemm(dev(dy), dev(x), g, trans_a=True, trans_b=False, accumulate=True, algo=2, a_colsum=gb)
    assert ops.last_algo().startswith("mfma_tn")
    close(g, g0.double() + dy.double().T @ x.double(), 1e-4, 2e-4 * math.sqrt(R) / 4)
    close(gb, b0.double() + dy.double().sum(0), 1e-4, 2e-4 * math.sqrt(R) / 4, "fused bias gradient")
    gb2 = dev(b0).clone()
    ops.gemm(dev(dy), dev(x), g.clone(), trans_a=True, trans_b=False, accumulate=True, algo=1, a_colsum=gb2)
    close(gb2, b0.double() + dy.double().sum(0), 1e-4, 2e-4 * math.sqrt(R) / 4, "generic bias gradient")
    g2 = torch.full((M, N), 3.0, device=DEV)
    ops.gemm(dev(dy), dev(x), g2, trans_a=True, trans_b=False, accumulate=False, algo=2)
    close(g2, dy.double().T @ x.double(), 1e-4, 2e-4 * math.sqrt(R) / 4)


def test_gemm_mfma_tn_identity(ops):
    R = 128
    dy = torch.eye(R).bfloat16()                                    # dy^T x = x
    x = (torch.arange(R * 64).view(R, 64) % 241 - 120).float().bfloat16()
    g = torch.zeros(R, 64, device=DEV)
    ops.gemm(dev(dy), dev(x), g, trans_a=True, trans_b=False, algo=2)
    close(g, x.float(), 0, 0)


# ------------------------------------------------------------------ MFMA attention (head size 64)
def _attn_ref(q, k, v, key_pad, causal, keep=None, dscale=1.0):
    """fp64 reference incl. the dropout mask; q,k,v (B,T,H,dh) fp32 (already bf16-rounded)."""
    qr, kr, vr = (t.double().transpose(1, 2).requires_grad_(True) for t in (q, k, v))
    dh = q.shape[-1]
    s = (qr @ kr.transpose(-1, -2)) / math.sqrt(dh)
    masked = torch.zeros(s.shape, dtype=torch.bool)
    if key_pad is not None:
        masked |= key_pad[:, None, None, :]
    if causal:
        masked |= torch.ones(s.shape[-2], s.shape[-1], dtype=torch.bool).triu(1)
    p = torch.softmax(s.masked_fill(masked, -1e30), -1).masked_fill(masked, 0.0)
    if keep is not None:
        p = p * keep * dscale
    return qr, kr, vr, p @ vr


@pytest.mark.parametrize("B,H,Tq,Tk,causal,pad,pdrop", [
    (2, 2, 128, 128, True, True, 0.0), (2, 3, 100, 200, False, True, 0.0), (1, 2, 300, 300, False, False, 0.0),
    (1, 2, 192, 192, True, False, 0.0), (2, 2, 128, 256, False, True, 0.1), (1, 2, 160, 160, True, True, 0.1),
    (2, 2, 130, 520, False, "blocks", 0.1), (2, 2, 256, 56, False, True, 0.1), (2, 2, 56, 56, False, True, 0.1),
    (1, 2, 320, 320, True, True, 0.1), (2, 2, 512, 1024, False, "blocks", 0.1)])
def test_attention_mfma(ops, B, H, Tq, Tk, causal, pad, pdrop):
    dh, dt = 64, torch.bfloat16
    q, k, v, key_pad = _attn_case(B, H, Tq, Tk, dh, causal, bool(pad), seed=10)
    if pad == "blocks":   # whole 64-key tiles (and a whole 32-key wave block) of padding, as in multimodal batches
        key_pad = torch.zeros(B, Tk, dtype=torch.bool)
        key_pad[0, 64:192] = True; key_pad[0, 300:] = True
        key_pad[1, 128:160] = True; key_pad[1, 448:512] = True
    q, k, v = q.bfloat16().float(), k.bfloat16().float(), v.bfloat16().float()
    D = H * dh
    qd, kd, vd = (dev(t.reshape(-1, D), dt) for t in (q, k, v))
    kp = None if key_pad is None else dev(key_pad.to(torch.uint8))
    seed, site = 4242, 3
    keep, dscale = None, 1.0
    if pdrop > 0:
        km, dscale = keep_mask16(pdrop, seed, site, B * H * Tq * Tk, Tk)
        keep = torch.from_numpy(km).view(B, H, Tq, Tk)
    res = {}
    for algo in (1, 2):
        o = torch.empty(B * Tq, D, dtype=dt, device=DEV); lse = torch.empty(B * H * Tq, device=DEV)
        shp = ops.attn_shape(B, H, Tq, Tk, dh, dt, D, D, D, D, kp, causal, ops.drop(pdrop, seed, site), algo=algo)
        if Tq in (256, 320, 512, 1024):      # these cases run the 8-wave staggered forward / dQ kernels (opt-in forms), Tq = 300 the default ones
            shp.reserved = 16
        guard = None
        if pdrop > 0 and algo == 2 and Tq % 2 == 0:   # half of the dropout cases run with the keep-bit tensor, half re-hash
            nb = ops.attn_drop_bits_words(B, H, Tq, Tk)
            guard = torch.full((nb + 256,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64, device=DEV)   # guard words behind the tensor
            guard[:nb] = 0
            ops.attn_set_drop_bits(shp, guard)
        ops.attn_fwd(shp, qd, kd, vd, o, lse)
        assert ops.last_algo() == ("attn_generic" if algo == 1 else "attn_mfma")
        if guard is not None:
            assert bool((guard[nb:] == 0x5A5A5A5A5A5A5A5A).all())          # the scalar stores stay inside the tensor
        res[algo] = (o, lse, shp)
    qr, kr, vr, ref = _attn_ref(q, k, v, key_pad, causal, keep, dscale)
    ref_o = ref.transpose(1, 2).reshape(B * Tq, D)
    close(res[2][0], ref_o, 2e-2, 2e-2, "mfma fwd vs oracle")
    close(res[2][1], res[1][1].cpu(), 4e-3, 4e-3, "lse mfma vs generic")      # (Q * scale * log2e is rounded to bf16 once more in the 8-wave kernel)
    do = rnd(B * Tq, D, seed=9).bfloat16().float()
    ref.backward(do.double().view(B, Tq, H, dh).transpose(1, 2))
    o, lse, shp = res[2]
    dq, dk, dv = (torch.empty(n, D, dtype=dt, device=DEV) for n in (B * Tq, B * Tk, B * Tk))
    ops.attn_bwd(shp, qd, kd, vd, o, dev(do, dt), lse, torch.empty_like(lse), dq, dk, dv, D, D, D)
    assert ops.last_algo() == "attn_mfma"
    for name, got, r, T in (("dq", dq, qr, Tq), ("dk", dk, kr, Tk), ("dv", dv, vr, Tk)):
        want = r.grad.transpose(1, 2).reshape(B * T, D)
        err = float((got.float().cpu().double() - want).norm() / want.norm())
        assert err < 2e-2, (name, err)
        close(got, want, 5e-2, 5e-2 * float(want.abs().max()), name)


def test_attention_mfma_strided_packed_qkv(ops):
    # q/k/v addressed in place inside a packed (rows, 3d) projection, as the engine does
    B, H, T, dh = 2, 8, 128, 64
    d = H * dh
    qkv = rnd(B * T, 3 * d, seed=5).bfloat16()
    qkvd = dev(qkv)
    o = torch.empty(B * T, d, dtype=torch.bfloat16, device=DEV); lse = torch.empty(B * H * T, device=DEV)
    shp = ops.attn_shape(B, H, T, T, dh, torch.bfloat16, 3 * d, 3 * d, 3 * d, d, None, False, algo=2)
    ops.attn_fwd(shp, qkvd[:, :d], qkvd[:, d:2 * d], qkvd[:, 2 * d:], o, lse)
    q, k, v = (qkv[:, i * d:(i + 1) * d].float().view(B, T, H, dh) for i in range(3))
    _, _, _, ref = _attn_ref(q, k, v, None, False)
    close(o, ref.transpose(1, 2).reshape(B * T, d), 2e-2, 2e-2)


@pytest.mark.parametrize("variant", [12, 13, 22, 24, 25, 100])
@pytest.mark.parametrize("M,N,K", [(512, 256, 512), (1000, 1536, 512), (300, 136, 2048), (129, 24, 64), (70000, 384, 128)])
def test_gemm_mfma_nt_ring_variants(ops, variant, M, N, K):
    a, w, bias = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2).bfloat16(), rnd(N, seed=3)
    c = torch.empty(M, N, device=DEV)
    ops.gemm(dev(a), dev(w), c, bias=dev(bias), algo=2, variant=variant)
    close(c, a.double() @ w.double().T + bias.double(), 1e-4, 2e-4 * math.sqrt(K) / 8)


@pytest.mark.parametrize("algo,dt", [(1, torch.float32), (2, torch.bfloat16)])
def test_gemm_gelu_bwd_epilogue(ops, algo, dt):
    """du = dropout'(dy W2) * gelu'(u) fused in the dgrad epilogue == separate glu_bwd."""
    M, f, d = 384, 256, 128
    dy, w2t, u = rnd(M, d, seed=1), rnd(f, d, seed=2), rnd(M, f, seed=3)
    if dt == torch.bfloat16:
        dy, w2t, u = dy.bfloat16().float(), w2t.bfloat16().float(), u.bfloat16().float()
    p, seed, site = 0.1, 5, 6
    du = torch.empty(M, f, dtype=dt, device=DEV)
    ops.gemm(dev(dy, dt), dev(w2t, dt), du, act=3, pre_act=dev(u, dt), dropout=ops.drop(p, seed, site), algo=algo)
    keep = torch.from_numpy(keep_mask(p, seed, site, M * f)).view(M, f)
    ur = u.double().requires_grad_(True)
    O.gelu(ur).backward(torch.ones(M, f, dtype=torch.float64))
    ref = (dy.double() @ w2t.double().T) * keep / (1 - p) * ur.grad
    tol = dict(rtol=1e-4, atol=1e-4) if dt == torch.float32 else dict(rtol=2e-2, atol=5e-2)
    close(du, ref, **tol)


@pytest.mark.parametrize("adt", [torch.float32, torch.bfloat16])
def test_layernorm_fused_residual_add(ops, adt):
    rows, d = 37, 512
    x, br = rnd(rows, d, seed=1), rnd(rows, d, seed=2)
    gam, bet = 1 + 0.1 * rnd(d, seed=3), 0.1 * rnd(d, seed=4)
    brd = dev(br, adt)
    y = torch.empty(rows, d, dtype=torch.bfloat16, device=DEV); xs = torch.empty(rows, d, device=DEV)
    mean = torch.empty(rows, device=DEV); rstd = torch.empty(rows, device=DEV)
    xd = dev(x)
    ops.layernorm_fwd(xd, dev(gam), dev(bet), y, mean, rstd, add=brd, x_sum=xs)
    ref_sum = x + brd.float().cpu()
    close(xs, ref_sum, 0, 0)
    assert torch.equal(xd.cpu(), x)                       # the input stream is left untouched
    close(y, O.layer_norm(ref_sum.double(), gam.double(), bet.double()), 1e-2, 1e-2)


@pytest.mark.parametrize("variant", [12, 24, 25])
@pytest.mark.parametrize("cdt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("mode", ["bias", "gelu_preact_drop", "gelu_bwd", "residual_acc", "gelu_save_grad_pair"])
def test_gemm_persistent_staged_epilogue(ops, cdt, mode, variant):
    """Full tiles (256x128) + edge tiles through the persistent kernel's LDS-staged epilogue."""
    M, N, K = 1024 + 77, 512 + 40, 128
    a, w, bias = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2).bfloat16(), rnd(N, seed=3)
    t = a.double() @ w.double().T
    c = torch.zeros(M, N, dtype=cdt, device=DEV)
    p, seed, site = 0.1, 31, 9
    keep = torch.from_numpy(keep_mask(p, seed, site, M * N)).view(M, N)
    tol = dict(rtol=2e-2, atol=3e-2) if cdt == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    if mode == "bias":
        ops.gemm(dev(a), dev(w), c, bias=dev(bias), algo=2, variant=variant)
        close(c, t + bias.double(), **tol)
    elif mode == "gelu_preact_drop":
        pre = torch.zeros_like(c)
        ops.gemm(dev(a), dev(w), c, bias=dev(bias), act=2, pre_act=pre, dropout=ops.drop(p, seed, site), algo=2, variant=variant)
        close(pre, t + bias.double(), **tol)
        close(c, O.gelu(t + bias.double()) * keep / (1 - p), **tol)
    elif mode == "gelu_bwd":
        u = rnd(M, N, seed=4)
        ud = dev(u, cdt)
        ops.gemm(dev(a), dev(w), c, act=3, pre_act=ud, dropout=ops.drop(p, seed, site), algo=2, variant=variant)
        ur = ud.float().cpu().double().requires_grad_(True)
        O.gelu(ur).backward(torch.ones(M, N, dtype=torch.float64))
        close(c, t * keep / (1 - p) * ur.grad, **tol)
    elif mode == "gelu_save_grad_pair":
        # forward stores keep*scale*gelu'(T) (act 4); the dgrad multiplies by it (act 5): same values as act 2 / act 3
        if variant != 24 and not (variant == 25 and cdt == torch.float32):
            pytest.skip("only the loader-wave kernel (and the FMA kernel) know this epilogue pair")
        alg = 2 if variant == 24 and cdt == torch.bfloat16 else 1
        if alg == 2:     # the MFMA form exists for whole tiles only (partial tiles: FMA kernel, refused under algo=2)
            with pytest.raises(Exception):
                ops.gemm(dev(a), dev(w), c, bias=dev(bias), act=4, pre_act=torch.zeros_like(c), algo=2, variant=24)
            M, N = 1024, 512
            a, w, bias = a[:M], w[:N], bias[:N]
            t = a.double() @ w.double().T
            c = torch.zeros(M, N, dtype=cdt, device=DEV)
            keep = torch.from_numpy(keep_mask(p, seed, site, M * N)).view(M, N)
        aa, ww = (dev(a), dev(w)) if alg == 2 else (dev(a).float(), dev(w).float())
        gp = torch.zeros_like(c)
        ops.gemm(aa, ww, c, bias=dev(bias), act=4, pre_act=gp, dropout=ops.drop(p, seed, site), algo=alg, variant=variant if alg == 2 else 0)
        tr = (t + bias.double()).requires_grad_(True)
        O.gelu(tr).backward(torch.ones(M, N, dtype=torch.float64))
        close(c, O.gelu(tr.detach()) * keep / (1 - p), **tol)
        close(gp, tr.grad * keep / (1 - p), **tol)
        c2 = torch.zeros_like(c)
        ops.gemm(aa, ww, c2, act=5, pre_act=gp, algo=alg, variant=variant if alg == 2 else 0)
        close(c2, t * gp.float().cpu().double(), **tol)
    else:
        if cdt == torch.bfloat16:
            pytest.skip("bf16 residual/accumulate take the fragment epilogue (covered elsewhere)")
        res, c0 = rnd(M, N, seed=5), rnd(M, N, seed=6)
        c.copy_(dev(c0))
        ops.gemm(dev(a), dev(w), c, bias=dev(bias), residual=dev(res), accumulate=True, algo=2, variant=variant)
        close(c, t + bias.double() + res.double() + c0.double(), **tol)


# ---------------------------------------------------------------- input path
def _patch_golden():
    import json
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "patches.npz"))
    return g, json.loads(bytes(g["meta"]).decode())


@pytest.mark.parametrize("seq_first", [False, True])
def test_patch_preprocess_matches_reference_goldens(ops, seq_first):
    """afm_patch_preprocess vs the reference's PatchPreprocessor outputs (tests/golden/patches.npz): bit-exact."""
    g, meta = _patch_golden()
    for name, kw in meta.items():
        sp = torch.from_numpy(g[f"{name}/spectra"]).to(DEV)
        pr = torch.from_numpy(g[f"{name}/present"]).to(DEV)
        p, m = ops.patch_preprocess(sp, pr, kw["mean"], kw["std"], kw["patch_size"], masking=kw["masking"],
                                    interpolation=kw["interpolation"], overlap=kw.get("overlap", 1),
                                    derivative=kw.get("derivative", False), seq_first=seq_first)
        if seq_first:
            p, m = p.transpose(0, 1), m.transpose(0, 1)
        assert torch.equal(p.cpu(), torch.from_numpy(g[f"{name}/patches"])), name
        assert torch.equal(m.cpu(), torch.from_numpy(g[f"{name}/mask"])), name


@pytest.mark.parametrize("B,Ln,ps,overlap,interp,deriv", [(128, 1984, 2, 1, False, False), (33, 1800, 75, 1, True, True),
                                                          (7, 1791, 125, 1, True, False), (5, 1000, 64, 4, False, True),
                                                          (1, 130, 130, 1, False, False)])
def test_patch_preprocess_vs_oracle_shapes(ops, B, Ln, ps, overlap, interp, deriv):
    from oracle import afm_oracle as Orc
    rng = np.random.default_rng(B * 1000 + Ln)
    sp = np.abs(rng.standard_normal((B, Ln))).astype(np.float32)
    pr = rng.random(B) > 0.2
    mean, std = 0.7312345678, 0.5923456789
    ref_p, ref_m = Orc.patch_preprocess(sp, pr, mean, std, ps, False, interp, overlap, deriv)
    p, m = ops.patch_preprocess(torch.from_numpy(sp).to(DEV), torch.from_numpy(pr).to(DEV), mean, std, ps,
                                interpolation=interp, overlap=overlap, derivative=deriv)
    assert torch.equal(p.cpu(), torch.from_numpy(ref_p)) and torch.equal(m.cpu(), torch.from_numpy(ref_m))


def test_patch_preprocessor_class_mirrors_reference_interface(ops):
    """Host mirror (multimodalanalytical_amd.preprocess.PatchPreprocessor): list-of-lists with None in, same outputs."""
    from multimodalanalytical_amd.preprocess import PatchPreprocessor
    g, meta = _patch_golden()
    kw = meta["ps125"]
    pp = PatchPreprocessor(patch_size=125, masking=False, interpolation=False)
    pp.mean, pp.std = kw["mean"], kw["std"]
    rows = [r.astype(np.float64).tolist() if ok else None for r, ok in zip(g["ps125/spectra"], g["ps125/present"])]
    p, m = pp(rows)
    assert torch.equal(p.cpu(), torch.from_numpy(g["ps125/patches"])) and torch.equal(m.cpu(), torch.from_numpy(g["ps125/mask"]))
    fit = {"IR": [[0.0, 1.0, 3.0], [2.0, 0.0, 0.0]]}
    pp.initialise(fit, "IR")
    assert pp.mean == 2.0 and abs(pp.std - np.std([1.0, 3.0, 2.0])) < 1e-12      # non-zero entries only (patches.py:37-39)
    with pytest.raises(Exception):
        ops.patch_preprocess(torch.zeros(2, 100, device=DEV), None, 0.0, 0.0, 10)   # std == 0: refused, nothing launched


def test_device_collator_batch_dict(ops):
    """DeviceCollator emits the reference collator's sequence-first batch dict (datamodules.py:201-218)."""
    from multimodalanalytical_amd.preprocess import DeviceCollator, PatchPreprocessor
    from oracle import afm_oracle as Orc
    rng = np.random.default_rng(11)
    B, S, T, Ln = 5, 9, 12, 1800
    dc = {"Formula": {"type": "text", "vocab_size": 45, "pad_token_id": 0, "target": False},
          "IR": {"type": "1D_patches", "target": False},
          "Smiles": {"type": "text", "vocab_size": 26, "pad_token_id": 0, "target": True}}
    lens = rng.integers(3, S + 1, B); tl = rng.integers(4, T + 2, B)
    f_ids = np.zeros((B, S), dtype=np.int64); f_att = np.zeros((B, S), dtype=np.int64)
    t_ids = np.zeros((B, T + 1), dtype=np.int64); t_att = np.zeros((B, T + 1), dtype=np.int64)
    for b in range(B):
        f_ids[b, :lens[b]] = rng.integers(4, 45, lens[b]); f_att[b, :lens[b]] = 1
        t_ids[b, :tl[b]] = rng.integers(4, 26, tl[b]); t_att[b, :tl[b]] = 1
    sp = np.abs(rng.standard_normal((B, Ln))).astype(np.float32)
    pr = np.array([True, True, False, True, True])
    pp = PatchPreprocessor(patch_size=125, masking=False, interpolation=False)
    pp.mean, pp.std = 0.81, 0.6
    col = DeviceCollator(dc, {"IR": pp}, "Smiles")
    d = lambda a: torch.from_numpy(a).to(DEV)
    out = col({"Formula": {"input_ids": d(f_ids), "attention_mask": d(f_att)},
               "IR": {"spectra": d(sp), "present": d(pr)},
               "Smiles": {"input_ids": d(t_ids), "attention_mask": d(t_att)}})
    ref_p, ref_m = Orc.patch_preprocess(sp, pr, pp.mean, pp.std, 125)
    assert torch.equal(out["encoder_input"]["Formula"].cpu(), torch.from_numpy(f_ids.T))
    assert torch.equal(out["encoder_input"]["IR"].cpu(), torch.from_numpy(ref_p).transpose(0, 1))
    assert torch.equal(out["encoder_pad_mask"].cpu(), torch.from_numpy(np.concatenate([f_att.T == 0, ref_m.T], 0)))
    assert torch.equal(out["decoder_input"]["Smiles"].cpu(), torch.from_numpy(t_ids.T[:-1]))
    assert torch.equal(out["target"].cpu(), torch.from_numpy(t_ids.T[1:]))
    assert torch.equal(out["decoder_pad_mask"].cpu(), torch.from_numpy(t_att.T[:-1] == 0))
    assert torch.equal(out["target_mask"].cpu(), torch.from_numpy(t_att.T[1:] == 0))


# ---------------------------------------------------------------- alignment head
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_masked_mean_fwd_bwd(ops, dt):
    B, S, d = 5, 77, 96
    x = rnd(B * S, d, seed=1)
    if dt == torch.bfloat16:
        x = x.bfloat16().float()
    pad = torch.rand(B, S, generator=torch.Generator().manual_seed(2)) < 0.3
    pad[:, 0] = False
    keep = (~pad).double().unsqueeze(-1)
    ref = (x.double().view(B, S, d) * keep).sum(1) / keep.sum(1)
    out = torch.empty(B, d, device=DEV)
    ops.masked_mean_fwd(dev(x, dt), pad.to(torch.uint8).to(DEV), B, S, out)
    close(out, ref, 1e-5, 1e-6)
    dy = rnd(B, d, seed=3)
    dx = torch.full((B * S, d), 7.0, device=DEV)
    ops.masked_mean_bwd(dev(dy), pad.to(torch.uint8).to(DEV), B, S, dx, accumulate=False)
    refdx = (dy.double().unsqueeze(1) * keep / keep.sum(1, keepdim=True)).view(B * S, d)
    close(dx, refdx, 1e-6, 1e-7)
    ops.masked_mean_bwd(dev(dy), pad.to(torch.uint8).to(DEV), B, S, dx, accumulate=True)
    close(dx, 2 * refdx, 1e-6, 1e-7)


@pytest.mark.parametrize("kind", ["mse", "mae", "sid"])
def test_align_loss_and_gradient(ops, kind):
    """sigmoid + loss (nn.MSELoss / nn.L1Loss / the reference's own kl_div pair, modeling/utils.py:8-22) and d/dz."""
    B, n = 6, 50
    z = rnd(B, n, seed=1)
    t = torch.rand(B, n, generator=torch.Generator().manual_seed(2))
    t[:, ::7] = 0.0
    zr = z.double().requires_grad_(True)
    acfg = {"align_network": "mlp", "loss_function": kind}
    p = torch.sigmoid(zr)
    if kind == "mse": loss = ((p - t.double()) ** 2).mean()
    elif kind == "mae": loss = (p - t.double()).abs().mean()
    else:
        pc, tc = p.clamp(min=1e-16), t.double().clamp(min=1e-16)
        loss = (pc * (pc / tc).log()).sum() / B + (tc * (tc / pc).log()).sum() / B
    loss.backward()
    stats = torch.zeros(1, device=DEV); dz = torch.empty(B, n, device=DEV)
    ops.align_loss(dev(z), dev(t), kind, 0.25, stats, dz)
    close(stats[0], loss.detach(), 2e-5, 1e-6)
    close(dz, 0.25 * zr.grad, 2e-4, 1e-7)


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("N,Ln,nc,ratio", [(40, 1800, 2, [0.5, 0.5]), (25, 1791, 3, [0.2, 0.5, 0.3]), (9, 300, 2, [0.9, 0.1])])
def test_mix_spectra_vs_oracle(ops, normalize, N, Ln, nc, ratio):
    """afm_mix_spectra vs the numpy restatement of data/datasets.py:49-56,118-126 (itself pinned to the reference's records:
    tests/test_oracle_golden.py; the device generator against the same records: test_mixture_generator_vs_reference_records)."""
    from oracle import afm_oracle as Orc
    rng = np.random.default_rng(N + Ln)
    table = rng.standard_normal((N, Ln)).astype(np.float32)      # negatives exercise the clip-after-min/max quirk
    table[3] = 0.25                                                # flat rows: mixture of two of them normalises to zeros
    table[4] = 0.25
    idx = rng.integers(0, N, (17, nc)).astype(np.int64)
    idx[0, :] = [3, 4, 3][:nc]
    ref = Orc.mix_spectra(table, idx, ratio, normalize)
    out = ops.mix_spectra(torch.from_numpy(table).to(DEV), torch.from_numpy(idx).to(DEV), ratio, normalize=normalize)
    assert torch.equal(out.cpu(), torch.from_numpy(ref))


def test_mixture_generator_records(ops):
    from multimodalanalytical_amd.preprocess import MixtureGenerator
    from oracle import afm_oracle as Orc
    rng = np.random.default_rng(5)
    table = np.abs(rng.standard_normal((12, 1800))).astype(np.float32)
    cfg = dict(n_compounds=2, compounds_ratio=[0.5, 0.5], parallel_samples=8, train_max_n_samples=16, normalize=True)
    rounds = list(MixtureGenerator(torch.from_numpy(table).to(DEV), cfg, "train"))
    ref_rounds = list(Orc.mix_indices(12, cfg, "train"))
    assert len(rounds) == len(ref_rounds) > 0
    for got, ri in zip(rounds, ref_rounds):
        assert np.array_equal(got["indices"], ri)
        ref = Orc.mix_spectra(table, ri, [0.5, 0.5], True)
        assert torch.equal(got["IR"].cpu(), torch.from_numpy(ref).repeat_interleave(2, dim=0))
        assert torch.equal(got["compound"].cpu(), torch.from_numpy(ri.reshape(-1)))
        assert torch.equal(got["IR_target"].cpu(), torch.from_numpy(table[ri.reshape(-1)]))


def test_mixture_generator_vs_reference_records(ops):
    """preprocess.MixtureGenerator (host index stream + afm_mix_spectra on the device) against the records the reference's own
    mix_spectra produced on the same table (tests/golden/mixture.npz): order, mixed spectra bit for bit, targets, percentages;
    incl. zero-weight compounds, spectra shorter than 1800 points and the `mixed` pass-through mode."""
    from multimodalanalytical_amd.preprocess import MixtureGenerator
    from tests.test_oracle_golden import _mixture_cases
    seen = 0
    for tag, cfg, g in _mixture_cases():
        table = torch.from_numpy(g["table"]).to(DEV)
        ir, comp, tgt, pct = [], [], [], []
        for rnd_ in MixtureGenerator(table, cfg, "train", seed=3247):
            ir.append(rnd_["IR"].cpu()); comp.append(rnd_["compound"].cpu()); tgt.append(rnd_["IR_target"].cpu()); pct.append(rnd_["Percentage"])
        ir, comp, tgt, pct = torch.cat(ir), torch.cat(comp), torch.cat(tgt), torch.cat(pct)
        assert torch.equal(comp, torch.from_numpy(g["smiles"])), tag
        assert torch.equal(ir, torch.from_numpy(g["ir"].astype(np.float32))), tag
        assert torch.equal(tgt.double(), torch.from_numpy(g["ir_target"])), tag
        assert [float(x) for x in g["percentage"]] == pct.tolist(), tag
        seen += 1
    assert seen == 6


@pytest.mark.parametrize("K,N,act", [(2, 512, 0), (5, 256, 1), (8, 64, 0)])
def test_gemm_skinny_k_patch_embed_forward(ops, K, N, act):
    """(rows x K<=8) @ (K x N): the two-point-patch embedder of the IR-only workload (modeling/utils.py:107-136)."""
    M = 9000
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    y = torch.empty(M, N, device=DEV)
    ops.gemm(dev(x), dev(w), y, trans_b=True, bias=dev(b), act=act)
    assert ops.last_algo() == "skinny_k"
    ref = x.double() @ w.double().T + b.double()
    close(y, torch.relu(ref) if act else ref, 1e-5, 1e-6)


@pytest.mark.parametrize("accumulate", [False, True])
def test_gemm_skinny_n_patch_embed_wgrad(ops, accumulate):
    R, M, N = 9001, 512, 2
    dy, x = rnd(R, M, seed=1), rnd(R, N, seed=2)
    g0 = rnd(M, N, seed=3)
    g = dev(g0).clone(); gb = torch.zeros(M, device=DEV)
    ops.gemm(dev(dy), dev(x), g, trans_a=True, trans_b=False, accumulate=accumulate, a_colsum=gb)
    assert ops.last_algo() == "skinny_n"
    ref = dy.double().T @ x.double() + (g0.double() if accumulate else 0)
    close(g, ref, 1e-4, 1e-4)
    close(gb, dy.double().sum(0), 1e-4, 1e-4)


@pytest.mark.parametrize("mode", ["bias", "drop", "gelu_preact_drop", "gelu_bwd", "gelu_save_grad_pair"])
def test_gemm_256x256_variant(ops, mode):
    """Persistent 256x256-tile kernel (variant 28; picked automatically for N >= 1024 at M >= 32k): whole tiles only."""
    M, N, K = 768, 512, 256
    a, w, bias = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2).bfloat16(), rnd(N, seed=3)
    t = a.double() @ w.double().T
    c = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    p, seed, site = 0.1, 31, 9
    keep = torch.from_numpy(keep_mask(p, seed, site, M * N)).view(M, N)
    tol = dict(rtol=2e-2, atol=4e-2)
    if mode == "bias":
        ops.gemm(dev(a), dev(w), c, bias=dev(bias), algo=2, variant=28)
        close(c, t + bias.double(), **tol)
    elif mode == "drop":
        ops.gemm(dev(a), dev(w), c, bias=dev(bias), dropout=ops.drop(p, seed, site), algo=2, variant=28)
        close(c, (t + bias.double()) * keep / (1 - p), **tol)
    elif mode == "gelu_preact_drop":
        pre = torch.zeros_like(c)
        ops.gemm(dev(a), dev(w), c, bias=dev(bias), act=2, pre_act=pre, dropout=ops.drop(p, seed, site), algo=2, variant=28)
        close(pre, t + bias.double(), **tol)
        close(c, O.gelu(t + bias.double()) * keep / (1 - p), **tol)
    elif mode == "gelu_save_grad_pair":
        gp = torch.zeros_like(c)
        ops.gemm(dev(a), dev(w), c, bias=dev(bias), act=4, pre_act=gp, dropout=ops.drop(p, seed, site), algo=2, variant=28)
        tr = (t + bias.double()).requires_grad_(True)
        O.gelu(tr).backward(torch.ones(M, N, dtype=torch.float64))
        close(c, O.gelu(tr.detach()) * keep / (1 - p), **tol)
        close(gp, tr.grad * keep / (1 - p), **tol)
        c2 = torch.zeros_like(c)
        ops.gemm(dev(a), dev(w), c2, act=5, pre_act=gp, algo=2, variant=28)
        close(c2, t * gp.float().cpu().double(), **tol)
    else:
        u = rnd(M, N, seed=4)
        ud = dev(u, torch.bfloat16)
        ops.gemm(dev(a), dev(w), c, act=3, pre_act=ud, dropout=ops.drop(p, seed, site), algo=2, variant=28)
        ur = ud.float().cpu().double().requires_grad_(True)
        O.gelu(ur).backward(torch.ones(M, N, dtype=torch.float64))
        close(c, t * keep / (1 - p) * ur.grad, **tol)
    with pytest.raises(Exception):
        ops.gemm(dev(a[:700]), dev(w), c[:700], algo=2, variant=28)     # partial tiles are refused, not mishandled


@pytest.mark.parametrize("R,M,N", [(8192, 512, 512), (12288, 768, 256), (4096 + 64, 304, 520)])
def test_gemm_tn_256x256_variant(ops, R, M, N):
    """wgrad on 256 x 256 tiles (variant 105): full and ragged tiles, split-K atomics, fused bias gradient."""
    dy, x = rnd(R, M, seed=1).bfloat16(), rnd(R, N, seed=2).bfloat16()
    g0 = rnd(M, N, seed=3)
    g = dev(g0).clone(); gb = torch.zeros(M, device=DEV)
    ops.gemm(dev(dy), dev(x), g, trans_a=True, trans_b=False, accumulate=True, algo=2, a_colsum=gb, variant=105)
    assert ops.last_algo().startswith("mfma_tn_ring256")
    close(g, dy.double().T @ x.double() + g0.double(), 2e-3, 2e-2 * math.sqrt(R / 4096))
    close(gb, dy.double().sum(0), 2e-3, 2e-2 * math.sqrt(R / 4096))


@pytest.mark.parametrize("adt,d", [(torch.bfloat16, 512), (torch.float32, 96)])
def test_layernorm_residual_add_with_branch_dropout(ops, adt, d):
    """x_sum = x + dropout(branch) inside the LayerNorm that reads the stream next (vectorised and scalar kernels);
    the mask is the site's stream over the branch tensor, the same one the backward's dx_drop uses."""
    rows = 130
    x, br = rnd(rows, d, seed=1), rnd(rows, d, seed=2)
    gam, bet = 1 + 0.1 * rnd(d, seed=3), 0.1 * rnd(d, seed=4)
    brd = dev(br, adt)
    p, seed, site = 0.1, 77, 5
    keep = torch.from_numpy(keep_mask(p, seed, site, rows * d)).view(rows, d)
    y = torch.empty(rows, d, dtype=torch.bfloat16, device=DEV); xs = torch.empty(rows, d, device=DEV)
    ops.layernorm_fwd(dev(x), dev(gam), dev(bet), y, add=brd, x_sum=xs, add_dropout=ops.drop(p, seed, site))
    ref_sum = x.double() + brd.float().cpu().double() * keep / (1 - p)
    close(xs, ref_sum, 1e-6, 1e-6)
    close(y, O.layer_norm(ref_sum, gam.double(), bet.double()), 1e-2, 1e-2)
    # and the matching backward mask: dx_drop of layernorm_bwd keeps exactly the same elements
    dy = dev(rnd(rows, d, seed=6))
    mean = torch.empty(rows, device=DEV); rstd = torch.empty(rows, device=DEV)
    y32 = torch.empty(rows, d, device=DEV)
    ops.layernorm_fwd(xs, dev(gam), dev(bet), y32, mean, rstd)
    dx = torch.empty(rows, d, device=DEV); dxd = torch.empty(rows, d, device=DEV)
    gg, gb = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    ws = torch.empty(ops.layernorm_bwd_ws(rows, d), device=DEV)
    ops.layernorm_bwd(dy, xs, dev(gam), mean, rstd, dx, gg, gb, ws, dx_drop=dxd, dropout=ops.drop(p, seed, site))
    close(dxd, dx.cpu().double() * keep / (1 - p), 1e-6, 1e-6)
