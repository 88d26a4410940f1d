"""Leading-dimension padding.  The cross-attention experiment (attn_cross_layout.py) found that 128-byte pieces written 12 KB apart run
17 % slower than the same pieces 12 KB + 128 B apart: a channel / bank pattern of the memory system.  Here the same question for the
other strided accesses of the step: the GEMMs' A / C operands with rows of 1, 3, 4 KB (K = 512, N = 1 536, K = 2 048: powers of two
or three times one) and the self-attention's packed Q | K | V (3-KB rows), each with its natural row stride and padded by 128 bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops
from attn_m16 import t


def buf(rows, cols, pad, dt=torch.float16, rnd=True, scale=1.0):
    b = (torch.randn(rows, cols + pad, device="cuda") * scale).to(dt) if rnd else torch.empty(rows, cols + pad, dtype=dt, device="cuda")
    return b[:, :cols]


def gemms():
    M = 131072
    dr = ops.drop(0.1, 1, 1)
    # name, N, K, kwargs
    shapes = [("qkv fwd (C rows 3 KB)", 1536, 512, dict(bias=1)), ("out-proj (A, C rows 1 KB)", 512, 512, dict(bias=1)),
              ("ffn up EPI 5 (C, sg rows 4 KB)", 2048, 512, dict(bias=1, act=4, pre=1, drop=1)), ("ffn down (A rows 4 KB)", 512, 2048, dict(bias=1)),
              ("x stored dgrad EPI 6 (C, sg rows 4 KB)", 2048, 512, dict(act=5, pre=1)), ("ffn up dgrad (A rows 4 KB)", 512, 2048, {}),
              ("qkv dgrad (A rows 3 KB)", 512, 1536, {})]
    for name, N, K, kw in shapes:
        res = {}
        for rnd in range(2):
            for pad in ((0, 64) if rnd == 0 else (64, 0)):
                x = buf(M, K, pad); w = (torch.randn(N, K, device="cuda") * 0.05).half(); c = buf(M, N, pad, rnd=False)
                args = {}
                if kw.get("bias"): args["bias"] = torch.randn(N, device="cuda")
                if kw.get("pre"): args["pre_act"] = buf(M, N, pad)
                if kw.get("act"): args["act"] = kw["act"]
                if kw.get("drop"): args["dropout"] = dr
                ms = t(lambda: ops.gemm(x, w, c, **args), it=40, warm=20)
                res.setdefault(pad, []).append(ms)
        print(f"{name:42s} N {N:5d} K {K:5d}: natural rows " + " / ".join(f"{1e3 * v:.1f}" for v in res[0]) + " us   rows + 128 B " +
              " / ".join(f"{1e3 * v:.1f}" for v in res[64]) + f" us   [{ops.last_algo()}]", flush=True)


def gemms_c4():
    """c4's gated FFN: the fused up-projection (EPI 8: C = gelu(u) v rows of 6 KB, stored factors rows of 12 KB) and its data gradient
    (EPI 9: reads the 12-KB rows, writes [du | dv] rows of 12 KB), the down-projection and the 2f data gradient (A rows of 6 / 12 KB)."""
    M, d, f = 131072, 768, 3072
    dr = ops.drop(0.1, 1, 1)
    for name, N, K, kw in [("c4 glu fwd EPI 8 (C 6 KB, factors 12 KB rows)", 2 * f, d, dict(bias=1, act=7, pre=1, drop=1, glu=f)),
                           ("c4 glu dgrad EPI 9 (factors, C 12 KB rows)", f, d, dict(act=8, pre=1, glu=f)),
                           ("c4 ffn down (A rows 6 KB)", d, f, dict(bias=1)), ("c4 ffn up dgrad (A rows 12 KB)", d, 2 * f, {}),
                           ("c4 qkv fwd (C rows 4.5 KB)", 3 * d, d, dict(bias=1))]:
        res = {}
        for rnd in range(2):
            for pad in ((0, 64) if rnd == 0 else (64, 0)):
                act = kw.get("act", 0)
                cn = N // 2 if act == 7 else 2 * N if act == 8 else N
                x = buf(M, K, pad); w = (torch.randn(N, K, device="cuda") * 0.05).half(); c = buf(M, cn, pad, rnd=False)
                args = {}
                if kw.get("bias"): args["bias"] = torch.randn(N, device="cuda")
                if kw.get("pre"): args["pre_act"] = buf(M, 2 * N if act == 8 else N, pad)
                if act: args["act"] = act
                if kw.get("drop"): args["dropout"] = dr
                if kw.get("glu"): args["glu_rows"] = kw["glu"]
                try:
                    ms = t(lambda: ops.gemm(x, w, c, **args), it=20, warm=10)
                except AssertionError:      # the gated forms take contiguous C / factor tensors only
                    ms = float("nan")
                res.setdefault(pad, []).append(ms)
        print(f"{name:48s} N {N:5d} K {K:5d}: natural rows " + " / ".join(f"{1e3 * v:.1f}" for v in res[0]) + " us   rows + 128 B " +
              " / ".join(f"{1e3 * v:.1f}" for v in res[64]) + f" us   [{ops.last_algo()}]", flush=True)


def wgrads():
    M, d, f = 131072, 512, 2048
    for name, group in (("c2 encoder layer weight gradients (grouped)", [(d, f), (f, d), (d, d), (3 * d, d)]),):
        res = {}
        for rnd in range(2):
            for pad in ((0, 64) if rnd == 0 else (64, 0)):
                ten = [(buf(M, m, pad, scale=0.01), buf(M, n, pad), torch.zeros(m, n, device="cuda"), torch.zeros(m, device="cuda")) for m, n in group]
                descs = [ops.gemm_desc(dy, xx, gw, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb) for dy, xx, gw, gb in ten]
                res.setdefault(pad, []).append(t(lambda: ops.gemm_group(descs), it=20, warm=10))
        print(f"{name}: natural rows " + " / ".join(f"{1e3 * v:.1f}" for v in res[0]) + " us   rows + 128 B " + " / ".join(f"{1e3 * v:.1f}" for v in res[64]) + " us", flush=True)


def attention():
    B, H, S, dh, dt, dev = 128, 8, 1024, 64, torch.float16, "cuda:0"
    d = H * dh
    dr = ops.drop(0.1, 1, 3)
    kb = torch.zeros(ops.attn_drop_bits_words(B, H, S, S), dtype=torch.int64, device=dev)
    res = {}
    for rnd in range(2):
        for pad in ((0, 64) if rnd == 0 else (64, 0)):
            qkv = buf(B * S, 3 * d, pad); dqkv = buf(B * S, 3 * d, pad, rnd=False)
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            dq, dk, dv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
            o = buf(B * S, d, pad, rnd=False); do = buf(B * S, d, pad, scale=0.01)
            lse, delta = torch.empty(B * H * S, device=dev), torch.empty(B * H * S, device=dev)

            def shape(r):
                s = ops.attn_shape(B, H, S, S, dh, dt, ops._ld(q), ops._ld(k), ops._ld(v), ops._ld(o), None, False, dr)
                s.reserved = r
                return ops.attn_set_drop_bits(s, kb)
            s0, s1, s2 = shape(0), shape(1), shape(2)
            ops.attn_fwd(s0, q, k, v, o, lse)
            f1 = lambda: ops.attn_bwd(s1, q, k, v, o, do, lse, delta, dq, dk, dv, ops._ld(dq), ops._ld(dk), ops._ld(dv))
            f1()
            r = res.setdefault(pad, {})
            r.setdefault("fwd", []).append(t(lambda: ops.attn_fwd(s0, q, k, v, o, lse)))
            r.setdefault("dQ", []).append(t(f1))
            r.setdefault("dK/dV", []).append(t(lambda: ops.attn_bwd(s2, q, k, v, o, do, lse, delta, dq, dk, dv, ops._ld(dq), ops._ld(dk), ops._ld(dv))))
    for n in ("fwd", "dQ", "dK/dV"):
        print(f"self-attention {n:6s} (packed Q | K | V, 3-KB rows): natural " + " / ".join(f"{1e3 * v:.1f}" for v in res[0][n]) + " us   rows + 128 B " +
              " / ".join(f"{1e3 * v:.1f}" for v in res[64][n]) + " us", flush=True)


if __name__ == "__main__":
    if "--c4" in sys.argv:
        gemms_c4()
        sys.exit(0)
    gemms()
    wgrads()
    attention()
