"""Backward with the keep-bit tensor must equal backward with re-hashed dropout bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops
from multimodalanalytical_amd.x2 import X2
dev = "cuda:0"
for mode in ("bf16", "bf16x3"):
    for (B, H, Tq, Tk, causal, padded) in ((1, 2, 256, 256, False, False), (2, 2, 128, 256, False, False), (1, 2, 160, 160, True, False), (1, 1, 64, 64, False, False),
                                           (2, 8, 128, 1024, False, True), (2, 8, 128, 128, True, True), (2, 8, 1024, 1024, False, True), (2, 2, 100, 300, False, True)):
        cd = torch.bfloat16 if mode == "bf16" else X2.dtype
        d = H * 64
        torch.manual_seed(0)
        mk = lambda r, c, sc=1.0: ops.convert(torch.randn(r, c, device=dev) * sc, ops.empty(r, c, cd, dev))
        q, k, v, do = mk(B * Tq, d), mk(B * Tk, d), mk(B * Tk, d), mk(B * Tq, d, 0.1)
        pad = None
        if padded:   # key padding: ragged valid lengths, one row with a long padded tail
            lens = torch.tensor([max(1, Tk - 37 - 300 * (i % 2) if Tk > 400 else Tk - 5 - 40 * (i % 2)) for i in range(B)])
            pad = (torch.arange(Tk)[None, :] >= lens[:, None]).to(torch.uint8).to(dev).contiguous()
        res = {}
        for use_bits in (False, True):
            o = ops.empty(B * Tq, d, cd, dev); lse = torch.empty(B * H * Tq, device=dev); delta = torch.empty_like(lse)
            shp = ops.attn_shape(B, H, Tq, Tk, 64, cd, ops._ld(q), ops._ld(k), ops._ld(v), ops._ld(o), pad, causal, ops.drop(0.1, 7, 3), algo=2)
            if use_bits:
                ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=dev))
            ops.attn_fwd(shp, q, k, v, o, lse)
            dq, dk, dv = (ops.empty(n, d, cd, dev) for n in (B * Tq, B * Tk, B * Tk))
            ops.attn_bwd(shp, q, k, v, o, do, lse, delta, dq, dk, dv, ops._ld(dq), ops._ld(dk), ops._ld(dv))
            f = (lambda t: t.float()) 
            res[use_bits] = [f(o), f(dq), f(dk), f(dv)]
        names = ["o", "dq", "dk", "dv"]
        print(mode, (B, H, Tq, Tk, causal, padded), {n: float((a - b).abs().max()) for n, a, b in zip(names, res[False], res[True])})
