"""Gated-FFN fused epilogues (act 6 / 7 / 8) on the 256 x 256 persistent kernel (afm_gemm_desc.reserved = 28) against the 256 x 128
loader-wave kernel (24): bit-equality of every output, then times at the c4 (f 3072, d 768) and c5 (f 2048, d 512) shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops
from multimodalanalytical_amd.lib import ACT_GLU, ACT_GLU_SAVE, ACT_GLU_BWD

dev = "cuda:0"
H16 = torch.float16


def t(fn, it=20, warm=20):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def case(M, f, d, time_it=False):
    g0 = torch.Generator(device=dev).manual_seed(1)
    h = torch.randn(M, d, device=dev, generator=g0).half()
    wcat = torch.randn(2 * f, d, device=dev, generator=g0) * 0.05
    bias = torch.randn(2 * f, device=dev, generator=g0) * 0.1
    w_il, wt_il = torch.empty(2 * f, d, dtype=H16, device=dev), torch.empty(d, 2 * f, dtype=H16, device=dev)
    ops.cast_weights(wcat, w_il, wt_il, glu_rows=f)
    w2 = (torch.randn(d, f, device=dev, generator=g0) * 0.05).half()          # down projection (d x f): dg = dy W2
    dy = (torch.randn(M, d, device=dev, generator=g0) * 0.01).half()
    dr = ops.drop(0.1, 7, 3)
    out = {}
    for v in (24, 28):
        g = torch.full((M, f), float("nan"), dtype=H16, device=dev); uv = torch.full((M, 2 * f), float("nan"), dtype=H16, device=dev)
        ops.gemm(h, w_il, g, bias=bias, act=ACT_GLU_SAVE, pre_act=uv, glu_rows=f, dropout=dr, variant=v)
        g2 = torch.full_like(g, float("nan"))
        ops.gemm(h, w_il, g2, bias=bias, act=ACT_GLU, glu_rows=f, dropout=dr, variant=v)
        duv = torch.full((M, 2 * f), float("nan"), dtype=H16, device=dev)
        ops.gemm(dy, w2.t().contiguous(), duv, act=ACT_GLU_BWD, pre_act=uv, glu_rows=f, variant=v)
        torch.cuda.synchronize()
        out[v] = (g, uv, g2, duv)
    same = all(torch.equal(a, b) for a, b in zip(out[24], out[28]))
    fin = all(bool(torch.isfinite(x.float()).all()) for x in out[28])
    print(f"M {M} f {f} d {d}: 256x256 == 256x128: {same}, finite {fin}  [{ops.last_algo()}]", flush=True)
    if time_it:
        g, uv, g2, duv = out[28]
        w2t = w2.t().contiguous()
        for rnd in range(2):
            for v in (24, 28):
                a = t(lambda: ops.gemm(h, w_il, g, bias=bias, act=ACT_GLU_SAVE, pre_act=uv, glu_rows=f, dropout=dr, variant=v))
                b = t(lambda: ops.gemm(dy, w2t, duv, act=ACT_GLU_BWD, pre_act=uv, glu_rows=f, variant=v))
                print(f"   variant {v}: up-projection + GELU pair + dropout + stored {a:.3f} ms   data gradient x stored {b:.3f} ms", flush=True)
    return same and fin


ok = case(8192, 2048, 512) and case(16384, 1024, 256)
print("ALL OK" if ok else "FAILURES", flush=True)
case(131072, 3072, 768, time_it=True)
case(131072, 2048, 512, time_it=True)
