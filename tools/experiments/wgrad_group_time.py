"""Grouped (afm_gemm_group) vs one-by-one weight gradients of a c2 encoder / decoder layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops


def t(fn, it=20, warm=10):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


dev = "cuda:0"
sets = {"enc layer (131072 tokens)": [(131072, 512, 512), (131072, 1536, 512), (131072, 512, 2048), (131072, 2048, 512)],
        "dec layer (16384 tokens)": [(16384, 512, 512)] * 3 + [(16384, 1536, 512), (16384, 512, 2048), (16384, 2048, 512)],
        "dec layer + its memory k|v": [(16384, 512, 512)] * 3 + [(16384, 1536, 512), (16384, 512, 2048), (16384, 2048, 512), (131072, 1024, 512)],
        "memory k|v of 6 layers": [(131072, 1024, 512)] * 6}
CS = os.environ.get("NO_COLSUM") is None
for name, probs in sets.items():
    ten, descs, flop = [], [], 0.0
    for R, M, N in probs:
        dy = torch.randn(R, M, device=dev).half(); x = torch.randn(R, N, device=dev).half()
        g = torch.zeros(M, N, device=dev); gb = torch.zeros(M, device=dev) if CS else None
        ten.append((dy, x, g, gb)); flop += 2.0 * R * M * N
        descs.append(ops.gemm_desc(dy, x, g, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb))
    one = t(lambda: [ops.gemm(dy, x, g, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb) for dy, x, g, gb in ten])
    grp = t(lambda: ops.gemm_group(descs))
    print(f"{name:30s} one by one {one:7.3f} ms ({flop / one / 1e9:5.0f} TF/s)   grouped {grp:7.3f} ms ({flop / grp / 1e9:5.0f} TF/s)  {ops.last_algo()}")
