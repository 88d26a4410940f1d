import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops
from multimodalanalytical_amd.x2 import X2
dev = "cuda:0"


def t(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3


for mode in ("bf16x3", "bf16"):
    cd = X2.dtype if mode == "bf16x3" else torch.bfloat16
    for (M, N, K) in ((128, 512, 512), (128, 1536, 512), (128, 2048, 512), (128, 512, 2048), (640, 512, 512), (640, 2048, 512), (128, 128, 512)):
        x = ops.convert(torch.randn(M, K, device=dev), ops.empty(M, K, cd, dev))
        w = ops.convert(torch.randn(N, K, device=dev) * 0.05, ops.empty(N, K, cd, dev))
        b = torch.zeros(N, device=dev)
        c = ops.empty(M, N, cd, dev)
        cf = torch.empty(M, N, device=dev)
        us = t(lambda: ops.gemm(x, w, c, trans_b=True, bias=b)); a1 = ops.last_algo()
        us2 = t(lambda: ops.gemm(x, w, cf, trans_b=True, bias=b)); a2 = ops.last_algo()
        print(f"{mode} {M}x{N}x{K}: {us:6.1f} us [{a1}]   fp32 out {us2:6.1f} us [{a2}]")
