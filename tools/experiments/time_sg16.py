"""FFN up-projection forward in fp16 at the c2 shape: plain (+ bias) vs GELU + dropout + stored factor, and the same without dropout
(how much of the fused epilogue is the hash).  `AFM_LIB_OVERRIDE` may name an experiment build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops
from multimodalanalytical_amd.lib import ACT_GELU_SAVE_GRAD

dev = "cuda:0"
M, N, K = 131072, 2048, 512
x = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half(); b = torch.zeros(N, device=dev)
g = torch.empty(M, N, dtype=torch.float16, device=dev); pre = torch.empty_like(g)


def t(fn, iters=30, warm=30):
    for _ in range(warm): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters


for rep in range(2):
    print("plain                    %.3f ms" % t(lambda: ops.gemm(x, w, g, trans_b=True, bias=b)))
    print("gelu_sg + dropout 0.1    %.3f ms" % t(lambda: ops.gemm(x, w, g, trans_b=True, bias=b, act=ACT_GELU_SAVE_GRAD, pre_act=pre, dropout=ops.drop(0.1, 1, 3))))
    print("gelu_sg, no dropout      %.3f ms" % t(lambda: ops.gemm(x, w, g, trans_b=True, bias=b, act=ACT_GELU_SAVE_GRAD, pre_act=pre)))
