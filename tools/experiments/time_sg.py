import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops
from multimodalanalytical_amd.lib import ACT_GELU_SAVE_GRAD, ACT_NONE
from multimodalanalytical_amd.x2 import X2
dev = "cuda:0"
M, N, K = 131072, 2048, 512
x = ops.convert(torch.randn(M, K, device=dev), X2.empty(M, K, dev))
w = ops.convert(torch.randn(N, K, device=dev) * 0.05, X2.empty(N, K, dev))
b = torch.zeros(N, device=dev)
g, pre = X2.empty(M, N, dev), X2.empty(M, N, dev)
dr = ops.drop(0.1, 1, 3)


def t(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters


for rep in range(2):
    print("plain          %.3f ms" % t(lambda: ops.gemm(x, w, g, trans_b=True, bias=b)))
    print("gelu_sg pairs  %.3f ms" % t(lambda: ops.gemm(x, w, g, trans_b=True, bias=b, act=ACT_GELU_SAVE_GRAD, pre_act=pre, dropout=dr)))
    print("gelu_sg hi     %.3f ms" % t(lambda: ops.gemm(x, w, g, trans_b=True, bias=b, act=ACT_GELU_SAVE_GRAD, pre_act=pre, dropout=dr, sg_hi_only=True)))
