"""How much of a wgrad launch is its split-K fp32 atomics: the same token count at growing dW sizes (more tiles -> less split-K)."""
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
for R in (131072, 16384):
    for M, N in ((512, 512), (512, 2048), (1024, 2048), (2048, 2048), (2048, 4096), (4096, 4096)):
        dy = torch.randn(R, M, device=dev).half(); x = torch.randn(R, N, device=dev).half()
        g = torch.zeros(M, N, device=dev)
        ms = t(lambda: ops.gemm(dy, x, g, trans_a=True, trans_b=False, accumulate=True))
        print(f"tokens {R:6d} dW {M:4d} x {N:4d}  {ms:7.3f} ms  {2.0 * M * N * R / ms / 1e9:6.0f} TF/s  {ops.last_algo()}")
