"""Library reference for the plain NT products of the c2 step: torch's F.linear (hipBLASLt / rocBLAS underneath) against afm_gemm on
the same tensors, one process, HIP events.  A yardstick only: the product path never calls it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from multimodalanalytical_amd import ops


def t(fn, it=30, warm=30):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


M = 131072
shapes = [("qkv fwd", M, 1536, 512), ("out fwd", M, 512, 512), ("ffn1 plain", M, 2048, 512), ("ffn2 fwd", M, 512, 2048),
          ("qkv dgrad", M, 512, 1536), ("mem kv", M, 1024, 512), ("dec qkv", 16384, 1536, 512)]
for rnd in range(2):
    for name, m, n, k in shapes:
        a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * 0.05).half()
        bias = torch.randn(n, device="cuda"); bh = bias.half()
        c = torch.empty(m, n, dtype=torch.float16, device="cuda")
        ms_lib = t(lambda: F.linear(a, w, bh))
        ms_lib_nb = t(lambda: torch.matmul(a, w.t(), out=c))
        ms_afm = t(lambda: ops.gemm(a, w, c, bias=bias))
        fl = 2.0 * m * n * k
        print(f"{name:11s} {m}x{n}x{k}: library + bias {ms_lib * 1e3:6.1f} us ({fl / ms_lib / 1e9:5.0f} TF)  library plain {ms_lib_nb * 1e3:6.1f} us ({fl / ms_lib_nb / 1e9:5.0f} TF)"
              f"  afm_gemm + bias {ms_afm * 1e3:6.1f} us ({fl / ms_afm / 1e9:5.0f} TF) [{ops.last_algo()}]", flush=True)
    # weight gradient (TN): dW = dY^T X
    for name, r, mm, nn in [("ffn1 wgrad", M, 2048, 512), ("qkv wgrad", M, 1536, 512)]:
        dy = torch.randn(r, mm, device="cuda").half(); x = torch.randn(r, nn, device="cuda").half()
        ms_lib = t(lambda: torch.matmul(dy.t(), x))
        print(f"{name:11s} {mm}x{nn} over {r} rows: library {ms_lib * 1e3:6.1f} us ({2.0 * r * mm * nn / ms_lib / 1e9:5.0f} TF)", flush=True)
