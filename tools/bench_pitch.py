"""Does a power-of-two row pitch cost bandwidth (channel camping)?  NT GEMM with padded operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalanalytical_amd import ops
from bench_gemm import t
dev = "cuda:0"
for name, M, N, K in [("qkv", 131072, 1536, 512), ("ffn1", 131072, 2048, 512), ("ffn2", 131072, 512, 2048)]:
    for pad in (0, 8, 32, 64, 128):
        a = torch.randn(M, K + pad, device=dev).bfloat16()[:, :K]
        w = torch.randn(N, K + pad, device=dev).bfloat16()[:, :K]
        for cpad in (0, 64):
            c = torch.empty(M, N + cpad, dtype=torch.bfloat16, device=dev)[:, :N]
            res = []
            for var in (0, 6, 10):
                ms = t(lambda: ops.gemm(a, w, c, variant=var))
                res.append(f"v{var} {2*M*N*K/ms/1e9:6.0f}")
            print(f"{name} K={K} pad={pad:3d} cpad={cpad:3d} TF/s: " + "  ".join(res))
