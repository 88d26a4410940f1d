"""Run ONE kernel configuration a few times (for rocprofv3 --pmc / --kernel-trace runs)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalanalytical_amd import ops

SHAPES = {"qkv": (131072, 1536, 512), "out": (131072, 512, 512), "ffn1": (131072, 2048, 512), "ffn2": (131072, 512, 2048)}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="gemm_nt")
    ap.add_argument("--shape", default="qkv")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--out", default="bf16")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = "cuda:0"
    M, N, K = SHAPES[a.shape]
    if a.op == "gemm_nt":
        x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
        c = torch.empty(M, N, dtype=torch.bfloat16 if a.out == "bf16" else torch.float32, device=dev)
        if a.op == "gemm_nt" and a.shape == "ffn1" and a.out == "bf16" and a.variant == 0:
            # the training launch of the FFN up-projection: bias + GELU + dropout, keep*scale*GELU' stored for backward
            bias = torch.randn(N, device=dev); pre = torch.empty_like(c); dr = ops.drop(0.1, 1, 1)
            for _ in range(a.iters):
                ops.gemm(x, w, c, bias=bias, act=4, pre_act=pre, dropout=dr)
        else:
            for _ in range(a.iters):
                ops.gemm(x, w, c, variant=a.variant)
    elif a.op == "gemm_tn":
        dy = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, K, device=dev).bfloat16()
        g = torch.zeros(N, K, device=dev)
        for _ in range(a.iters):
            ops.gemm(dy, x, g, trans_a=True, trans_b=False, accumulate=True)
    torch.cuda.synchronize()

if __name__ == "__main__":
    main()
