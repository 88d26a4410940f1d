"""Time split-pair GEMMs at the workload's shapes, with the 256 x 256 kernel's ablation switches (variant 320 + bits:
1 skip LDS reads + MFMAs, 2 skip LDS-DMA, 4 skip epilogue)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalanalytical_amd import ops
from multimodalanalytical_amd.x2 import X2

SHAPES = {"qkv": (131072, 1536, 512), "out": (131072, 512, 512), "ffn1": (131072, 2048, 512), "ffn2": (131072, 512, 2048)}


def t(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="qkv,ffn1,ffn2,out")
    ap.add_argument("--variants", default="0,31,32,321,322,323,324,326")
    ap.add_argument("--out", default="x2")
    a = ap.parse_args()
    dev = "cuda:0"
    for name in a.shapes.split(","):
        M, N, K = SHAPES[name]
        x = ops.convert(torch.randn(M, K, device=dev), X2.empty(M, K, dev))
        w = ops.convert(torch.randn(N, K, device=dev), X2.empty(N, K, dev))
        c = X2.empty(M, N, dev) if a.out == "x2" else torch.empty(M, N, device=dev)
        for v in [int(z) for z in a.variants.split(",")]:
            ms = t(lambda: ops.gemm(x, w, c, variant=v))
            print(f"{name:5s} {M}x{N}x{K} variant {v:3d} [{ops.last_algo()}] {ms:.3f} ms  executed {3 * 2.0 * M * N * K / ms / 1e9:.0f} TF/s "
                  f"({3 * 2.0 * M * N * K / ms / 1e9 / 2500:.3f} of peak)")
        # wgrad of the same layer
        dy = ops.convert(torch.randn(M, N, device=dev), X2.empty(M, N, dev))
        g = torch.zeros(N, K, device=dev)
        for v in (0, 105, 106):
            ms = t(lambda: ops.gemm(dy, x, g, trans_a=True, trans_b=False, accumulate=True, variant=v))
            print(f"{name:5s} wgrad {N}x{K} over {M} variant {v:3d} [{ops.last_algo()}] {ms:.3f} ms  executed {3 * 2.0 * M * N * K / ms / 1e9:.0f} TF/s")


if __name__ == "__main__":
    main()
