"""Decoder-row NT products (M = B * T = 16 384: 64 row tiles of 256 on 256 CUs): every tile shape the dispatcher has,
plain bias epilogue, fp16, two alternating rounds in one process.   python tools/experiments/dec_rows_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops


def t(fn, it=50, warm=30):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def main():
    dev = "cuda:0"
    M = int(os.environ.get("M", "16384"))
    shapes = [(1536, 512), (512, 512), (512, 2048), (512, 1536), (2048, 512), (2304, 768), (768, 768), (768, 3072), (1024, 512)]
    variants = [0, 12, 13, 22, 24, 100]
    for N, K in shapes:
        a = (torch.randn(M, K, device=dev) * 0.5).half()
        w = (torch.randn(N, K, device=dev) * 0.05).half()
        bias = torch.randn(N, device=dev)
        c = torch.empty(M, N, dtype=torch.float16, device=dev)
        res = {v: [] for v in variants}
        algo = {}
        for rnd in range(2):
            for v in (variants if rnd == 0 else variants[::-1]):
                try:
                    us = t(lambda: ops.gemm(a, w, c, bias=bias, variant=v, algo=2))
                    algo[v] = ops.last_algo()
                except Exception as e:      # noqa: BLE001
                    us = float("nan")
                res[v].append(us)
        fl = 2.0 * M * N * K
        print(f"{M}x{N}x{K}: " + "  ".join(f"v{v}[{algo.get(v, '-')[8:]}] {res[v][0]:6.1f}/{res[v][1]:6.1f}us ({fl / min(res[v]) / 1e6:5.0f} TF/s)" for v in variants), flush=True)


if __name__ == "__main__":
    main()
