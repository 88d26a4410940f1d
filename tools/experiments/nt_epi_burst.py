"""What the epilogue of the K = 512 NT GEMMs costs, and why (round 4).  Needs an ablation build:

    AFM_BUILD_VARIANT=abl AFM_EXTRA_FLAGS=-DAFM_GEMM_ABLATIONS python -m multimodalanalytical_amd.csrc.build
    AFM_LIB_OVERRIDE=tools/experiments/_abl/libafm_abl.so python tools/experiments/nt_epi_burst.py

variants of the loader-wave kernel (256 x 128 tiles, plain compile-time epilogue, bias): 249 as shipped, 248 the same instruction
stream with every tile's stores aimed at tile 0 (L2-resident: the epilogue's LDS / vector work without its HBM writes), 250 no
epilogue at all.  If 248 sits near 250, the epilogue's cost is the write burst of all CUs at once, not its arithmetic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops


def t(fn, it=30, warm=30):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def main():
    dev = "cuda:0"
    B, S, d, f = 128, 1024, 512, 2048
    M = B * S
    shapes = [("qkv fwd", M, 3 * d, d), ("out fwd", M, d, d), ("ffn1 plain", M, f, d), ("ffn2 fwd", M, d, f), ("qkv dgrad", M, d, 3 * d)]
    labels = {249: "at-once", 251: "metered", 256: "at-once-nt", 257: "at-once-sc1nt", 254: "metered-nt", 255: "metered-sc1nt", 248: "stores->L2", 250: "no-epilogue", 246: "lds+mfma-only", 28: "256x256"}
    dr = ops.drop(0.1, 1, 1)
    # the metered stores must not change a bit: plain, GELU + dropout + stored factor, x stored
    for name, m, n, k in shapes[:3]:
        a = torch.randn(m, k, device=dev).half(); w = (torch.randn(n, k, device=dev) * 0.05).half(); bias = torch.randn(n, device=dev)
        c0 = torch.empty(m, n, dtype=torch.float16, device=dev); c1 = torch.empty_like(c0); p0 = torch.empty_like(c0); p1 = torch.empty_like(c0)
        ops.gemm(a, w, c0, bias=bias, variant=249)
        for v in (251, 254, 255, 256, 257):
            c1.zero_(); ops.gemm(a, w, c1, bias=bias, variant=v)
            print(f"{name}: variant {v} == at-once: {torch.equal(c0, c1)}", flush=True)
        if n % 256 == 0:
            ops.gemm(a, w, c0, bias=bias, act=4, pre_act=p0, dropout=dr, variant=28); ops.gemm(a, w, c1, bias=bias, act=4, pre_act=p1, dropout=dr, variant=24)
            print(f"{name}: gelu-sg 256x256 == metered 256x128: C {torch.equal(c0, c1)} P {torch.equal(p0, p1)}", flush=True)
            pre = torch.randn(m, n, device=dev).half()
            ops.gemm(a, w, c0, act=5, pre_act=pre, variant=28); ops.gemm(a, w, c1, act=5, pre_act=pre, variant=24)
            print(f"{name}: x saved 256x256 == metered 256x128: {torch.equal(c0, c1)}", flush=True)
    for rnd in range(2):
        for name, m, n, k in shapes:
            a = torch.randn(m, k, device=dev).half(); w = (torch.randn(n, k, device=dev) * 0.05).half()
            c = torch.empty(m, n, dtype=torch.float16, device=dev); bias = torch.randn(n, device=dev)
            res = []
            for var, lab in labels.items():
                ms = t(lambda: ops.gemm(a, w, c, bias=bias, variant=var))
                res.append(f"{lab} {ms * 1e3:6.1f}us")
            print(f"{name:11s} {m}x{n}x{k}: " + "  ".join(res), flush=True)
            if n % 256 == 0:
                pre = torch.randn(m, n, device=dev).half(); res = []
                for var in (24, 28):
                    ms = t(lambda: ops.gemm(a, w, c, bias=bias, act=4, pre_act=pre, dropout=dr, variant=var)); res.append(f"gelu-sg v{var} {ms * 1e3:6.1f}us")
                    ms = t(lambda: ops.gemm(a, w, c, act=5, pre_act=pre, variant=var)); res.append(f"x-saved v{var} {ms * 1e3:6.1f}us")
                    ms = t(lambda: ops.gemm(a, w, c, bias=bias, dropout=dr, variant=var)); res.append(f"drop v{var} {ms * 1e3:6.1f}us")
                print(f"{'':11s} " + "  ".join(res), flush=True)


if __name__ == "__main__":
    main()
