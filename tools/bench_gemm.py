"""Micro-benchmark of afm_gemm at the training step's GEMM shapes (HIP events, per-launch average)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalanalytical_amd import ops

def t(fn, it=20, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it

def main():
    B, S, T, d, f = 128, 1024, 128, 512, 2048
    dev = "cuda:0"
    shapes = [("qkv fwd", B*S, 3*d, d), ("out fwd", B*S, d, d), ("ffn1 fwd", B*S, f, d), ("ffn2 fwd", B*S, d, f),
              ("dec qkv", B*T, 3*d, d), ("lm head", B*T, 128, d)]
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
        for cdt in (torch.bfloat16, torch.float32):
            c = torch.empty(M, N, dtype=cdt, device=dev)
            res = []
            for var in (12, 24) + ((28,) if cdt == torch.bfloat16 else ()):
                ms = t(lambda: ops.gemm(a, w, c, variant=var))
                res.append(f"v{var} {2*M*N*K/ms/1e9:6.0f}")
            if cdt == torch.bfloat16:   # fused training epilogues: fwd bias+GELU+dropout (pre-activation kept), bwd GELU'
                bias = torch.randn(N, device=dev); pre = torch.empty_like(c); dr = ops.drop(0.1, 1, 1)
                for var in (24, 28):
                    ms = t(lambda: ops.gemm(a, w, c, bias=bias, act=2, pre_act=pre, dropout=dr, variant=var))
                    res.append(f"gelu-v{var} {2*M*N*K/ms/1e9:5.0f}")
                    ms = t(lambda: ops.gemm(a, w, c, act=3, pre_act=pre, dropout=dr, variant=var))
                    res.append(f"gbwd-v{var} {2*M*N*K/ms/1e9:5.0f}")
                    if M % 256 == 0 and N % 256 == 0:   # save-grad pair (whole tiles only)
                        ms = t(lambda: ops.gemm(a, w, c, bias=bias, act=4, pre_act=pre, dropout=dr, variant=var))
                        res.append(f"gsg-v{var} {2*M*N*K/ms/1e9:5.0f}")
                        ms = t(lambda: ops.gemm(a, w, c, act=5, pre_act=pre, variant=var))
                        res.append(f"mul-v{var} {2*M*N*K/ms/1e9:5.0f}")
                    ms = t(lambda: ops.gemm(a, w, c, bias=bias, dropout=dr, variant=var))
                    res.append(f"drop-v{var} {2*M*N*K/ms/1e9:5.0f}")
            print(f"NT {name:10s} {M}x{N}x{K} out {str(cdt)[6:]:8s} TF/s: " + "  ".join(res))
    if "--ablate" in sys.argv:   # needs a build with AFM_EXTRA_FLAGS=-DAFM_GEMM_ABLATIONS
        labels = {24: "v24", 241: "no-mfma", 242: "no-dma", 243: "epi-only", 244: "no-epi", 246: "mfma-only", 247: "loop-only"}
        for name, M, N, K in shapes[:4]:
            a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
            c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            res = []
            for var, lab in labels.items():
                ms = t(lambda: ops.gemm(a, w, c, variant=var))
                res.append(f"{lab} {ms*1e3:6.0f}us")
            print(f"ABL {name:10s}: " + "  ".join(res))
    for name, R, M, N in [("wgrad qkv", B*S, 3*d, d), ("wgrad out", B*S, d, d), ("wgrad ffn1", B*S, f, d), ("wgrad ffn2", B*S, d, f)]:
        dy = torch.randn(R, M, device=dev).bfloat16(); x = torch.randn(R, N, device=dev).bfloat16()
        g = torch.zeros(M, N, device=dev)
        gb = torch.zeros(M, device=dev)
        for var in (106, 105):
            ms = t(lambda: ops.gemm(dy, x, g, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb, variant=var))
            print(f"TN {name:10s} {R}: {M}x{N} {ops.last_algo():20s} {ms:8.3f} ms {2*M*N*R/ms/1e9:8.1f} TF/s")

if __name__ == "__main__":
    main()
