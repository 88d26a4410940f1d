"""The step's dominant NT GEMM launches (fp16, c2 shapes), each R times in a fixed order, for rocprofv3 --pmc / --kernel-trace runs
(tools/prof_gemm_pmc.sh).  `--list` prints the configuration names in launch order (the reducer maps the i-th run of R consecutive
k_gemm dispatches to the i-th name)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

M, d, f = 131072, 512, 2048
# name, N, K, kwargs (bias / act / pre_act / dropout), variant
CONFIGS = [
    ("qkv_fwd_default", 3 * d, d, dict(bias=1), 0),
    ("qkv_fwd_ws256x128", 3 * d, d, dict(bias=1), 24),
    ("qkv_fwd_pingpong", 3 * d, d, dict(bias=1), 30),
    ("ffn2_fwd_default", d, f, dict(bias=1), 0),
    ("ffn2_fwd_pingpong", d, f, dict(bias=1), 30),
    ("ffn1_fwd_gelu_drop_sg_default", f, d, dict(bias=1, act=4, pre=1, drop=1), 0),
    ("ffn2_dgrad_xsaved_default", f, d, dict(act=5, pre=1), 0),
    ("ffn1_plain_default", f, d, dict(bias=1), 0),
    ("ffn1_plain_pingpong", f, d, dict(bias=1), 30),
    ("qkv_dgrad_default", d, 3 * d, dict(), 0),
    ("ffn1_wgrad_default", f, d, dict(tn=1), 0),          # dW1 (f x d) over M tokens, bias gradient fused
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", action="store_true"); ap.add_argument("--reps", type=int, default=12); ap.add_argument("--warm", type=int, default=0)
    a = ap.parse_args()
    if a.list:
        print("\n".join(c[0] for c in CONFIGS)); return
    import torch
    from multimodalanalytical_amd import ops
    dev = "cuda:0"
    dr = ops.drop(0.1, 1, 1)
    for name, N, K, kw, var in CONFIGS:
        if kw.get("tn"):
            dy = (torch.randn(M, N, device=dev) * 0.01).half(); xx = torch.randn(M, K, device=dev).half()
            gw = torch.zeros(N, K, device=dev); gb = torch.zeros(N, device=dev)
            torch.cuda.synchronize()
            for _ in range(a.reps): ops.gemm(dy, xx, gw, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb)
            torch.cuda.synchronize()
            print(name, ops.last_algo(), flush=True)
            continue
        x = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
        c = torch.empty(M, N, dtype=torch.float16, device=dev)
        args = dict(variant=var)
        if kw.get("bias"): args["bias"] = torch.randn(N, device=dev)
        if kw.get("pre"): args["pre_act"] = torch.randn(M, N, device=dev).half()
        if kw.get("act"): args["act"] = kw["act"]
        if kw.get("drop"): args["dropout"] = dr
        torch.cuda.synchronize()
        for _ in range(a.reps): ops.gemm(x, w, c, **args)
        torch.cuda.synchronize()
        print(name, ops.last_algo(), flush=True)


if __name__ == "__main__":
    main()
