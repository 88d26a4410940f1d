"""The step's dominant NT GEMM launches (fp16, c2 shapes), each R times in a fixed order, for rocprofv3 --pmc / --kernel-trace runs
(tools/prof_gemm_pmc.sh).  `--list` prints the configuration names in launch order (the reducer maps the i-th run of R consecutive
k_gemm dispatches to the i-th name)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

M = 131072


def configs(which):
    """name, N, K, kwargs (bias / act / pre_act / dropout / tn / group / glu), variant.  c2: d 512, f 2048 (plain FFN); c4: d 768, f 3072 (gated)."""
    if which == "c4":
        d, f = 768, 3072
        return [
            ("c4_qkv_fwd_default", 3 * d, d, dict(bias=1), 0),
            ("c4_qkv_fwd_pingpong", 3 * d, d, dict(bias=1), 32),
            ("c4_ffn2_fwd_default", d, f, dict(bias=1), 0),
            ("c4_ffn2_fwd_pingpong", d, f, dict(bias=1), 32),
            ("c4_qkv_dgrad_default", d, 3 * d, dict(), 0),
            ("c4_ffn1_dgrad_2f_default", d, 2 * f, dict(), 0),
            ("c4_glu_fwd_sg_default", 2 * f, d, dict(bias=1, act=7, pre=1, drop=1, glu=f), 0),      # EPI 8: gelu(u) v + dropout + stored factors
            ("c4_glu_dgrad_default", f, d, dict(act=8, pre=1, glu=f), 0),                            # EPI 9: [du | dv] = dg x stored factors
            ("c4_enc_layer_wgrad_group_default", 0, 0, dict(group=[(d, f), (2 * f, d), (d, d), (3 * d, d)], glu=f), 0),
            ("c4_enc_layer_wgrad_group_eightwave", 0, 0, dict(group=[(d, f), (2 * f, d), (d, d), (3 * d, d)], glu=f), 107),
        ]
    d, f = 512, 2048
    return [
        ("qkv_fwd_default", 3 * d, d, dict(bias=1), 0),
        ("qkv_fwd_ws256x128", 3 * d, d, dict(bias=1), 24),
        ("qkv_fwd_pingpong", 3 * d, d, dict(bias=1), 30),
        ("ffn2_fwd_default", d, f, dict(bias=1), 0),
        ("ffn2_fwd_pingpong", d, f, dict(bias=1), 32),
        ("ffn1_fwd_gelu_drop_sg_default", f, d, dict(bias=1, act=4, pre=1, drop=1), 0),
        ("ffn2_dgrad_xsaved_default", f, d, dict(act=5, pre=1), 0),
        ("ffn1_plain_default", f, d, dict(bias=1), 0),
        ("ffn1_plain_pingpong", f, d, dict(bias=1), 30),
        ("qkv_dgrad_default", d, 3 * d, dict(), 0),
        ("qkv_dgrad_pingpong", d, 3 * d, dict(), 32),
        ("ffn1_wgrad_default", f, d, dict(tn=1), 0),          # dW1 (f x d) over M tokens, bias gradient fused
        ("ffn1_wgrad_eightwave", f, d, dict(tn=1), 107),
        ("enc_layer_wgrad_group_default", 0, 0, dict(group=[(d, f), (f, d), (d, d), (3 * d, d)]), 0),
        ("enc_layer_wgrad_group_eightwave", 0, 0, dict(group=[(d, f), (f, d), (d, d), (3 * d, d)]), 107),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", action="store_true"); ap.add_argument("--reps", type=int, default=12); ap.add_argument("--warm", type=int, default=0)
    ap.add_argument("--set", default="c2")
    a = ap.parse_args()
    CONFIGS = configs(a.set)
    if a.list:
        print("\n".join(c[0] for c in CONFIGS)); return
    import torch
    from multimodalanalytical_amd import ops
    dev = "cuda:0"
    dr = ops.drop(0.1, 1, 1)
    for name, N, K, kw, var in CONFIGS:
        if kw.get("group"):
            ten = []
            for i, (m, n) in enumerate(kw["group"]):
                dy = (torch.randn(M, m, device=dev) * 0.01).half(); xx = torch.randn(M, n, device=dev).half()
                ten.append((dy, xx, torch.zeros(m, n, device=dev), torch.zeros(m, device=dev), kw.get("glu", 0) if (i == 1 and kw.get("glu")) else 0))
            descs = [ops.gemm_desc(dy, xx, gw, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb, variant=var, glu_rows=gl) for dy, xx, gw, gb, gl in ten]
            torch.cuda.synchronize()
            for _ in range(a.reps): ops.gemm_group(descs)
            torch.cuda.synchronize()
            print(name, ops.last_algo(), flush=True)
            continue
        if kw.get("tn"):
            dy = (torch.randn(M, N, device=dev) * 0.01).half(); xx = torch.randn(M, K, device=dev).half()
            gw = torch.zeros(N, K, device=dev); gb = torch.zeros(N, device=dev)
            torch.cuda.synchronize()
            for _ in range(a.reps): ops.gemm(dy, xx, gw, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb, variant=var)
            torch.cuda.synchronize()
            print(name, ops.last_algo(), flush=True)
            continue
        x = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
        act = kw.get("act", 0)
        cn = N // 2 if act == 7 else 2 * N if act == 8 else N                       # gated forms: C is (M, N/2) resp. (M, 2N)
        c = torch.empty(M, cn, dtype=torch.float16, device=dev)
        args = dict(variant=var)
        if kw.get("bias"): args["bias"] = torch.randn(N, device=dev)
        if kw.get("pre"): args["pre_act"] = torch.randn(M, 2 * N if act == 8 else N, device=dev).half()
        if act: args["act"] = act
        if kw.get("drop"): args["dropout"] = dr
        if kw.get("glu") and act in (7, 8): args["glu_rows"] = kw["glu"]
        torch.cuda.synchronize()
        for _ in range(a.reps): ops.gemm(x, w, c, **args)
        torch.cuda.synchronize()
        print(name, ops.last_algo(), flush=True)


if __name__ == "__main__":
    main()
