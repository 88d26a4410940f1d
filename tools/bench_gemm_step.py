"""Every GEMM of one c2 training micro-batch (B = 128, S = 1024, T = 128, d = 512, f = 2048) in the single-pass modes, with the
flags the engine passes, timed per launch next to the two bounds of the shape: MFMA time at the 2.5 PF dense peak and the HBM time
of the compulsory traffic at 8 TB/s.  `--variant N` forces a tile form (afm_gemm_desc.reserved) where the shape allows it."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalanalytical_amd import ops


def t(fn, it=20, warm=10):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16"); ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    cd = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    dev = "cuda:0"
    B, S, T, d, f = 128, 1024, 128, 512, 2048
    Me, Md = B * S, B * T
    dr = ops.drop(0.1, 1, 1)
    # (name, M, N, K, launches per micro-batch, kwargs builder)
    nt = [("enc qkv fwd", Me, 3 * d, d, 6, dict(bias=1)), ("enc out fwd", Me, d, d, 6, dict(bias=1)),
          ("enc ffn1 fwd +gelu+drop+sg", Me, f, d, 6, dict(bias=1, act=4, pre=1, drop=1)), ("enc ffn2 fwd", Me, d, f, 6, dict(bias=1)),
          ("enc out dgrad", Me, d, d, 6, {}), ("enc qkv dgrad", Me, d, 3 * d, 6, {}),
          ("enc ffn2 dgrad xsaved", Me, f, d, 6, dict(act=5, pre=1)), ("enc ffn1 dgrad", Me, d, f, 6, {}),
          ("dec mem kv fwd", Me, 2 * d, d, 6, dict(bias=1)), ("dec mem dgrad (all layers)", Me, d, 12 * d, 1, {}),
          ("dec qkv fwd", Md, 3 * d, d, 6, dict(bias=1)), ("dec d x d", Md, d, d, 30, dict(bias=1)),
          ("dec ffn1 fwd", Md, f, d, 6, dict(bias=1, act=4, pre=1, drop=1)), ("dec ffn2 fwd", Md, d, f, 6, dict(bias=1)),
          ("dec ffn2 dgrad", Md, f, d, 6, dict(act=5, pre=1)), ("dec ffn1 dgrad", Md, d, f, 6, {}), ("dec qkv dgrad", Md, d, 3 * d, 6, {})]
    tn = [("enc out wgrad", Me, d, d, 6), ("enc qkv wgrad", Me, 3 * d, d, 6), ("enc ffn2 wgrad", Me, d, f, 6), ("enc ffn1 wgrad", Me, f, d, 6),
          ("dec mem kv wgrad", Me, 2 * d, d, 6), ("dec d x d wgrad", Md, d, d, 18), ("dec qkv wgrad", Md, 3 * d, d, 6),
          ("dec ffn2 wgrad", Md, d, f, 6), ("dec ffn1 wgrad", Md, f, d, 6)]
    tot = tot_b = 0.0
    print(f"{'':34s} {'M':>7s} {'N':>5s} {'K':>6s}  {'ms':>7s} {'TF/s':>6s} {'mfma-bound':>10s} {'hbm-bound':>9s}  x/launches  algo")
    for name, M, N, K, cnt, kw in nt:
        if a.only and a.only not in name: continue
        x = torch.randn(M, K, device=dev).to(cd); w = torch.randn(N, K, device=dev).to(cd) * 0.05
        c = torch.empty(M, N, dtype=cd, device=dev)
        args = dict(variant=a.variant)
        if kw.get("bias"): args["bias"] = torch.randn(N, device=dev)
        if kw.get("pre"): args["pre_act"] = torch.randn(M, N, device=dev).to(cd)
        if kw.get("act"): args["act"] = kw["act"]
        if kw.get("drop"): args["dropout"] = dr
        try:
            ms = t(lambda: ops.gemm(x, w, c, **args))
        except Exception as e:      # noqa: BLE001
            print(f"{name:34s} {M:7d} {N:5d} {K:6d}  unsupported here: {e}"); continue
        flop = 2.0 * M * N * K
        byt = 2.0 * (M * K + N * K + M * N * (1 + (1 if kw.get("pre") else 0)))
        mb, hb = flop / 2.5e15 * 1e3, byt / 8e12 * 1e3
        tot += ms * cnt; tot_b += max(mb, hb) * cnt
        print(f"{name:34s} {M:7d} {N:5d} {K:6d}  {ms:7.3f} {flop / ms / 1e9:6.0f} {mb:10.3f} {hb:9.3f}  x{cnt:<3d} {ms * cnt:6.2f}  {ops.last_algo()}")
    for name, R, M, N, cnt in tn:
        if a.only and a.only not in name: continue
        dy = torch.randn(R, M, device=dev).to(cd); x = torch.randn(R, N, device=dev).to(cd)
        g = torch.zeros(M, N, device=dev); gb = torch.zeros(M, device=dev)
        ms = t(lambda: ops.gemm(dy, x, g, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb, variant=a.variant))
        flop = 2.0 * M * N * R
        byt = 2.0 * R * (M + N) + 8.0 * M * N
        mb, hb = flop / 2.5e15 * 1e3, byt / 8e12 * 1e3
        tot += ms * cnt; tot_b += max(mb, hb) * cnt
        print(f"{name:34s} {R:7d} {M:5d} {N:6d}  {ms:7.3f} {flop / ms / 1e9:6.0f} {mb:10.3f} {hb:9.3f}  x{cnt:<3d} {ms * cnt:6.2f}  {ops.last_algo()}")
    print(f"GEMMs per micro-batch: {tot:.2f} ms measured, {tot_b:.2f} ms at the per-shape bounds")


if __name__ == "__main__":
    main()
