"""The XCD-aware 2-D tile walk of the persistent NT kernels (csrc/afm_gemm_mfma_impl.h tile_mn / nt_pick_xgc): c4's gated FFN launches and
the other wide products, timed per launch.  Run once as is and once with AFM_NT_XGC=1 (row-major walk everywhere) in separate processes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops


def t(fn, it=30, warm=20):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


dev, M = "cuda:0", 131072
dr = ops.drop(0.1, 1, 1)
print("AFM_NT_XGC =", os.environ.get("AFM_NT_XGC", "(auto)"))
for tag, d, f in (("c4", 768, 3072), ("c2", 512, 2048)):
    x = torch.randn(M, d, device=dev).half()
    if tag == "c4":
        w = (torch.randn(2 * f, d, device=dev) * 0.05).half(); bias = torch.randn(2 * f, device=dev)
        g = torch.empty(M, f, dtype=torch.float16, device=dev); uv = torch.empty(M, 2 * f, dtype=torch.float16, device=dev)
        ms = t(lambda: ops.gemm(x, w, g, bias=bias, act=7, pre_act=uv, dropout=dr, glu_rows=f))
        print(f"{tag} gated FFN up forward (EPI 8) {M}x{2 * f}x{d}: {ms:.4f} ms  [{ops.last_algo()}]", flush=True)
        ref = g.clone()
        dy = (torch.randn(M, d, device=dev) * 0.01).half(); w2t = (torch.randn(f, d, device=dev) * 0.05).half()
        duv = torch.empty(M, 2 * f, dtype=torch.float16, device=dev)
        ms = t(lambda: ops.gemm(dy, w2t, duv, act=8, pre_act=uv, glu_rows=f))
        print(f"{tag} gated FFN down dgrad (EPI 9) {M}x{f}x{d}: {ms:.4f} ms  [{ops.last_algo()}]", flush=True)
        print("checksums", float(ref.float().abs().sum()), float(duv.float().abs().sum()))
    for name, N, K in ((f"{tag} qkv fwd", 3 * d, d), (f"{tag} ffn1 plain", f, d), (f"{tag} ffn2 fwd", d, f)):
        a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half(); b = torch.randn(N, device=dev)
        c = torch.empty(M, N, dtype=torch.float16, device=dev)
        ms = t(lambda: ops.gemm(a, w, c, bias=b))
        print(f"{name} {M}x{N}x{K}: {ms:.4f} ms  [{ops.last_algo()}]", flush=True)
