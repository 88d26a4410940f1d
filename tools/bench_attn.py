"""Micro-benchmark of afm_attn_fwd/bwd at the training step's attention shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalanalytical_amd import ops
from bench_gemm import t

def main():
    dev, dh = "cuda:0", 64
    for name, B, H, Tq, Tk, causal, p in [("enc self", 128, 8, 1024, 1024, False, 0.0), ("enc self drop", 128, 8, 1024, 1024, False, 0.1),
                                           ("dec self", 128, 8, 128, 128, True, 0.1), ("dec cross", 128, 8, 128, 1024, False, 0.1)]:
        d = H * dh
        q = torch.randn(B * Tq, d, device=dev).bfloat16(); k = torch.randn(B * Tk, d, device=dev).bfloat16(); v = torch.randn(B * Tk, d, device=dev).bfloat16()
        o = torch.empty_like(q); lse = torch.empty(B * H * Tq, device=dev); do = torch.randn_like(q)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v); delta = torch.empty_like(lse)
        pad = torch.zeros(B, Tk, dtype=torch.uint8, device=dev)
        shp = ops.attn_shape(B, H, Tq, Tk, dh, torch.bfloat16, d, d, d, d, pad, causal, ops.drop(p, 1, 1))
        fl = 4.0 * B * H * Tq * Tk * dh * (0.5 if causal else 1.0)
        ms = t(lambda: ops.attn_fwd(shp, q, k, v, o, lse)); algo = ops.last_algo()
        msb = t(lambda: ops.attn_bwd(shp, q, k, v, o, do, lse, delta, dq, dk, dv, d, d, d))
        print(f"{name:14s} B{B} H{H} {Tq}x{Tk} {algo}: fwd {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s | bwd {msb:7.3f} ms {2.5*fl/msb/1e9:7.1f} TF/s (2.5x fwd flops)")

if __name__ == "__main__":
    main()
