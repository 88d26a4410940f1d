"""The dQ kernel alone (afm_attn_shape.reserved & 3 == 1) at the c2 encoder and cross shapes, keep-bit dropout: for A / B runs of library builds
(AFM_LIB_OVERRIDE), alternating processes (tools/experiments/r6_dq_saddr.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalanalytical_amd import ops
from bench_gemm import t

dev, dh = "cuda:0", 64
for name, B, H, Tq, Tk, p in [("enc self", 128, 8, 1024, 1024, 0.1), ("enc self nodrop", 128, 8, 1024, 1024, 0.0), ("cross", 128, 8, 128, 1024, 0.1), ("c4 enc self", 32, 12, 1024, 1024, 0.1)]:
    d = H * dh
    g = torch.Generator().manual_seed(1)
    q = (torch.randn(B * Tq, d, generator=g) * 0.5).to(dev).half()
    kv = (torch.randn(B * Tk, 2 * d, generator=g) * 0.5).to(dev).half()
    do = (torch.randn(B * Tq, d, generator=g) * 0.1).to(dev).half()
    pad = torch.zeros(B, Tk, dtype=torch.uint8, device=dev)
    o = torch.empty_like(q); lse = torch.empty(B * H * Tq, device=dev)
    dq = torch.empty_like(q); dkv = torch.empty_like(kv); delta = torch.empty_like(lse)
    shp = ops.attn_shape(B, H, Tq, Tk, dh, torch.float16, d, 2 * d, 2 * d, d, pad, False, ops.drop(p, 1, 1) if p else ops.NO_DROP)
    if p:
        ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=dev))
    ops.attn_fwd(shp, q, kv[:, :d], kv[:, d:], o, lse)
    base = shp.reserved
    forms = (("32x32x16", 32768), ("16x16x32 occ3", 1024), ("16x16x32 occ2", 1024 | 2048)) if "--forms" in sys.argv else (("32x32x16", 32768),)
    for fname, fl in forms:
        shp.reserved = base | 1 | fl
        ms = t(lambda: ops.attn_bwd(shp, q, kv[:, :d], kv[:, d:], o, do, lse, delta, dq, dkv[:, :d], dkv[:, d:], d, 2 * d, 2 * d))
        print(f"dQ {name:16s} {fname:14s} {ms*1e3:7.1f} us  checksum {float(dq.float().abs().sum()):.6e}", flush=True)
