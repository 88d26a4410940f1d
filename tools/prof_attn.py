"""One attention configuration a few times (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalanalytical_amd import ops
dev, dh = "cuda:0", 64
B, H, Tq, Tk, p = 128, 8, 1024, 1024, float(os.environ.get("AFM_P", "0.1"))
d = H * dh
q = torch.randn(B * Tq, d, device=dev).bfloat16(); k = torch.randn(B * Tk, d, device=dev).bfloat16(); v = torch.randn(B * Tk, d, device=dev).bfloat16()
o = torch.empty_like(q); lse = torch.empty(B * H * Tq, device=dev); do = torch.randn_like(q)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v); delta = torch.empty_like(lse)
pad = torch.zeros(B, Tk, dtype=torch.uint8, device=dev)
shp = ops.attn_shape(B, H, Tq, Tk, dh, torch.bfloat16, d, d, d, d, pad, False, ops.drop(p, 1, 1))
for _ in range(3):
    ops.attn_fwd(shp, q, k, v, o, lse)
    ops.attn_bwd(shp, q, k, v, o, do, lse, delta, dq, dk, dv, d, d, d)
torch.cuda.synchronize()
