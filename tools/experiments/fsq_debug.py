import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops
DEV = "cuda:0"
B, H, dh = 8, 8, 64
d = H * dh
Tq, Tk = int(sys.argv[1]), int(sys.argv[2])
lens = (1024, 1, 63, 65, 512, 200, 999, 128)
dtype = torch.float16
p = 0.1
g = torch.Generator().manual_seed(3)
n = torch.tensor(lens).clamp(max=Tk)
pad = (torch.arange(Tk)[None, :] >= n[:, None]).to(torch.uint8).to(DEV)
q = (torch.randn(B * Tq, d, generator=g) * 0.5).to(DEV).to(dtype)
kv = (torch.randn(B * Tk, 2 * d, generator=g) * 0.5).to(DEV).to(dtype)
do = (torch.randn(B * Tq, d, generator=g) * 0.1).to(DEV).to(dtype)
drop = ops.drop(p, 4, 1)
def run(flag):
    shp = ops.attn_shape(B, H, Tq, Tk, dh, dtype, d, 2 * d, 2 * d, d, pad, False, drop)
    ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=DEV))
    o = torch.full((B * Tq, d), 3.0, dtype=dtype, device=DEV)
    lse = torch.full((B * H * Tq,), 3.0, device=DEV)
    ops.attn_fwd(shp, q, kv[:, :d], kv[:, d:], o, lse)
    shp.reserved |= flag if flag else 32768
    dq = torch.full((B * Tq, d), 3.0, dtype=dtype, device=DEV)
    dkv = torch.full((B * Tk, 2 * d), float("nan"), dtype=dtype, device=DEV)
    delta = torch.full_like(lse, 7.0)
    ops.attn_bwd(shp, q, kv[:, :d], kv[:, d:], o, do, lse, delta, dq, dkv[:, :d], dkv[:, d:], d, 2 * d, 2 * d)
    print(ops.last_algo())
    return dkv.float().view(B, Tk, 2, H, dh)
a, b = run(0), run(262144)
e = (a - b).abs()
print("max", float(e.max()), "ref max", float(a.abs().max()))
for which in (0, 1):
    ee = e[:, :, which]
    print("dK" if which == 0 else "dV", "per sample", [round(float(x), 5) for x in ee.amax((1, 2, 3))])
    print("   per head", [round(float(x), 5) for x in ee.amax((0, 1, 3))])
    print("   per key tile(64) of sample with max", [round(float(x), 5) for x in ee[int(ee.amax((1, 2, 3)).argmax())].view(Tk // 8, 8, H, dh).amax((1, 2, 3))[:16]])
    print("   per column", [round(float(x), 4) for x in ee.amax((0, 1, 2))])
    print("   count >1e-4", int((ee > 1e-4).sum()), "of", ee.numel())
