"""Self-attention backward over a padded batch (c3-like tails): dQ / dK-dV kernel times with and without the padded-query skip
(afm_attn_shape.reserved bit 6)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops


def t(fn, it=20, warm=20):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


dev = "cuda:0"
B, H, T, dh = 128, 8, 1024, 64
D = H * dh
g = torch.Generator(device=dev).manual_seed(1)
qkv = torch.randn(B * T, 3 * D, device=dev, generator=g).half()
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
lens = torch.randint(56 + 100, 56 + 760, (B,), generator=torch.Generator().manual_seed(2))
order = os.environ.get("ORDER", "random")      # the batch order the kernels' workgroups are dispatched in: random | desc | asc
if order == "desc": lens = torch.sort(lens, descending=True).values
if order == "asc": lens = torch.sort(lens).values
print("batch order:", order)
pad = (torch.arange(T)[None, :] >= lens[:, None])
print("padded fraction", float(pad.float().mean()))
kp = pad.to(torch.uint8).to(dev).contiguous()
do = torch.randn(B * T, D, device=dev, generator=g).half() * 0.01
do[pad.reshape(-1).to(dev)] = 0
o = torch.empty(B * T, D, dtype=torch.float16, device=dev); lse = torch.empty(B * H * T, device=dev)
dqkv = torch.empty(B * T, 3 * D, dtype=torch.float16, device=dev)
dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
dr = ops.drop(0.1, 1, 3)
bits = torch.zeros(ops.attn_drop_bits_words(B, H, T, T), dtype=torch.int64, device=dev)
for flag in (0, 64):
    for which, name in ((1, "dq"), (2, "dkv")):
        s = ops.attn_shape(B, H, T, T, dh, torch.float16, 3 * D, 3 * D, 3 * D, D, kp, False, dr)
        ops.attn_set_drop_bits(s, bits)
        ops.attn_fwd(s, q, k, v, o, lse)
        s.reserved = which | flag
        ms = t(lambda: ops.attn_bwd(s, q, k, v, o, do, lse, torch.empty_like(lse), dq, dk, dv, 3 * D, 3 * D, 3 * D))
        print(f"flag {flag:2d} {name:3s} {ms:.3f} ms")
ms = t(lambda: ops.attn_fwd(s, q, k, v, o, lse))
print(f"fwd {ms:.3f} ms")
