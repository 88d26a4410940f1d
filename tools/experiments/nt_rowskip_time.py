"""dgrad-form NT GEMMs at the c3 mask density with and without afm_gemm_desc.k_live."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops


def t(fn, it=20, warm=10):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


dev = "cuda:0"
B, S = 128, 1024
M = B * S
lens = torch.randint(156, 816, (B,), generator=torch.Generator().manual_seed(2))
pad = torch.arange(S)[None, :] >= lens[:, None]
live = (~pad).view(B, S // 64, 64).any(-1).to(torch.uint8).reshape(-1).contiguous().to(dev)
print("dead 64-blocks", 1 - float(live.float().mean()), " dead 256-tiles", float((live.view(-1, 4).sum(1) == 0).float().mean()))
for name, N, K, act in (("out dgrad", 512, 512, 0), ("qkv dgrad", 512, 1536, 0), ("ffn2 dgrad x saved", 2048, 512, 5), ("ffn1 dgrad", 512, 2048, 0)):
    a = torch.randn(M, K, device=dev).half(); a[pad.reshape(-1).to(dev)] = 0
    w = (torch.randn(N, K, device=dev) * 0.05).half()
    c = torch.empty(M, N, dtype=torch.float16, device=dev)
    pre = torch.randn(M, N, device=dev).half() if act == 5 else None
    t0 = t(lambda: ops.gemm(a, w, c, act=act, pre_act=pre))
    t1 = t(lambda: ops.gemm(a, w, c, act=act, pre_act=pre, k_live=live))
    print(f"{name:22s} {t0:.3f} -> {t1:.3f} ms  [{ops.last_algo()}]")
