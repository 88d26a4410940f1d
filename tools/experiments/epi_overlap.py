"""Round 6 probe: the FFN up-projection's save-grad epilogue (EPI 5: bias + GELU + dropout, two M x f outputs) and its data-gradient
partner (EPI 6: x stored factors) sit below BOTH roofs on 256 x 256 tiles with one workgroup per CU -- main loop (K = 512: 8 steps) and
epilogue take turns.  Variant 213: 128 x 128 tiles, four waves, two workgroups per CU (one's epilogue under the other's main loop);
214: 256 x 128, eight waves, 2 stages; 24: the loader-wave 256 x 128 kernel; 28 (default): 256 x 256.  Alternating rounds, one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops
from multimodalanalytical_amd.lib import ACT_GELU_SAVE_GRAD, ACT_MUL_SAVED


def t(fn, it=40, warm=30):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


dev, M = "cuda:0", 131072
dr = ops.drop(0.1, 1, 1)
for d, f in ((512, 2048), (768, 3072)):
    x = (torch.randn(M, d, device=dev) * 0.5).half()
    w = (torch.randn(f, d, device=dev) * 0.05).half(); bias = torch.randn(f, device=dev)
    g = torch.empty(M, f, dtype=torch.float16, device=dev); sg = torch.empty(M, f, dtype=torch.float16, device=dev)
    dy = (torch.randn(M, d, device=dev) * 0.01).half(); w2t = (torch.randn(f, d, device=dev) * 0.05).half()
    du = torch.empty(M, f, dtype=torch.float16, device=dev)
    ref = {}
    for name, fn in (("EPI5 fwd", lambda v: ops.gemm(x, w, g, bias=bias, act=ACT_GELU_SAVE_GRAD, pre_act=sg, dropout=dr, variant=v)),
                     ("EPI6 dgrad", lambda v: ops.gemm(dy, w2t, du, act=ACT_MUL_SAVED, pre_act=sg, variant=v))):
        res = {}
        for rnd in range(2):
            for v in ((0, 24, 213, 214) if rnd == 0 else (214, 213, 24, 0)):
                us = t(lambda: fn(v))
                res.setdefault(v, []).append(us)
                out = (g if name.startswith("EPI5") else du).float().abs().sum().item()
                ref.setdefault(name, out)
                assert abs(out - ref[name]) <= 1e-6 * abs(ref[name]), (name, v, out, ref[name])
        print(f"d {d} f {f} {name}: " + "  ".join(f"v{v} {min(r):6.1f}/{max(r):6.1f} us" for v, r in res.items()), flush=True)
