"""The software-pipelined dK/dV kernel (csrc/afm_attn_pipe_impl.h) against the round-3 kernel (afm_attn_shape.reserved & 128):
bit-for-bit agreement of dK and dV over dense, padded, short and dropout-free cases, then the times at the c2 encoder shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops


def t(fn, it=30, warm=30):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


dev = "cuda:0"


def case(B, H, Tq, Tk, p, padded, qskip, seed=1, time_it=False, dtype=torch.float16):
    dh = 64
    D = H * dh
    g = torch.Generator(device=dev).manual_seed(seed)
    self_attn = Tq == Tk
    q = torch.randn(B * Tq, D, device=dev, generator=g).to(dtype)
    k = torch.randn(B * Tk, D, device=dev, generator=g).to(dtype)
    v = torch.randn(B * Tk, D, device=dev, generator=g).to(dtype)
    kp = None
    do = (torch.randn(B * Tq, D, device=dev, generator=g) * 0.01).to(dtype)
    if padded:
        lens = torch.randint(max(1, Tk // 8), Tk + 1, (B,), generator=torch.Generator().manual_seed(seed + 1))
        pad = (torch.arange(Tk)[None, :] >= lens[:, None])
        kp = pad.to(torch.uint8).to(dev).contiguous()
        if self_attn and qskip: do[pad.reshape(-1).to(dev)] = 0
    o = torch.empty(B * Tq, D, dtype=dtype, device=dev); lse = torch.empty(B * H * Tq, device=dev)
    dr = ops.drop(p, 1, 3) if p > 0 else ops.NO_DROP
    s = ops.attn_shape(B, H, Tq, Tk, dh, dtype, D, D, D, D, kp, False, dr)
    if p > 0:
        bits = torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=dev)
        ops.attn_set_drop_bits(s, bits)
    ops.attn_fwd(s, q, k, v, o, lse)
    delta = torch.empty_like(lse)
    flag = 64 if (qskip and self_attn and padded) else 0
    out = {}
    for tag, extra in (("r3", 128), ("pipe32", 0), ("pipe64", 512), ("pipe8w", 256)):
        dq = torch.zeros(B * Tq, D, dtype=dtype, device=dev)
        dk = torch.full((B * Tk, D), float("nan"), dtype=dtype, device=dev); dv = torch.full_like(dk, float("nan"))
        s.reserved = flag | extra
        ops.attn_bwd(s, q, k, v, o, do, lse, delta, dq, dk, dv, D, D, D)
        torch.cuda.synchronize()
        out[tag] = (dk, dv)
    same = all(torch.equal(out["r3"][i], out[n][i]) for i in (0, 1) for n in ("pipe64", "pipe32", "pipe8w"))
    fin = bool(torch.isfinite(out["pipe64"][0]).all() and torch.isfinite(out["pipe64"][1]).all())
    md = max((out["r3"][0].float() - out["pipe64"][0].float()).abs().max().item(), (out["r3"][1].float() - out["pipe64"][1].float()).abs().max().item())
    print(f"B{B} H{H} Tq{Tq} Tk{Tk} p{p} padded={padded} qskip={qskip} {str(dtype)[6:]}: identical={same} finite={fin} maxdiff={md:.3e}", flush=True)
    if time_it:
        for rnd in range(2):
            for tag, extra in (("r3", 128), ("pipe32", 0), ("pipe64", 512), ("pipe8w", 256)):
                s.reserved = 2 | flag | extra
                ms = t(lambda: ops.attn_bwd(s, q, k, v, o, do, lse, delta, dq, dk, dv, D, D, D))
                print(f"   dK/dV {tag:5s} {ms:.4f} ms", flush=True)
    return same and fin


if "--abl" in sys.argv:      # an AFM_ATTN_ABLATIONS build (AFM_LIB_OVERRIDE): times only, the results are wrong by construction
    B, H, T, D = 128, 8, 1024, 512
    g = torch.Generator(device=dev).manual_seed(1)
    q, k, v = (torch.randn(B * T, D, device=dev, generator=g).half() for _ in range(3))
    do = (torch.randn(B * T, D, device=dev, generator=g) * 0.01).half()
    o = torch.empty_like(q); lse = torch.empty(B * H * T, device=dev); delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    s = ops.attn_shape(B, H, T, T, 64, torch.float16, D, D, D, D, None, False, ops.drop(0.1, 1, 3))
    ops.attn_set_drop_bits(s, torch.zeros(ops.attn_drop_bits_words(B, H, T, T), dtype=torch.int64, device=dev))
    ops.attn_fwd(s, q, k, v, o, lse)
    s.reserved = 0; ops.attn_bwd(s, q, k, v, o, do, lse, delta, dq, dk, dv, D, D, D)
    names = {0: "full", 1: "no mfma", 2: "no arith", 4: "no slot reads", 8: "no barrier", 16: "no preamble reads", 32: "no dma", 6: "mfma + skeleton",
             3: "reads + skeleton", 7: "skeleton", 23: "skeleton - preamble", 31: "skeleton - preamble - barrier", 63: "loop only", 22: "mfma + skeleton - preamble",
             54: "mfma + skeleton - preamble - dma", 64: "no slot waits", 80: "no slot waits, no preamble", 112: "no waits / preamble / dma",
             120: "no waits / preamble / dma / barrier"}
    for rnd in range(2):
        for abl in (0, 1, 2, 4, 8, 16, 32, 64, 80, 112, 120, 6, 22, 54, 3, 7, 23, 31, 63):
            s.reserved = 2 | (abl << 20)
            ms = t(lambda: ops.attn_bwd(s, q, k, v, o, do, lse, delta, dq, dk, dv, D, D, D))
            print(f"abl {abl:3d} {names[abl]:36s} {ms:.4f} ms", flush=True)
    sys.exit(0)
ok = True
ok &= case(2, 2, 64, 128, 0.1, False, False)
ok &= case(2, 2, 128, 128, 0.1, False, False)
ok &= case(2, 8, 256, 256, 0.1, False, False)
ok &= case(3, 8, 192, 320, 0.1, True, False)
ok &= case(4, 8, 512, 512, 0.1, True, True)
ok &= case(4, 8, 512, 512, 0.0, True, True)
ok &= case(2, 8, 1024, 1024, 0.0, False, False)
ok &= case(2, 8, 1024, 1024, 0.1, False, False, dtype=torch.bfloat16)
ok &= case(16, 8, 128, 1024, 0.1, True, False)      # decoder cross-attention shape
for rep in range(3):
    ok &= case(8, 8, 1024, 1024, 0.1, True, True, seed=10 + rep)
print("ALL OK" if ok else "FAILURES", flush=True)
case(128, 8, 1024, 1024, 0.1, False, False, time_it=True)
case(128, 8, 1024, 1024, 0.1, True, True, time_it=True)
case(128, 8, 128, 1024, 0.1, False, False, time_it=True)
