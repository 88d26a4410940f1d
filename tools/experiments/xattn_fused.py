"""Cross-attention backward (128 queries x Tk keys): the two general kernels against the fused short-query kernel (reserved bit 18),
with and without keep-bit dropout, over an unpadded and a c3-like padded memory.  `python tools/experiments/xattn_fused.py [H16|BF16]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalanalytical_amd import ops
from bench_gemm import t

def main():
    dev, dh = "cuda:0", 64
    dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "BF16") else torch.float16
    for name, B, H, Tq, Tk, live, p in [("c2 cross", 128, 8, 128, 1024, 1.0, 0.1), ("c2 cross nodrop", 128, 8, 128, 1024, 1.0, 0.0),
                                        ("c3 cross (47% live)", 128, 8, 128, 1024, 0.47, 0.1), ("c4 cross", 32, 12, 128, 1024, 0.57, 0.1),
                                        ("c5 cross", 256, 8, 128, 56, 1.0, 0.1)]:
        d = H * dh
        g = torch.Generator().manual_seed(1)
        q = (torch.randn(B * Tq, d, generator=g) * 0.5).to(dev).to(dt)
        kv = (torch.randn(B * Tk, 2 * d, generator=g) * 0.5).to(dev).to(dt)
        do = (torch.randn(B * Tq, d, generator=g) * 0.1).to(dev).to(dt)
        n = (torch.rand(B, generator=g) * 0.3 + live - 0.15).clamp(0.02, 1.0) * Tk if live < 1 else torch.full((B,), float(Tk))
        pad = (torch.arange(Tk)[None, :] >= n.long()[:, None]).to(torch.uint8).to(dev)
        o = torch.empty_like(q); lse = torch.empty(B * H * Tq, device=dev)
        dq = torch.empty_like(q); dkv = torch.empty_like(kv); delta = torch.empty_like(lse)
        res = {}
        for flag in (0, 262144):
            shp = ops.attn_shape(B, H, Tq, Tk, dh, dt, d, 2 * d, 2 * d, d, pad, False, ops.drop(p, 1, 1) if p else ops.NO_DROP)
            if p:
                ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=dev))
            ops.attn_fwd(shp, q, kv[:, :d], kv[:, d:], o, lse)
            shp.reserved |= flag
            ms = t(lambda: ops.attn_bwd(shp, q, kv[:, :d], kv[:, d:], o, do, lse, delta, dq, dkv[:, :d], dkv[:, d:], d, 2 * d, 2 * d))
            res[flag] = (ms, ops.last_algo())
        fl = 10.0 * B * H * Tq * float((pad == 0).sum()) / B * dh
        print(f"{name:22s} B{B} H{H} {Tq}x{Tk}: two kernels {res[0][0]*1e3:7.1f} us ({res[0][1]}) | fused {res[262144][0]*1e3:7.1f} us ({res[262144][1]}) "
              f"{fl/res[262144][0]/1e9:6.1f} TF/s of the 5 live products", flush=True)

if __name__ == "__main__" and "--abl" not in sys.argv and "--pmc" not in sys.argv:
    main()


def ablate():
    """AFM_ATTN_ABLATIONS build (AFM_LIB_OVERRIDE): what each part of the fused kernel costs at the c2 cross shape (times only)."""
    dev, dh, B, H, Tq, Tk, p = "cuda:0", 64, 128, 8, 128, 1024, 0.1
    d = H * dh
    g = torch.Generator().manual_seed(1)
    q = (torch.randn(B * Tq, d, generator=g) * 0.5).to(dev).half()
    kv = (torch.randn(B * Tk, 2 * d, generator=g) * 0.5).to(dev).half()
    do = (torch.randn(B * Tq, d, generator=g) * 0.1).to(dev).half()
    pad = torch.zeros(B, Tk, dtype=torch.uint8, device=dev)
    o = torch.empty_like(q); lse = torch.empty(B * H * Tq, device=dev)
    dq = torch.empty_like(q); dkv = torch.empty_like(kv); delta = torch.empty_like(lse)
    shp = ops.attn_shape(B, H, Tq, Tk, dh, torch.float16, d, 2 * d, 2 * d, d, pad, False, ops.drop(p, 1, 1))
    ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=dev))
    ops.attn_fwd(shp, q, kv[:, :d], kv[:, d:], o, lse)
    names = {0: "full", 1: "no dK/dV stores", 2: "no phase B", 3: "no phase B, no stores", 4: "no P/dS LDS stores", 7: "1+2+4", 8: "no dQ product",
             15: "1+2+4+8", 16: "no exp2/dropout/mask", 23: "1+2+4+16", 31: "all off (S, dP products + barriers + DMA)"}
    for abl, nm in names.items():
        shp.reserved = 262144 | (abl << 20)
        ms = t(lambda: ops.attn_bwd(shp, q, kv[:, :d], kv[:, d:], o, do, lse, delta, dq, dkv[:, :d], dkv[:, d:], d, 2 * d, 2 * d))
        print(f"abl {abl:3d} {nm:44s} {ms*1e3:7.1f} us", flush=True)


def pmc():
    """A few launches of both forms at the c2 cross shape (for rocprofv3 --pmc: tools/experiments/fsq_pmc.sh)."""
    dev, dh, B, H, Tq, Tk, p = "cuda:0", 64, 128, 8, 128, 1024, 0.1
    d = H * dh
    g = torch.Generator().manual_seed(1)
    q = (torch.randn(B * Tq, d, generator=g) * 0.5).to(dev).half()
    kv = (torch.randn(B * Tk, 2 * d, generator=g) * 0.5).to(dev).half()
    do = (torch.randn(B * Tq, d, generator=g) * 0.1).to(dev).half()
    pad = torch.zeros(B, Tk, dtype=torch.uint8, device=dev)
    o = torch.empty_like(q); lse = torch.empty(B * H * Tq, device=dev)
    dq = torch.empty_like(q); dkv = torch.empty_like(kv); delta = torch.empty_like(lse)
    shp = ops.attn_shape(B, H, Tq, Tk, dh, torch.float16, d, 2 * d, 2 * d, d, pad, False, ops.drop(p, 1, 1))
    ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=dev))
    ops.attn_fwd(shp, q, kv[:, :d], kv[:, d:], o, lse)
    for flag in (0, 262144):
        shp.reserved = flag
        for _ in range(3):
            ops.attn_bwd(shp, q, kv[:, :d], kv[:, d:], o, do, lse, delta, dq, dkv[:, :d], dkv[:, d:], d, 2 * d, 2 * d)
    torch.cuda.synchronize()


if __name__ == "__main__" and "--abl" in sys.argv:
    ablate()
if __name__ == "__main__" and "--pmc" in sys.argv:
    pmc()
