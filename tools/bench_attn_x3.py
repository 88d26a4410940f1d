"""Time the attention kernels of one encoder self-attention call (C2 shape by default) in a precision mode."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalanalytical_amd import ops
from multimodalanalytical_amd.x2 import X2


def t(fn, iters=20, warm=30):      # (a kernel timed right after process start reads ~15 % slow: warm the clocks up)
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="bf16x3")
    ap.add_argument("--B", type=int, default=128); ap.add_argument("--H", type=int, default=8)
    ap.add_argument("--S", type=int, default=1024); ap.add_argument("--Tq", type=int, default=0)
    ap.add_argument("--p", type=float, default=0.1); ap.add_argument("--causal", type=int, default=0)
    ap.add_argument("--bits", type=int, default=1, help="1: forward stores the keep-bit tensor, backward reads it; 0: re-hash")
    ap.add_argument("--old", type=int, default=0, help="1: also time the 8-wave staggered forms (afm_attn_shape.reserved | 16)")
    a = ap.parse_args()
    dev = "cuda:0"
    cd = {"bf16": torch.bfloat16, "fp16": torch.float16, "bf16x3": X2.dtype, "fp32": torch.float32}[a.mode]
    B, H, S, dh = a.B, a.H, a.S, 64
    Tq = a.Tq or S
    d = H * dh

    def rnd(r, c, sc=1.0):
        x = torch.randn(r, c, device=dev) * sc
        return x if cd == torch.float32 else ops.convert(x, ops.empty(r, c, cd, dev))
    if Tq == S:
        qkv = rnd(B * S, 3 * d); q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        dqkv = ops.empty(B * S, 3 * d, cd, dev); dq, dk, dv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
    else:
        q = rnd(B * Tq, d); kv = rnd(B * S, 2 * d); k, v = kv[:, :d], kv[:, d:]
        dq = ops.empty(B * Tq, d, cd, dev); dkv = ops.empty(B * S, 2 * d, cd, dev); dk, dv = dkv[:, :d], dkv[:, d:]
    o = ops.empty(B * Tq, d, cd, dev); do = rnd(B * Tq, d, 0.01)
    lse = torch.empty(B * H * Tq, device=dev); delta = torch.empty_like(lse)
    dr = ops.drop(a.p, 1, 3)
    prod = 2.0 * B * H * Tq * S * dh

    def shp(res):
        s = ops.attn_shape(B, H, Tq, S, dh, cd, ops._ld(q), ops._ld(k), ops._ld(v), ops._ld(o), None, bool(a.causal), dr)
        s.reserved = res
        return ops.attn_set_drop_bits(s, bits)
    bits = torch.zeros(ops.attn_drop_bits_words(B, H, Tq, S), dtype=torch.int64, device=dev) if a.bits and a.p > 0 else None
    s0, s1, s2 = shp(0), shp(1), shp(2)
    passes = 3 if a.mode == "bf16x3" else 1
    cases = []
    if bits is not None and a.mode in ("fp16", "bf16"):     # keep bits filled ahead + the forward that reads them
        sr = shp(0)
        cases.append(("fill", lambda: ops.attn_fill_drop_bits(sr), 0))
        cases.append(("fwdR", lambda: ops.attn_fwd(sr, q, k, v, o, lse), 2))
    if a.old:
        s8, s9 = shp(16), shp(17)
        cases.append(("fwd8", lambda: ops.attn_fwd(s8, q, k, v, o, lse), 2))
        cases.append(("dq8", lambda: ops.attn_bwd(s9, q, k, v, o, do, lse, delta, dq, dk, dv, ops._ld(dq), ops._ld(dk), ops._ld(dv)), 3))
    for name, fn, np_ in cases + [("fwd", lambda: ops.attn_fwd(s0, q, k, v, o, lse), 2),
                          ("dq", lambda: ops.attn_bwd(s1, q, k, v, o, do, lse, delta, dq, dk, dv, ops._ld(dq), ops._ld(dk), ops._ld(dv)), 3),
                          ("dkv", lambda: ops.attn_bwd(s2, q, k, v, o, do, lse, delta, dq, dk, dv, ops._ld(dq), ops._ld(dk), ops._ld(dv)), 4)]:
        ms = t(fn)
        print(f"{a.mode} {name:4s} [{ops.last_algo()}] {ms:.3f} ms  {np_ * prod / ms / 1e9:.0f} TF/s algorithmic, "
              f"{passes * np_ * prod / ms / 1e9 / 2500:.3f} of MFMA peak executed")


if __name__ == "__main__":
    main()
