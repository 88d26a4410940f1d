"""Cross-attention kernels (T = 128 queries, S = 1 024 keys, B = 128, H = 8) against the K / V buffer layout: the row stride of the buffer
the per-head 128-byte K and V pieces are read from (and dK / dV written to) -- separate K and V matrices (1 KB rows), one layer's [K | V]
(2 KB), the engine's all-layers buffer [K0 | V0 | K1 | ... ] (12 KB rows).  Same kernels, same bytes; only the address pattern changes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops
from attn_m16 import t


def main():
    dev, dh, dt = "cuda:0", 64, torch.float16
    B, H, Tq, Tk = 128, 8, 128, 1024
    d = H * dh
    q = torch.randn(B * Tq, d, device=dev).to(dt)
    o = torch.empty(B * Tq, d, dtype=dt, device=dev); do = (torch.randn(B * Tq, d, device=dev) * 0.01).to(dt)
    dq = torch.empty_like(q)
    lse, delta = torch.empty(B * H * Tq, device=dev), torch.empty(B * H * Tq, device=dev)
    dr = ops.drop(0.1, 1, 3)
    kb = torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=dev)
    prod = 2.0 * B * H * Tq * Tk * dh
    cases = [("separate K, V (1 KB rows)", d, 0, None), ("[K | V] of one layer (2 KB rows)", 2 * d, 0, d),
             ("all six layers (12 KB rows), layer 2", 12 * d, 4 * d, 5 * d)]
    cases += [(f"all six layers, rows padded by {2 * pad} B, layer 2", 12 * d + pad, 4 * d, 5 * d) for pad in (64, 128, 256, 512, 1024, 2048)]
    for name, width, koff, voff in cases:
        if voff is None:
            kbuf = torch.randn(B * Tk, d, device=dev).to(dt); vbuf = torch.randn(B * Tk, d, device=dev).to(dt)
            k, v = kbuf, vbuf
            dkb, dvb = torch.empty_like(kbuf), torch.empty_like(vbuf)
            dk, dv = dkb, dvb
        else:
            buf = torch.randn(B * Tk, width, device=dev).to(dt); gb = torch.empty_like(buf)
            k, v = buf[:, koff:koff + d], buf[:, voff:voff + d]
            dk, dv = gb[:, koff:koff + d], gb[:, voff:voff + d]

        def shape(res):
            s = ops.attn_shape(B, H, Tq, Tk, dh, dt, ops._ld(q), ops._ld(k), ops._ld(v), ops._ld(o), None, False, dr)
            s.reserved = res
            return ops.attn_set_drop_bits(s, kb)
        s0, s1, s2 = shape(0), shape(1), shape(2)
        ops.attn_fwd(s0, q, k, v, o, lse)
        fns = {"fwd": lambda: ops.attn_fwd(s0, q, k, v, o, lse),
               "dQ": lambda: ops.attn_bwd(s1, q, k, v, o, do, lse, delta, dq, dk, dv, ops._ld(dq), ops._ld(dk), ops._ld(dv)),
               "dK/dV": lambda: ops.attn_bwd(s2, q, k, v, o, do, lse, delta, dq, dk, dv, ops._ld(dq), ops._ld(dk), ops._ld(dv))}
        fns["dQ"]()
        byt = {"fwd": 2 * B * Tk * d * 2 + 2 * B * Tq * d * 2, "dQ": 2 * B * Tk * d * 2 + 4 * B * Tq * d * 2, "dK/dV": 4 * B * Tk * d * 2 + 2 * B * Tq * d * 2}
        for rnd in range(2):
            ms = {n: t(f, it=60, warm=30) for n, f in fns.items()}
            print(f"{name}, round {rnd}: " + "   ".join(f"{n} {1e3 * ms[n]:.0f} us ({byt[n] / ms[n] / 1e9:.2f} TB/s of tensors)" for n in fns), flush=True)


if __name__ == "__main__":
    main()
