"""Cross-attention shapes of the training step (decoder queries T = 128 against S = 1 024 memory keys): wall times of the forward, the dQ
kernel and the dK/dV kernel forms, alternating in one process.  The dK/dV kernels give a workgroup 128 keys and loop over 64-query
tiles -- TWO tiles here -- so the per-workgroup prologue (fragment loads, masks, tile list, ring fill) is most of a workgroup's life."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops
from attn_m16 import t, _shape


def check():
    """dK / dV of the short-query kernel (reserved & 65536) against the general kernels over ragged / padded / dropout cases."""
    dev, dh = "cuda:0", 64
    ok = True
    for (B, H, Tq, Tk, p, pad, dt) in [(2, 4, 128, 1024, 0.1, False, torch.float16), (3, 2, 128, 1000, 0.1, True, torch.float16), (2, 2, 100, 520, 0.1, True, torch.float16),
                                       (2, 4, 64, 256, 0.0, False, torch.float16), (2, 2, 192, 768, 0.1, True, torch.float16), (1, 3, 160, 300, 0.0, True, torch.float16),
                                       (2, 8, 128, 1024, 0.1, True, torch.bfloat16), (5, 8, 40, 2048, 0.1, True, torch.float16), (128, 8, 128, 1024, 0.1, False, torch.float16)]:
        d = H * dh
        gen = torch.Generator(device=dev).manual_seed(B * 1000 + Tq)
        rnd = lambda r, c, sc=1.0: (torch.randn(r, c, device=dev, generator=gen) * sc).to(dt)
        q, k, v = rnd(B * Tq, d), rnd(B * Tk, d), rnd(B * Tk, d)
        o, do = torch.empty(B * Tq, d, dtype=dt, device=dev), rnd(B * Tq, d, 0.05)
        lse, delta = torch.empty(B * H * Tq, device=dev), torch.empty(B * H * Tq, device=dev)
        kp = None
        if pad:
            n = torch.randint(1, Tk + 1, (B,), device=dev, generator=gen)
            n[0] = Tk // 3
            kp = (torch.arange(Tk, device=dev)[None, :] >= n[:, None]).to(torch.uint8).contiguous()
        dr = ops.drop(p, 11, 3) if p > 0 else ops.NO_DROP
        kb = torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=dev) if p > 0 else None
        mk = lambda res: ops.attn_set_drop_bits(_shape(B, H, Tq, Tk, dh, dt, q, k, v, o, kp, False, dr, res), kb)
        ops.attn_fwd(mk(0), q, k, v, o, lse)
        res = []
        for flag in (0, 65536):
            dq = torch.empty_like(q); dk = torch.full_like(k, float("nan")); dv = torch.full_like(v, float("nan"))
            ops.attn_bwd(mk(flag), q, k, v, o, do, lse, delta, dq, dk, dv, ops._ld(dq), ops._ld(dk), ops._ld(dv))
            res.append((dk.float(), dv.float()))
        ek = float((res[0][0] - res[1][0]).abs().max() / res[0][0].abs().max()); ev = float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max())
        tol = 4e-3 if dt == torch.float16 else 3e-2
        good = bool(torch.isfinite(res[1][0]).all() and torch.isfinite(res[1][1]).all()) and ek < tol and ev < tol
        ok &= good
        print(f"B{B} H{H} Tq{Tq} Tk{Tk} p={p} pad={pad} {dt}: short-q dK {ek:.2e} dV {ev:.2e} {'ok' if good else 'FAIL'}", flush=True)
    print("ALL OK" if ok else "FAILURES", flush=True)


def main():
    if "--time-only" not in sys.argv:
        check()
    dev, dh, dt = "cuda:0", 64, torch.float16
    for (B, H, Tq, Tk, padfrac) in ((128, 8, 128, 1024, 0.0), (128, 12, 128, 1024, 0.0), (128, 8, 128, 1024, 0.5), (128, 8, 256, 56, 0.0)):
        d = H * dh
        q = torch.randn(B * Tq, d, device=dev).to(dt); kv = torch.randn(B * Tk, 2 * d, device=dev).to(dt)
        k, v = kv[:, :d], kv[:, d:]
        o = torch.empty(B * Tq, d, dtype=dt, device=dev); do = (torch.randn(B * Tq, d, device=dev) * 0.01).to(dt)
        dq = torch.empty_like(q); dkv = torch.empty_like(kv); dk, dv = dkv[:, :d], dkv[:, d:]
        lse, delta = torch.empty(B * H * Tq, device=dev), torch.empty(B * H * Tq, device=dev)
        kp = None
        if padfrac > 0:
            n = torch.randint(int(Tk * (1 - 2 * padfrac)) + 1, Tk + 1, (B,), device=dev)
            kp = (torch.arange(Tk, device=dev)[None, :] >= n[:, None]).to(torch.uint8).contiguous()
        dr = ops.drop(0.1, 1, 3)
        kb = torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=dev)
        mk = lambda res: ops.attn_set_drop_bits(_shape(B, H, Tq, Tk, dh, dt, q, k, v, o, kp, False, dr, res), kb)
        ops.attn_fwd(mk(0), q, k, v, o, lse)
        prod = 2.0 * B * H * Tq * Tk * dh
        forms = {"fwd": None, "dQ": mk(1), "dQ 32x32x16": mk(1 | 32768), "dQ 16x16x32 occ2": mk(1 | 1024 | 2048),
                 "dkv default": mk(2), "dkv pipe 32x32x16": mk(2 | 16384), "dkv round-3": mk(2 | 128), "dkv round-3 on 16x16x32": mk(2 | 4096),
                 "dkv short-q": mk(2 | 65536)}
        fw = mk(0)
        fns = {n: ((lambda: ops.attn_fwd(fw, q, k, v, o, lse)) if sh is None else
                   (lambda sh=sh: ops.attn_bwd(sh, q, k, v, o, do, lse, delta, dq, dk, dv, ops._ld(dq), ops._ld(dk), ops._ld(dv)))) for n, sh in forms.items()}
        fns["dQ"]()      # delta for the dK/dV-only calls
        nprod = {"fwd": 2, "dQ": 3}
        for rnd in range(2):
            order = list(fns) if rnd % 2 == 0 else list(fns)[::-1]
            ms = {n: t(fns[n], it=60, warm=30) for n in order}
            print(f"B{B} H{H} Tq{Tq} Tk{Tk} pad {padfrac}, round {rnd}: " +
                  "   ".join(f"{n} {1e3 * ms[n]:.0f} us ({nprod.get(n.split()[0], 4) * prod / ms[n] / 1e9:.0f} TF/s)" for n in fns), flush=True)


if __name__ == "__main__":
    main()
