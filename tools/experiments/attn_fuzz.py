"""Random-shape sweep of the single-pass attention path (algo 2: forward, dQ, dK/dV defaults of round 5 -- the 16x16x32 dK/dV pipeline, the
16x16x32 dQ kernel where no keep bits are read) against the generic exact-fp32 kernels (algo 1) on the same fp16 inputs and the same
dropout stream: O, lse, dQ, dK, dV.  Shapes are drawn around the kernels' branch points: Tq a multiple of 64 or not, Tk ragged / < 64 /
not a multiple of 32, B * H a multiple of 8 or not (XCD map), padded key blocks, whole padded tiles, causal self-attention, dropout with
and without the keep-bit tensor, the padded-query skip.  Not part of the test suite (minutes): a one-off check of new default kernels."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops


def run(B, H, Tq, Tk, causal, padmode, p, bits, qskip, dt, seed):
    dev, dh = "cuda:0", 64
    d = H * dh
    g = torch.Generator(device=dev).manual_seed(seed)
    rnd = lambda r, c, sc=1.0: (torch.randn(r, c, device=dev, generator=g) * sc).to(dt)
    q, k, v, do = rnd(B * Tq, d), rnd(B * Tk, d), rnd(B * Tk, d), rnd(B * Tq, d, 0.1)
    kp = None
    if padmode:
        n = torch.randint(1, Tk + 1, (B,), device=dev, generator=g)
        if padmode == 2 and Tk >= 192:
            n[0] = Tk - 130          # more than two whole 64-key tiles of padding
        kpb = torch.arange(Tk, device=dev)[None, :] >= n[:, None]
        kp = kpb.to(torch.uint8).contiguous()
        if qskip:
            do = do.clone(); do[kpb.reshape(-1)] = 0
    dr = ops.drop(p, 1234 + seed, 3) if p > 0 else ops.NO_DROP
    out = {}
    for algo in (1, 2):
        o = torch.full((B * Tq, d), float("nan"), dtype=dt, device=dev); lse = torch.full((B * H * Tq,), float("nan"), device=dev)
        shp = ops.attn_shape(B, H, Tq, Tk, dh, dt, d, d, d, d, kp, causal, dr, algo=algo)
        if algo == 2 and p > 0 and bits:
            ops.attn_set_drop_bits(shp, torch.zeros(ops.attn_drop_bits_words(B, H, Tq, Tk), dtype=torch.int64, device=dev))
        ops.attn_fwd(shp, q, k, v, o, lse)
        if algo == 2 and qskip:
            shp.reserved |= 64
        dq, dk, dv = (torch.full((n_, d), float("nan"), dtype=dt, device=dev) for n_ in (B * Tq, B * Tk, B * Tk))
        ops.attn_bwd(shp, q, k, v, o, do, lse, torch.empty_like(lse), dq, dk, dv, d, d, d)
        out[algo] = (o.float(), lse, dq.float(), dk.float(), dv.float(), ops.last_algo())
    a, b = out[1], out[2]
    tol = 6e-3 if dt == torch.float16 else 4e-2
    errs = []
    for i, nm in ((0, "O"), (2, "dQ"), (3, "dK"), (4, "dV")):
        ok = bool(torch.isfinite(b[i]).all())
        e = float((a[i] - b[i]).abs().max() / a[i].abs().max().clamp_min(1e-3)) if ok else float("inf")      # (a single live key: dQ = dK = 0 up to rounding)
        errs.append((nm, e))
    inf = torch.isinf(a[1])
    el = float((a[1][~inf] - b[1][~inf]).abs().max()) if bool((~inf).any()) else 0.0
    good = all(e < tol for _, e in errs) and el < (2e-3 if dt == torch.float16 else 2e-2) and bool(torch.equal(inf, torch.isinf(b[1])))
    return good, errs, el, b[5]


def main():
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    bad = 0
    for it in range(n):
        causal = rng.random() < 0.25
        Tq = rng.choice([32, 40, 64, 100, 128, 130, 192, 256, 300, 320, 512, 1024])
        Tk = Tq if causal else rng.choice([24, 56, 64, 66, 100, 128, 200, 256, 300, 520, 1000, 1024])
        if Tk % 2:
            Tk += 1
        qskip = (not causal or True) and Tq == Tk and rng.random() < 0.5
        padmode = rng.choice([0, 1, 2]) if not qskip else rng.choice([1, 2])
        B, H = rng.choice([(1, 1), (1, 3), (2, 4), (3, 2), (2, 8), (5, 8), (1, 12)])
        if Tq * Tk * B * H > 64 * 1024 * 1024:
            B, H = 1, 2
        p = rng.choice([0.0, 0.1, 0.1])
        bits = rng.random() < 0.7
        dt = torch.float16 if rng.random() < 0.8 else torch.bfloat16
        good, errs, el, algo = run(B, H, Tq, Tk, causal, padmode, p, bits, qskip, dt, it)
        bad += not good
        print(f"{'ok  ' if good else 'FAIL'} B{B} H{H} Tq{Tq} Tk{Tk} causal={int(causal)} pad={padmode} p={p} bits={int(bits)} qskip={int(qskip)} {str(dt)[6:]} [{algo}]: " +
              " ".join(f"{nm} {e:.1e}" for nm, e in errs) + f" lse {el:.1e}", flush=True)
    print("ALL OK" if bad == 0 else f"{bad} FAILURES", flush=True)


if __name__ == "__main__":
    main()
