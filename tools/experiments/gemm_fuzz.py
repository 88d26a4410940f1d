"""Random-shape sweep of the GEMM dispatch of round 5 against fp32 references on the same fp16 / bf16 inputs: NT products through the
default dispatch (four-wave kernel, ping-pong, loader-wave, XCD-aware 2-D tile walk for wide weights, padded-row hints) and grouped weight
gradients (`afm_gemm_group`: the multi-round chunk plan, bias gradients, token-block hints, odd token counts).  One-off check."""
import os, sys, random, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops


def nt_case(rng, it):
    dt = torch.float16 if rng.random() < 0.8 else torch.bfloat16
    M = rng.choice([256, 512, 768, 2048, 4096, 16384, 33024, 65536, 131072, 100, 1000, 70000])
    N = rng.choice([64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 2304, 3072, 6144, 200])
    K = rng.choice([64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 72])
    if M * N > 131072 * 2304 or M * K > 131072 * 3072:
        M = 16384
    g = torch.Generator(device="cuda").manual_seed(it)
    a = torch.randn(M, K, device="cuda", generator=g).to(dt); w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dt)
    bias = torch.randn(N, device="cuda", generator=g) if rng.random() < 0.6 else None
    kw = {}
    live = None
    if bias is None and M % 64 == 0 and rng.random() < 0.5:
        live = (torch.rand(M // 64, device="cuda", generator=g) > 0.4)
        a[~live.repeat_interleave(64)] = 0
        kw["k_live"] = live.to(torch.uint8).contiguous()
    c = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
    ops.gemm(a, w, c, bias=bias, **kw)
    algo = ops.last_algo()
    ref = a.float() @ w.float().t() + (bias if bias is not None else 0)
    err = float((c.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-3))
    ok = bool(torch.isfinite(c.float()).all()) and err < (2e-3 if dt == torch.float16 else 1.5e-2)
    print(f"{'ok  ' if ok else 'FAIL'} NT {str(dt)[6:]} {M}x{N}x{K} bias={int(bias is not None)} hint={int(live is not None)} [{algo}]: {err:.1e}", flush=True)
    return ok


def group_case(rng, it):
    dt = torch.float16 if rng.random() < 0.8 else torch.bfloat16
    R = rng.choice([4096, 16384, 16320, 65536, 131072, 33024, 1000])
    dims = [64, 128, 256, 512, 768, 1024, 1536, 2048, 3072, 200]
    np_ = rng.randint(1, 5)
    g = torch.Generator(device="cuda").manual_seed(1000 + it)
    live = None
    if R % 64 == 0 and rng.random() < 0.5:
        live = (torch.rand(R // 64, device="cuda", generator=g) > 0.3)
    descs, keep, refs = [], [], []
    for _ in range(np_):
        m, n = rng.choice(dims), rng.choice(dims)
        if m * n > 2048 * 2048:
            m = 512
        dy = (torch.randn(R, m, device="cuda", generator=g) * 0.05).to(dt); x = torch.randn(R, n, device="cuda", generator=g).to(dt)
        if live is not None:
            dy[~live.repeat_interleave(64)] = 0
        gw0 = torch.randn(m, n, device="cuda", generator=g); gb0 = torch.randn(m, device="cuda", generator=g)
        gw, gb = gw0.clone(), gb0.clone()
        kw = dict(trans_a=True, trans_b=False, accumulate=True, a_colsum=gb)
        if live is not None:
            kw["k_live"] = live.to(torch.uint8).contiguous()
        descs.append(ops.gemm_desc(dy, x, gw, **kw)); keep.append((dy, x, gw, gb, kw))
        refs.append((gw0.double() + dy.double().t() @ x.double(), gb0.double() + dy.double().sum(0)))
    ops.gemm_group(descs)
    algo = ops.last_algo()
    ok = True
    worst = 0.0
    for (dy, x, gw, gb, _), (rw, rb) in zip(keep, refs):
        tol = 2e-4 * math.sqrt(R) / 4 + 1e-4 * float(rw.abs().max())
        e = float((gw.double() - rw).abs().max()); eb = float((gb.double() - rb).abs().max())
        worst = max(worst, e / tol, eb / (2e-4 * math.sqrt(R) / 4 + 1e-4 * float(rb.abs().max())))
        ok &= bool(torch.isfinite(gw).all())
    ok &= worst < 1.0
    print(f"{'ok  ' if ok else 'FAIL'} group {str(dt)[6:]} R={R} problems={[(k[0].shape[1], k[1].shape[1]) for k in keep]} hint={int(live is not None)} [{algo}]: worst err / tol {worst:.2f}", flush=True)
    return ok


def main():
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    bad = 0
    for it in range(n):
        bad += not nt_case(rng, it)
        bad += not group_case(rng, it)
    print("ALL OK" if bad == 0 else f"{bad} FAILURES", flush=True)


if __name__ == "__main__":
    main()
