"""The four-wave NT GEMM (variant 40, csrc/afm_gemm_w4_impl.h: 128 x 128 per wave, the overlap inside the wave) against the ping-pong
kernel (variant 30 / 32): bit-equality (same products, same accumulation order), an fp32 reference, run-to-run identity, the padded-row
hint, and per-launch times at the c2 and c4 steps' plain-product shapes, both kernels in ONE process (order swapped between rounds).
`--abl` needs an AFM_GEMM_ABLATIONS build (variants 401 .. 406)."""
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


def check(m, n, k, bias=True, seed=0, dt=torch.float16):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(m, k, device="cuda", generator=g).to(dt); w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(dt)
    b = torch.randn(n, device="cuda", generator=g) if bias else None
    c = torch.full((m, n), float("nan"), dtype=dt, device="cuda")
    ops.gemm(a, w, c, bias=b, variant=40)
    assert ops.last_algo() == "mfma_nt_w4", ops.last_algo()
    ref = a.float() @ w.float().t() + (b if bias else 0)
    c30 = torch.empty_like(c); ops.gemm(a, w, c30, bias=b, variant=30)
    err = ((c.float() - ref).abs().max() / ref.abs().max()).item()
    same = bool(torch.equal(c, c30))
    ok = torch.isfinite(c).all().item() and err < (2e-3 if dt == torch.float16 else 1.5e-2) and same
    print(f"check {dt} {m}x{n}x{k} bias={bias}: rel err vs fp32 {err:.2e}  bit-equal to the ping-pong kernel: {same}  {'ok' if ok else 'FAIL'}", flush=True)
    return ok


def main():
    ok = True
    only = [a.split("=")[1] for a in sys.argv if a.startswith("--variants=")]
    if only:
        return times({int(v): f"v{v}" for v in only[0].split(",")}, long_k_only=True)
    for (m, n, k) in [(256, 256, 256), (512, 256, 256), (256, 512, 384), (2048, 768, 512), (8192, 1536, 512), (131072, 512, 512),
                      (16384, 512, 2048), (256 * 37, 256 * 3, 128 * 5), (256 * 9, 256 * 12, 256)]:
        ok &= check(m, n, k)
    ok &= check(4096, 1024, 512, bias=False)
    ok &= check(8192, 768, 3072, dt=torch.bfloat16)
    for rep in range(3):   # races show up as run-to-run differences
        ok &= check(131072, 512, 2048, seed=rep)
    # padded-row hint: dead 256-row tiles are written as zeros, live ones equal the unhinted product
    m, n, k = 256 * 64, 512, 1536
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * 0.05).half()
    live = (torch.rand(m // 64, device="cuda") > 0.6)
    live[:8] = False; live[-4:] = False
    a[~live.repeat_interleave(64)] = 0
    c0 = torch.empty(m, n, dtype=torch.float16, device="cuda"); c1 = torch.full_like(c0, float("nan"))
    ops.gemm(a, w, c0, variant=40)
    ops.gemm(a, w, c1, variant=40, k_live=live.to(torch.uint8).contiguous())
    same = bool(torch.equal(c0, c1)); ok &= same
    print(f"k_live hint ({int((~live).sum())} of {m // 64} blocks dead): equal to the unhinted product: {same}", flush=True)
    print("ALL OK" if ok else "FAILURES", flush=True)
    variants = {40: "w4", 30: "pp", 32: "pp balanced", 24: "ws 256x128"}
    times(variants)


def times(variants, long_k_only=False):
    B, S = 128, 1024
    M = B * S
    shapes = []
    for tag, d, f in (("c2", 512, 2048), ("c4", 768, 3072)):
        shapes += [(f"{tag} qkv fwd", M, 3 * d, d), (f"{tag} out fwd", M, d, d), (f"{tag} ffn2 fwd", M, d, f), (f"{tag} qkv dgrad", M, d, 3 * d),
                   (f"{tag} mem kv", M, 2 * d, d), (f"{tag} ffn1 plain", M, f, d)]
    shapes += [("c4 ffn1 dgrad (2f)", M, 768, 6144), ("c2 dec ffn2", 16384, 512, 2048)]
    if long_k_only:
        shapes = [s for s in shapes if s[0] in ("c2 ffn2 fwd", "c2 qkv fwd", "c4 ffn2 fwd")]
    for rnd in range(2):
        order = list(variants.items())
        if rnd:
            order = order[::-1]
        for name, m, n, k in shapes:
            a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * 0.05).half()
            c = torch.empty(m, n, dtype=torch.float16, device="cuda"); bias = torch.randn(n, device="cuda")
            res = {}
            for var, lab in order:
                try:
                    ms = t(lambda: ops.gemm(a, w, c, bias=bias, variant=var))
                    res[var] = f"{lab} {ms * 1e3:6.1f}us ({2.0 * m * n * k / ms / 1e9:5.0f} TF)"
                except Exception as e:  # noqa: BLE001
                    res[var] = f"{lab} n/a"
            print(f"{name:19s} {m}x{n}x{k}: " + "  ".join(res[v] for v in variants), flush=True)


if __name__ == "__main__":
    main()
