"""The ping-pong NT GEMM (variant 30, csrc/afm_gemm_pp_impl.h) against the shipped tile forms: bit-for-bit agreement with the
loader-wave kernel (same products, same fp32 accumulation order along k? no: checked against an fp32 reference within fp16 rounding)
and per-launch times at the c2 step's shapes.  `--abl` needs an AFM_GEMM_ABLATIONS build (variants 301 .. 306)."""
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


def check(m, n, k, bias=True, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(m, k, device="cuda", generator=g).half(); w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
    b = torch.randn(n, device="cuda", generator=g) if bias else None
    c = torch.full((m, n), float("nan"), dtype=torch.float16, device="cuda")
    ops.gemm(a, w, c, bias=b, variant=30)
    assert ops.last_algo() == "mfma_nt_pp", ops.last_algo()
    ref = a.float() @ w.float().t() + (b if bias else 0)
    c24 = torch.empty_like(c); ops.gemm(a, w, c24, bias=b, variant=24)
    err = ((c.float() - ref).abs().max() / ref.abs().max()).item()
    same = (c == c24).float().mean().item()
    ok = torch.isfinite(c).all().item() and err < 2e-3
    print(f"check {m}x{n}x{k} bias={bias}: rel err vs fp32 {err:.2e}  equal to the 256x128 kernel's output: {same * 100:.3f} %  {'ok' if ok else 'FAIL'}", flush=True)
    return ok


def main():
    ok = True
    for (m, n, k) in [(256, 256, 128), (512, 256, 128), (256, 512, 192), (2048, 768, 512), (8192, 1536, 512), (131072, 512, 512),
                      (16384, 512, 2048), (256 * 37, 256 * 3, 64 * 5)]:
        ok &= check(m, n, k)
    ok &= check(4096, 1024, 512, bias=False)
    for (m, n, k) in [(256, 256, 128), (512, 768, 256), (2048, 768, 512), (16384, 512, 2048), (256 * 37, 256 * 3, 384), (131072, 1536, 512)]:
        a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * 0.05).half(); b = torch.randn(n, device="cuda")
        c0 = torch.empty(m, n, dtype=torch.float16, device="cuda"); c1 = torch.full_like(c0, float("nan"))
        ops.gemm(a, w, c0, bias=b, variant=30)
        for v in (32, 33, 34):
            c1.fill_(float("nan")); ops.gemm(a, w, c1, bias=b, variant=v)
            same = bool(torch.equal(c0, c1)); ok &= same
            print(f"variant {v} == variant 30 at {m}x{n}x{k}: {same}", flush=True)
    for rep in range(3):   # races show up as run-to-run differences
        ok &= check(131072, 1536, 512, seed=rep)
    print("ALL OK" if ok else "FAILURES", flush=True)
    B, S, d, f = 128, 1024, 512, 2048
    M = B * S
    shapes = [("qkv fwd", M, 3 * d, d), ("out fwd", M, d, d), ("ffn1 plain", M, f, d), ("ffn2 fwd", M, d, f), ("qkv dgrad", M, d, 3 * d),
              ("mem kv", M, 2 * d, d), ("dec qkv", 16384, 3 * d, d), ("dec ffn2", 16384, d, f)]
    variants = {30: "pp", 33: "pp one-barrier", 34: "pp 1bar+bal", 32: "pp balanced", 24: "ws 256x128", 28: "256x256"}
    if "--abl" in sys.argv:
        variants.update({304: "pp no-epilogue", 302: "pp no-dma", 306: "pp lds+mfma", 301: "pp no-mfma", 250: "ws no-epilogue", 246: "ws lds+mfma"})
    for rnd in range(2):
        for name, m, n, k in shapes:
            a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * 0.05).half()
            c = torch.empty(m, n, dtype=torch.float16, device="cuda"); bias = torch.randn(n, device="cuda")
            res = []
            for var, lab in variants.items():
                try:
                    ms = t(lambda: ops.gemm(a, w, c, bias=bias, variant=var))
                    res.append(f"{lab} {ms * 1e3:6.1f}us ({2.0 * m * n * k / ms / 1e9:5.0f} TF)")
                except Exception as e:  # noqa: BLE001
                    res.append(f"{lab} n/a")
            print(f"{name:11s} {m}x{n}x{k}: " + "  ".join(res), flush=True)


if __name__ == "__main__":
    main()
