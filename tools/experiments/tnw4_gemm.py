"""The four-wave weight-gradient unit (csrc/afm_gemm_tnw4_impl.h) against the eight-wave one (variant 107 keeps it): correctness
against fp32 (single launches with and without split-K, bias gradient, the padded-row hint, the gated de-interleave, accumulate), then
an encoder / decoder layer's grouped weight gradients at the c2 and c4 shapes, both forms alternating in ONE process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops


def t(fn, it=20, warm=15):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


dev = "cuda:0"


def check(R, M, N, dt=torch.float16, live_frac=None, glu=0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    dy = (torch.randn(R, M, device=dev, generator=g) * 0.5).to(dt); x = torch.randn(R, N, device=dev, generator=g).to(dt)
    kl = None
    if live_frac is not None:
        live = torch.rand(R // 64, device=dev, generator=g) < live_frac
        live[:3] = False; live[-2:] = False
        dy[~live.repeat_interleave(64)] = 0
        kl = live.to(torch.uint8).contiguous()
    ref = dy.float().t() @ x.float()
    refb = dy.float().sum(0)
    if glu:
        idx = torch.tensor([((n >> 3) << 2) + (n & 3) + ((n >> 2) & 1) * glu for n in range(M)], device=dev)
        full = torch.zeros_like(ref); full[idx] = ref; ref = full
        fb = torch.zeros_like(refb); fb[idx] = refb; refb = fb
    out = {}
    for var in (108, 107):
        gw = torch.full((M, N), 1.0, device=dev); gb = torch.full((M,), 2.0, device=dev)
        ops.gemm(dy, x, gw, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb, variant=var, k_live=kl, glu_rows=glu)
        out[var] = (gw - 1.0, gb - 2.0, ops.last_algo())
    e0 = float((out[108][0] - ref).abs().max() / ref.abs().max()); e1 = float((out[107][0] - ref).abs().max() / ref.abs().max())
    b0 = float((out[108][1] - refb).abs().max() / refb.abs().max())
    ok = "w4" in out[108][2] and "w4" not in out[107][2] and e0 < 3e-4 * (8 if dt == torch.bfloat16 else 1) and b0 < 1e-4
    print(f"check {R}x{M}x{N} {dt} live={live_frac} glu={glu}: [{out[108][2]}] dW err {e0:.2e} (eight-wave [{out[107][2]}] {e1:.2e}), db err {b0:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
    return ok


def main():
    ok = True
    if "--time-only" in sys.argv:
        return times()
    for (R, M, N) in [(65536, 1536, 512), (131072, 512, 2048), (131072, 2048, 512)]:
        ok &= check(R, M, N)
    ok &= check(65536, 768, 3072, dt=torch.bfloat16)
    ok &= check(131072, 1536, 512, live_frac=0.55)
    ok &= check(65536, 2048, 512, glu=1024)
    # grouped launch: every problem against fp32, both forms
    probs = [(131072, 512, 512), (131072, 1536, 512), (131072, 512, 2048), (131072, 2048, 512)]
    ten = []
    for R, M, N in probs:
        dy = (torch.randn(R, M, device=dev) * 0.5).half(); x = torch.randn(R, N, device=dev).half()
        ten.append((dy, x))
    for var in (108, 107):
        gs = [(torch.zeros(M, N, device=dev), torch.zeros(M, device=dev)) for _, M, N in probs]
        descs = [ops.gemm_desc(dy, x, g, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb, variant=var) for (dy, x), (g, gb) in zip(ten, gs)]
        ops.gemm_group(descs)
        algo = ops.last_algo()
        errs = [float((g - dy.float().t() @ x.float()).abs().max() / (dy.float().t() @ x.float()).abs().max()) for (dy, x), (g, _) in zip(ten, gs)]
        berr = [float((gb - dy.float().sum(0)).abs().max() / dy.float().sum(0).abs().max()) for (dy, x), (_, gb) in zip(ten, gs)]
        good = max(errs) < 3e-4 and max(berr) < 1e-4 and (("w4" in algo) == (var == 108))
        ok &= good
        print(f"grouped launch variant {var} [{algo}]: dW errs {[f'{e:.1e}' for e in errs]} db errs {[f'{e:.1e}' for e in berr]}  {'ok' if good else 'FAIL'}", flush=True)
    print("ALL OK" if ok else "FAILURES", flush=True)
    times()


def times():
    sets = {"c2 enc layer": [(131072, 512, 512), (131072, 1536, 512), (131072, 512, 2048), (131072, 2048, 512)],
            "c2 dec layer + memory k|v": [(16384, 512, 512)] * 3 + [(16384, 1536, 512), (16384, 512, 2048), (16384, 2048, 512), (131072, 1024, 512)],
            "c4 enc layer (gated)": [(131072, 768, 768), (131072, 2304, 768), (131072, 768, 3072), (131072, 6144, 768)],
            "c4 dec layer + memory k|v": [(16384, 768, 768)] * 3 + [(16384, 2304, 768), (16384, 768, 3072), (16384, 6144, 768), (131072, 1536, 768)]}
    dscale = float(os.environ.get("DY_SCALE", "1.0"))       # (bench.py's roofline entries use 0.01: activation-gradient magnitudes)
    for rnd in range(2):
        for name, probs in sets.items():
            ten, flop = [], 0.0
            for R, M, N in probs:
                dy = (torch.randn(R, M, device=dev) * dscale).half(); x = torch.randn(R, N, device=dev).half()
                ten.append((dy, x, torch.zeros(M, N, device=dev), torch.zeros(M, device=dev))); flop += 2.0 * R * M * N
            res = {}
            for var in ((108, 107) if rnd == 0 else (107, 108)):
                descs = [ops.gemm_desc(dy, x, g, trans_a=True, trans_b=False, accumulate=True, a_colsum=gb, variant=var) for dy, x, g, gb in ten]
                ms = t(lambda: ops.gemm_group(descs))
                res[var] = f"{ms:7.3f} ms ({flop / ms / 1e9:5.0f} TF/s) [{ops.last_algo()}]"
            print(f"{name:28s} four-wave {res[108]}   eight-wave {res[107]}", flush=True)


if __name__ == "__main__":
    main()
