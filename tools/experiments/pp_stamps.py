"""Where a wave of the ping-pong GEMM spends its cycles (needs the AFM_GEMM_ABLATIONS build): clock64 stamps of tile 3 of the first
32 workgroups, wave 0 (group 0) and wave 4 (group 1).  Per phase: read-section issue (reads + DMA pieces), vmcnt wait, lgkmcnt wait,
barrier, MFMA section, second barrier; per tile: k-loop and epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import numpy as np


def main():
    dev = "cuda:0"
    n1 = 32 * 2 * 8 * 4 * 8
    st = torch.zeros(n1 + 32 * 2 * 16 * 4, dtype=torch.int64, device=dev)
    os.environ["AFM_STAMPS"] = str(st.data_ptr())
    from multimodalanalytical_amd import ops
    M, N, K = 131072, int(sys.argv[1]) if len(sys.argv) > 1 else 1536, int(sys.argv[2]) if len(sys.argv) > 2 else 512
    var = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
    c = torch.empty(M, N, dtype=torch.float16, device=dev); bias = torch.randn(N, device=dev)
    for _ in range(20):
        ops.gemm(a, w, c, bias=bias, variant=var)
    st.zero_(); torch.cuda.synchronize()
    ops.gemm(a, w, c, bias=bias, variant=var); torch.cuda.synchronize()
    x = st.cpu().numpy().astype(np.float64)
    ph = x[:n1].reshape(32, 2, 8, 4, 8)
    tl = x[n1:].reshape(32, 2, 16, 4)
    names = ["issue(reads+dma)", "vmcnt wait", "lgkm wait", "barrier", "mfma section", "barrier2"]
    for g in (0, 1):
        print(f"group {g}: cycles per phase step, mean over 32 workgroups x K-tiles 1..6 (tile 3)")
        for p in range(4):
            d = [(ph[:, g, 1:7, p, i + 1] - ph[:, g, 1:7, p, i]).mean() for i in range(6)]
            print(f"  phase {p}: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, d)) + f"   total {sum(d):6.0f}")
        kt = (ph[:, g, 6, 0, 0] - ph[:, g, 1, 0, 0]).mean() / 5
        print(f"  K-tile period {kt:7.0f} cycles (ideal 4 x 2 x 256 = 2048)")
        ok = tl[:, g, 1:10, 0] > 0
        kl = (tl[:, g, 1:10, 1] - tl[:, g, 1:10, 0])[ok]; ep = (tl[:, g, 1:10, 2] - tl[:, g, 1:10, 1])[ok]
        gap = (tl[:, g, 2:10, 0] - tl[:, g, 1:9, 2])[ok[:, 1:]]
        print(f"  per tile: k-loop {kl.mean():8.0f}  epilogue {ep.mean():7.0f}  (between tiles {gap.mean():5.0f}) cycles")
        # in-kernel clock: shader-clock ticks per 100-MHz tick between the starts of tiles 1 and 9
        okc = (tl[:, g, 9, 0] > 0) & (tl[:, g, 1, 0] > 0)
        clk = ((tl[:, g, 9, 0] - tl[:, g, 1, 0]) / np.maximum(1.0, tl[:, g, 9, 3] - tl[:, g, 1, 3]))[okc] * 0.1
        per = ((tl[:, g, 9, 3] - tl[:, g, 1, 3]) / 8.0)[okc] * 10.0
        print(f"  in-kernel clock {np.median(clk):.3f} GHz; tile period {np.median(per):.0f} ns = {np.median(per) * np.median(clk):.0f} cycles")


if __name__ == "__main__":
    main()
