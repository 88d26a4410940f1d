"""In-kernel phase stamps of the persistent NT GEMM (needs AFM_EXTRA_FLAGS=-DAFM_GEMM_ABLATIONS build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

VAR = int(os.environ.get('VAR', '12'))

def main():
    dev = "cuda:0"
    st = torch.zeros(512 * 64 + 64 * 12 * 8 * 5, dtype=torch.int64, device=dev)
    os.environ["AFM_STAMPS"] = str(st.data_ptr())
    from multimodalanalytical_amd import ops
    M, N, K = 131072, int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 512
    mode = sys.argv[3] if len(sys.argv) > 3 else "plain"
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    bias = torch.randn(N, device=dev); pre = torch.empty_like(c); dr = ops.drop(0.1, 1, 1)
    for _ in range(3):
        st.zero_()
        if mode == "gelu": ops.gemm(a, w, c, bias=bias, act=2, pre_act=pre, dropout=dr, variant=VAR)
        else: ops.gemm(a, w, c, variant=VAR)
        torch.cuda.synchronize()
    ks = st.cpu()[512 * 64:].view(64, 12, 8, 5).numpy().astype('float64')
    s = st.cpu()[:512 * 64].view(512, 16, 4).numpy().astype("float64") / 100.0   # us (100 MHz)
    t0 = s[:256, 0, 0].min()
    for b in (0, 1, 8, 100, 255):
        row = s[b]
        print(f"block {b}: " + " | ".join(f"{row[i,0]-t0:6.1f} +{row[i,1]-row[i,0]:4.1f} +{row[i,2]-row[i,1]:4.1f}" for i in range(16) if row[i, 0] > 0))
    v = s[:256]
    ok = v[:, :, 0] > 0
    main_t = (v[:, :, 1] - v[:, :, 0])[ok]; epi_t = (v[:, :, 2] - v[:, :, 1])[ok]
    print(f"main loop mean {main_t.mean():.2f} us (min {main_t.min():.2f} max {main_t.max():.2f}); epilogue mean {epi_t.mean():.2f} us (min {epi_t.min():.2f} max {epi_t.max():.2f}); total span {v[:, :, 2].max() - t0:.1f} us")

    # k-step phases of tile 5 (clock64 units), per wave: vmcnt wait, barrier wait, DMA issue, reads+MFMA
    for wv in range(12):
        if ks[:, wv].max() == 0: continue
        x = ks[:, wv]
        ph = [(x[:, :, i + 1] - x[:, :, i]).mean() for i in range(4)]
        gap = (x[:, 1:, 0] - x[:, :-1, 4]).mean()
        print(f"wave {wv}: vmcnt {ph[0]:6.0f}  barrier {ph[1]:6.0f}  dma-issue {ph[2]:6.0f}  compute {ph[3]:6.0f}  loop-gap {gap:5.0f}   k-step {(x[:, 7, 4] - x[:, 0, 0]).mean() / 8:6.0f}")

if __name__ == "__main__":
    main()
