"""afm_patch_preprocess throughput at the benchmark workloads' shapes (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalanalytical_amd import ops

def t(fn, it=50, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it

def main():
    dev = "cuda:0"
    for name, B, L, ps, interp in [("c2 IR ps=2", 128, 1984, 2, False), ("c3/c4 IR ps=75 interp", 128, 1800, 75, True),
                                   ("c2 x64 (one 8192-sample shard)", 8192, 1984, 2, False)]:
        sp = torch.rand(B, L, device=dev)
        pr = torch.ones(B, dtype=torch.bool, device=dev)
        ms = t(lambda: ops.patch_preprocess(sp, pr, 0.5, 0.3, ps, interpolation=interp, seq_first=True))
        P = (1625 if interp else L) // ps
        by = 4 * B * L + 4 * B * P * ps + B * P
        print(f"{name:32s} B={B:5d}: {ms*1e3:8.1f} us  {by/ms/1e6:8.1f} GB/s algorithmic  {B/ms*1e3/1e6:8.2f} M spectra/s")

if __name__ == "__main__":
    main()
