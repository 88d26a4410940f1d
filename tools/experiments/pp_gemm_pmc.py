"""Ablations of the ping-pong NT GEMM at two c2 shapes, 300 launches each (the clock needs tens of milliseconds to settle), for tools/experiments/pp_gemm_pmc.sh (cycles vs time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodalanalytical_amd import ops

M = 131072
for name, n, k in (("ffn2 fwd", 512, 2048), ("qkv fwd", 1536, 512)):
    a = torch.randn(M, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * 0.05).half()
    c = torch.empty(M, n, dtype=torch.float16, device="cuda"); bias = torch.randn(n, device="cuda")
    for var in (30, 302, 304, 306):      # (301, no LDS reads + MFMAs, faults: its dead asm reads end up as address registers)
        for _ in range(300):
            ops.gemm(a, w, c, bias=bias, variant=var)
        torch.cuda.synchronize()
