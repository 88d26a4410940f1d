"""LayerNorm forward (+ residual add + branch dropout) and backward at the c2 encoder shape: time and HBM rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalanalytical_amd import ops


def t(fn, it=30, warm=10):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


dev = "cuda:0"
rows, d = 131072, 512
for a_ in sys.argv[1:]:      # --d=768 --rows=16384
    if a_.startswith("--d="): d = int(a_[4:])
    if a_.startswith("--rows="): rows = int(a_[7:])
print(f"rows {rows} d {d}  AFM_LN_FWD_BLOCKS={os.environ.get('AFM_LN_FWD_BLOCKS', '(by d)')} AFM_LN_BWD_BLOCKS={os.environ.get('AFM_LN_BWD_BLOCKS', '(by d)')}")
cd = torch.float16
x = torch.randn(rows, d, device=dev); br = torch.randn(rows, d, device=dev).to(cd)
gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
y = torch.empty(rows, d, dtype=cd, device=dev); xs = torch.empty_like(x)
mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
dr = ops.drop(0.1, 1, 2)
ms = t(lambda: ops.layernorm_fwd(x, gam, bet, y, mean, rstd, add=br, x_sum=xs, add_dropout=dr))
byt = rows * d * (4 + 2 + 4 + 2)
print(f"ln fwd + add + dropout  {ms * 1e3:7.1f} us  {byt / ms / 1e9:6.2f} TB/s")
ms = t(lambda: ops.layernorm_fwd(x, gam, bet, y, mean, rstd))
print(f"ln fwd plain            {ms * 1e3:7.1f} us  {rows * d * 6 / ms / 1e9:6.2f} TB/s")
dy = torch.randn(rows, d, device=dev).to(cd); dres = torch.randn(rows, d, device=dev)
dx = torch.empty_like(x); dxd = torch.empty(rows, d, dtype=cd, device=dev)
dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
ws = torch.empty(ops.layernorm_bwd_ws(rows, d), device=dev)
ms = t(lambda: ops.layernorm_bwd(dy, x, gam, mean, rstd, dx, dg, db, ws, dres=dres, dx_drop=dxd, dropout=dr))
print(f"ln bwd + dres + dropped copy {ms * 1e3:7.1f} us  {rows * d * (2 + 4 + 4 + 4 + 2) / ms / 1e9:6.2f} TB/s")
