import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodalanalytical_amd import ops
dev='cuda:0'
def t(fn, it=20, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
R=126976
dy=torch.randn(R,512,device=dev); x=torch.randn(R,2,device=dev); g=torch.zeros(512,2,device=dev); gb=torch.zeros(512,device=dev)
print("skinny_n us", t(lambda: ops.gemm(dy,x,g,trans_a=True,trans_b=False,accumulate=True,a_colsum=gb))*1e3, ops.last_algo())
w=torch.randn(512,2,device=dev); b=torch.randn(512,device=dev); y=torch.empty(R,512,device=dev)
print("skinny_k us", t(lambda: ops.gemm(x,w,y,trans_b=True,bias=b))*1e3, ops.last_algo())
