#!/bin/bash
# Round 5: the short-query dK/dV kernel (checks + times at the cross-attention shapes), then the GPU suite without -x.
mkdir -p gpurun_out/r5
timeout 900 python tools/experiments/attn_cross.py > gpurun_out/r5/attn_cross.log 2>&1
cat gpurun_out/r5/attn_cross.log | cut -c1-900
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r5/gputest_check6.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r5/gputest_check6.log | tail -8 | cut -c1-300
