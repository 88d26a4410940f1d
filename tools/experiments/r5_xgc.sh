#!/bin/bash
mkdir -p gpurun_out/r5
{
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp16.py -m gpu -x -q -k "gemm or glu or gated or nt_" 2>&1 | tail -3
for r in 1 2; do
  timeout 300 python tools/experiments/xgc_time.py
  AFM_NT_XGC=1 timeout 300 python tools/experiments/xgc_time.py
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/xgc.log
