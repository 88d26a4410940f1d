#!/bin/bash
# Round 5: the forward on 16x16x32 (checks + timings) and the dK/dV bit-identity / tolerance test in full.
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_fp16.py -m gpu -x -q -k "dkv_pipelined" 2>&1 | tail -40 > gpurun_out/r5/dkv_test.log
cat gpurun_out/r5/dkv_test.log | cut -c1-300
timeout 900 python tools/experiments/attn_m16.py > gpurun_out/r5/attn_fwd16.log 2>&1
grep -n "FAIL\|ALL OK\|FAILURES\|Error\|error" gpurun_out/r5/attn_fwd16.log | head; grep "fwd 32x32x16" gpurun_out/r5/attn_fwd16.log
