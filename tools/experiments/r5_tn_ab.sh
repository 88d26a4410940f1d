#!/bin/bash
# Round 5: grouped weight gradients -- the four-wave unit vs the eight-wave one under both chunk rules (one wave of units, rounds 3-4;
# several rounds of shorter units, round 5) and with activation-gradient magnitudes (DY_SCALE=0.01, as bench.py's roofline entries)
mkdir -p gpurun_out/r5
{
echo "== multi-round rule, dy scale 1"; timeout 300 python tools/experiments/tnw4_gemm.py --time-only
echo "== one-wave rule, dy scale 1"; AFM_TN_ONE_WAVE=1 timeout 300 python tools/experiments/tnw4_gemm.py --time-only
echo "== multi-round rule, dy scale 0.01"; DY_SCALE=0.01 timeout 300 python tools/experiments/tnw4_gemm.py --time-only
echo "== one-wave rule, dy scale 0.01"; DY_SCALE=0.01 AFM_TN_ONE_WAVE=1 timeout 300 python tools/experiments/tnw4_gemm.py --time-only
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/tn_ab.log
