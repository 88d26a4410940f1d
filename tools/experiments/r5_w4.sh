#!/bin/bash
# Round 5: the four-wave NT GEMM (variant 40) -- checks, then times against the ping-pong / loader-wave kernels.
OUT=gpurun_out/r5; mkdir -p $OUT
timeout 600 python tools/experiments/w4_gemm.py "$@" > $OUT/w4_gemm.log 2>&1; echo "rc=$?" >> $OUT/w4_gemm.log
grep -v "amdgpu.ids" $OUT/w4_gemm.log
