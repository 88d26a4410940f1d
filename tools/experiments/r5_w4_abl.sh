#!/bin/bash
# Round 5: the four-wave NT GEMM with parts left out (AFM_GEMM_ABLATIONS build), one variant per process (an ablation build that faults
# kills only its own process): 40 full, 404 no epilogue, 402 no LDS-DMA, 406 LDS reads + MFMAs only, 401 no MFMAs; 30 / 306 the ping-pong kernel.
OUT=gpurun_out/r5; mkdir -p $OUT
export AFM_LIB_OVERRIDE=$PWD/tools/experiments/_abl/libafm_abl.so
for v in 40 404 402 406 401 30 306 302; do
  timeout 120 python tools/experiments/w4_gemm.py --variants=$v 2>&1 | grep -v "amdgpu.ids" | sed "s/^/[$v] /"
done > $OUT/w4_gemm_abl.log 2>&1
cat $OUT/w4_gemm_abl.log
