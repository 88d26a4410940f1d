#!/bin/bash
OUT=gpurun_out/r5; mkdir -p $OUT
timeout 600 python tools/experiments/attn_m16.py > $OUT/attn_m16.log 2>&1; echo "rc=$?" >> $OUT/attn_m16.log
grep -v "amdgpu.ids" $OUT/attn_m16.log
bash tools/experiments/r5_w4_abl.sh
