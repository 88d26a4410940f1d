#!/bin/bash
# Round-4 profiles (run on the GPU box through gpurun; tools/merge_r04_profiles.py turns gpurun_out/prof_r04/ into profiles/r04_*):
#   step_fp16_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the benchmark command (3 optimiser steps, timed mode)
#   attn_fp16_pmc_{1..5}.json    five PMC passes of the attention kernels at the c2 encoder shape + their --stats run
#   gemm / step PMC              tools/prof_gemm_pmc.sh (the dominant NT / TN GEMM launches), tools/prof_step_pmc.sh (whole step)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/step_fp16 -o step -- python3 $R/bench.py --dtype fp16 --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline --no-input-compare > $O/step_fp16.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $O/step_fp16 -name "*.db" | head -1) $O/step_fp16_kernel_stats.csv 2> $O/step_fp16_total.txt
rm -rf $O/step_fp16
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
i=0
for g in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d $O/attn_fp16_$i -o pmc -- python3 $R/tools/bench_attn_x3.py --mode fp16 --old 1 > $O/attn_fp16_$i.log 2>&1
  python3 $R/tools/rocpd_pmc.py $(find $O/attn_fp16_$i -name "*.db" | head -1) k_attn > $O/attn_fp16_pmc_$i.json
  rm -rf $O/attn_fp16_$i
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/attn_stats -o st -- python3 $R/tools/bench_attn_x3.py --mode fp16 --old 1 > $O/attn_fp16_stats.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $O/attn_stats -name "*.db" | head -1) $O/attn_fp16_kernel_stats.csv 2> $O/attn_fp16_total.txt
rm -rf $O/attn_stats
bash $R/tools/prof_gemm_pmc.sh r04 > $O/gemm_pmc.log 2>&1
cp $R/gpurun_out/prof/r04_gemm_fp16_pmc.json $O/
bash $R/tools/prof_step_pmc.sh r04 c2 fp16 > $O/step_pmc.log 2>&1
cp $R/gpurun_out/prof/r04_c2_fp16_step_pmc.json $O/
ls $O
