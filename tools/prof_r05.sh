#!/bin/bash
# Round-5 profiles (run on the GPU box through gpurun; tools/merge_r05_profiles.py turns gpurun_out/prof_r05/ into profiles/r05_*):
#   step_{c2,c4}_fp16_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the benchmark command (3 optimiser steps, timed mode)
#   r05_{c2,c4}_fp16_step_pmc.json       tools/prof_step_pmc.sh: whole-step counters (three passes)
#   r05_gemm_fp16_pmc.json, r05_c4_gemm_fp16_pmc.json   tools/prof_gemm_pmc.sh: the dominant GEMM launches at the c2 / c4 shapes (five passes)
#   attn_fp16_pmc_{1..5}.json + stats    the attention kernels at the c2 encoder shape
#   which parts run is chosen by the arguments (default: all):  bash tools/prof_r05.sh [stats] [steppmc] [gemm] [attn]
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05
mkdir -p $O $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
what="${*:-stats steppmc gemm attn}"
if [[ " $what " == *" stats "* ]]; then
  for wl in c2 c4; do
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/step_$wl -o step -- python3 $R/bench.py --workload $wl --dtype fp16 --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline --no-input-compare --no-eval > $O/step_${wl}_fp16.log 2>&1
    python3 $R/tools/rocpd_stats.py $(find $O/step_$wl -name "*.db" | head -1) $O/step_${wl}_fp16_kernel_stats.csv --from k_patch_ k_gather_rows 2> $O/step_${wl}_fp16_total.txt
    rm -rf $O/step_$wl
  done
fi
if [[ " $what " == *" steppmc "* ]]; then
  for wl in c2 c4; do
    bash $R/tools/prof_step_pmc.sh r05 $wl fp16 > $O/step_pmc_$wl.log 2>&1
    cp $R/gpurun_out/prof/r05_${wl}_fp16_step_pmc.json $O/
  done
fi
if [[ " $what " == *" gemm "* ]]; then
  bash $R/tools/prof_gemm_pmc.sh r05 c2 > $O/gemm_pmc_c2.log 2>&1
  cp $R/gpurun_out/prof/r05_gemm_fp16_pmc.json $O/
  bash $R/tools/prof_gemm_pmc.sh r05 c4 > $O/gemm_pmc_c4.log 2>&1
  cp $R/gpurun_out/prof/r05_c4_gemm_fp16_pmc.json $O/
fi
if [[ " $what " == *" attn "* ]]; then
  G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
  G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
  i=0
  for g in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $g -d $O/attn_fp16_$i -o pmc -- python3 $R/tools/bench_attn_x3.py --mode fp16 > $O/attn_fp16_$i.log 2>&1
    python3 $R/tools/rocpd_pmc.py $(find $O/attn_fp16_$i -name "*.db" | head -1) k_attn > $O/attn_fp16_pmc_$i.json
    rm -rf $O/attn_fp16_$i
  done
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/attn_stats -o st -- python3 $R/tools/bench_attn_x3.py --mode fp16 > $O/attn_fp16_stats.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find $O/attn_stats -name "*.db" | head -1) $O/attn_fp16_kernel_stats.csv 2> $O/attn_fp16_total.txt
  rm -rf $O/attn_stats
  # the three attention kernels in both MFMA shapes: cycles, clock, MFMA busy per kernel name (template arguments = dropout path, build)
  i=0
  for g in "$G1" "$G2"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $g -d $O/dqshape_$i -o pmc -- python3 $R/tools/experiments/attn_m16.py --time-only > $O/dqshape_$i.log 2>&1
    python3 $R/tools/rocpd_pmc.py $(find $O/dqshape_$i -name "*.db" | head -1) k_attn > $O/dqshape_pmc_$i.json
    rm -rf $O/dqshape_$i
  done
fi
ls $O
