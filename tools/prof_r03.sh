#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun): per-kernel statistics of the benchmark step in the timed mode (fp16) and in
# bf16x3-mixed, and PMC passes (one rocprofv3 run per counter group, --kernel-trace + --pmc only) of the attention kernels at the
# C2 encoder shape in fp16.  Outputs: gpurun_out/prof_r03/ ; tools/merge_r03_profiles.py turns them into profiles/r03_*.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in fp16 bf16x3-mixed; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/step_$m -o step -- python3 $R/bench.py --dtype $m --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline > $O/step_$m.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find $O/step_$m -name "*.db" | head -1) $O/step_${m}_kernel_stats.csv 2> $O/step_${m}_total.txt
  rm -rf $O/step_$m
done
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
ls $O
