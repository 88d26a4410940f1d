#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun): per-kernel stats of the benchmark step in both precision modes and
# PMC passes (one rocprofv3 run per counter group, --kernel-trace + --pmc only) of the attention and GEMM kernels at the C2
# encoder shapes.  Outputs land in gpurun_out/prof_r02/ ; tools/rocpd_stats.py / rocpd_pmc.py turn them into profiles/r02_*.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in bf16x3-mixed bf16x3 bf16; do
  rocprofv3 --kernel-trace --stats -d $O/step_$m -o step -- python3 $R/bench.py --dtype $m --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline > $O/step_$m.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find $O/step_$m -name "*.db" | head -1) $O/step_${m}_kernel_stats.csv 2> $O/step_${m}_total.txt
  rm -rf $O/step_$m
done
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
for m in bf16x3 bf16; do
  i=0
  for g in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $g -d $O/attn_${m}_$i -o pmc -- python3 $R/tools/bench_attn_x3.py --mode $m > $O/attn_${m}_$i.log 2>&1
    python3 $R/tools/rocpd_pmc.py $(find $O/attn_${m}_$i -name "*.db" | head -1) k_attn > $O/attn_${m}_pmc_$i.json
    rm -rf $O/attn_${m}_$i
  done
done
# the same attention kernels re-hashing the dropout (no keep-bit tensor): instruction counters only, to show what the bits save
for m in bf16x3 bf16; do
  rocprofv3 --kernel-trace --pmc $G1 -d $O/attn_${m}_hash_1 -o pmc -- python3 $R/tools/bench_attn_x3.py --mode $m --bits 0 > $O/attn_${m}_hash_1.log 2>&1
  python3 $R/tools/rocpd_pmc.py $(find $O/attn_${m}_hash_1 -name "*.db" | head -1) k_attn > $O/attn_${m}_hash_pmc_1.json
  rm -rf $O/attn_${m}_hash_1
done
i=0
for g in "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $g -d $O/gemm_bf16x3_$i -o pmc -- python3 $R/tools/bench_gemm_x3.py --shapes ffn1 --variants 0 > $O/gemm_bf16x3_$i.log 2>&1
  python3 $R/tools/rocpd_pmc.py $(find $O/gemm_bf16x3_$i -name "*.db" | head -1) k_x3 > $O/gemm_bf16x3_pmc_$i.json
  rm -rf $O/gemm_bf16x3_$i
done
ls $O
