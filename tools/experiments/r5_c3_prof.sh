#!/bin/bash
# Round 5: kernel statistics and step counters of the padded workload c3 (same model as c2, 53 % of the encoder positions padded).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05
mkdir -p $O $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/step_c3 -o step -- python3 $R/bench.py --workload c3 --dtype fp16 --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline --no-input-compare --no-eval > $O/step_c3_fp16.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $O/step_c3 -name "*.db" | head -1) $O/step_c3_fp16_kernel_stats.csv --from k_patch_ k_gather_rows 2> $O/step_c3_fp16_total.txt
rm -rf $O/step_c3
bash $R/tools/prof_step_pmc.sh r05 c3 fp16 > $O/step_pmc_c3.log 2>&1
cp $R/gpurun_out/prof/r05_c3_fp16_step_pmc.json $O/
cat $O/step_c3_fp16_total.txt
