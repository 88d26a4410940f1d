#!/bin/bash
# rocprofv3 --kernel-trace --stats of the c4 step (12L d768 gated, all modalities, padded batch): bench.py --workload c4, 2 + 1 optimiser steps
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/step_c4 -o step -- python3 $R/bench.py --workload c4 --dtype fp16 --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline --no-input-compare --no-parity > $O/step_c4.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $O/step_c4 -name "*.db" | head -1) $O/step_c4_fp16_kernel_stats.csv 2> $O/step_c4_total.txt
rm -rf $O/step_c4
head -16 $O/step_c4_fp16_kernel_stats.csv | cut -c1-170; cat $O/step_c4_total.txt
