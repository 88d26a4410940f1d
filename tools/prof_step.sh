#!/bin/bash
# Per-kernel statistics of the benchmark step in one precision mode (run on the GPU box through gpurun):
#   bash tools/prof_step.sh <mode> <tag>     ->  gpurun_out/prof/<tag>_<mode>_kernel_stats.csv (+ _total.txt)
set -u
R=$GRAFT_REPO_ROOT
m=${1:-fp16}
tag=${2:-r03}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/step_$m -o step -- python3 $R/bench.py --dtype $m --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline > $O/${tag}_step_$m.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $O/step_$m -name "*.db" | head -1) $O/${tag}_c2_${m}_kernel_stats.csv 2> $O/${tag}_c2_${m}_total.txt
rm -rf $O/step_$m
head -40 $O/${tag}_c2_${m}_kernel_stats.csv | cut -c1-150; cat $O/${tag}_c2_${m}_total.txt
