#!/bin/bash
# Per-kernel statistics of the c3 step (padded multimodal batch) with and without the backward's padded-row hints.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in 1 0; do
  export AFM_ROW_SKIP=$e
  rocprofv3 --kernel-trace --stats -d $O/c3_$e -o step -- python3 $R/bench.py --workload c3 --dtype fp16 --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline > $O/c3_$e.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find $O/c3_$e -name "*.db" | head -1) $O/r03_c3_fp16_rowskip${e}_kernel_stats.csv 2> $O/r03_c3_rowskip${e}_total.txt
  rm -rf $O/c3_$e
  head -12 $O/r03_c3_fp16_rowskip${e}_kernel_stats.csv | cut -c1-140; cat $O/r03_c3_rowskip${e}_total.txt
done
