#!/bin/bash
# Round 5: per-kernel time per optimiser step at B = 256 x 2 against B = 128 x 4 (same 512 samples per step): which kernels carry the
# per-launch cost that the micro-batch sweep shows.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ba in "128 4" "256 2"; do
  set -- $ba
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/b$1 -o step -- python3 $R/bench.py --workload c2 --dtype fp16 --batch $1 --acc $2 --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity > $O/b$1.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find $O/b$1 -name "*.db" | head -1) $O/b$1_kernel_stats.csv --from k_patch_ k_gather_rows 2> $O/b$1_total.txt
  rm -rf $O/b$1
  cat $O/b$1_total.txt
done
