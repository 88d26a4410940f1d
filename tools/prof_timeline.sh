#!/bin/bash
# Kernel timeline of the benchmark step (run on the GPU box through gpurun): gaps between kernels, per kernel.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/tl -o tl -- python3 $R/bench.py --dtype fp16 --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline > $O/tl.log 2>&1
python3 $R/tools/rocpd_timeline.py $(find $O/tl -name "*.db" | head -1) ${1:-1400} ${2:-} > $O/timeline.txt
rm -rf $O/tl
cat $O/timeline.txt
