#!/bin/bash
# Step-level counters of the benchmark command: three rocprofv3 PMC passes (--kernel-trace --pmc only) over
#   python3 bench.py --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline --no-input-compare   (3 optimiser steps)
#   bash tools/prof_step_pmc.sh <tag> [workload] [dtype]  -> gpurun_out/prof/<tag>_<workload>_<dtype>_step_pmc.json
set -u
R=$GRAFT_REPO_ROOT
tag=${1:-r04}; wl=${2:-c2}; dt=${3:-fp16}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY"
G2="FETCH_SIZE"
G3="WRITE_SIZE"
i=0
for g in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $g -d $O/spmc_$i -o pmc -- python3 $R/bench.py --workload $wl --dtype $dt --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline --no-input-compare --no-eval > $O/${tag}_step_$i.log 2>&1
  cp $(find $O/spmc_$i -name "*.db" | head -1) $O/${tag}_step_pass$i.db 2>/dev/null
  rm -rf $O/spmc_$i
done
python3 $R/tools/prof_step_reduce.py $O $tag 3 $wl > $O/${tag}_${wl}_${dt}_step_pmc.json
rm -f $O/${tag}_step_pass*.db
head -60 $O/${tag}_${wl}_${dt}_step_pmc.json
