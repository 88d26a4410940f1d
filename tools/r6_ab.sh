#!/bin/bash
# A/B of one environment variable over a workload's training step on ONE box (run through gpurun):
#   bash tools/r6_ab.sh VAR "v1 v2" "c3 c4" [steps]      -> one line per run: workload, VAR=value, samples/s, ms per step
# Runs alternate (v1 v2 v1 v2) so clock drift of the box shows up as disagreement between repeats, not as a difference.
set -u
VAR=$1; VALS=$2; WLS=$3; STEPS=${4:-6}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r6; mkdir -p $O
for w in $WLS; do for rep in 1 2; do for v in $VALS; do
  env $VAR=$v python3 $R/bench.py --workload $w --steps $STEPS --warmup 2 --other-modes "" --no-eval --no-cpu-baseline --extra-workloads "" --no-roofline --no-parity --no-input-compare > $O/ab_${w}_${VAR}_$v.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$O/ab_${w}_${VAR}_$v.json')); print('AB $w $VAR=$v', d['value'], d['ms_per_step'])"
done; done; done
