#!/bin/bash
# Round 5: the per-unit overhead the grouped weight-gradient planner charges (AFM_TN_OVH, steps of 64 tokens; default 40) against the step.
mkdir -p gpurun_out/r5
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity --warmup 3"
for rep in 1 2; do
  for ovh in 40 10 20 80 160; do
    AFM_TN_OVH=$ovh python bench.py $Q --steps 4 --workload c4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 ovh $ovh rep $rep', d['value'], d['ms_per_step'])"
  done
  for ovh in 40 10 80 160; do
    AFM_TN_OVH=$ovh python bench.py $Q --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 ovh $ovh rep $rep', d['value'], d['ms_per_step'])"
  done
done | tee gpurun_out/r5/tn_ovh.log
