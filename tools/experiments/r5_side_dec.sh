#!/bin/bash
# Round 5: the decoder layers' weight gradients on a side stream (AFM_SIDE_WGRAD=dec) against the plain order, step level, alternating.
mkdir -p gpurun_out/r5
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity --warmup 3"
for rep in 1 2; do
  for f in dec "" all; do
    AFM_SIDE_WGRAD=$f python bench.py $Q --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 side [$f] rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
  done
  for f in dec ""; do
    AFM_SIDE_WGRAD=$f python bench.py $Q --steps 4 --workload c4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 side [$f] rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
  done
done | tee gpurun_out/r5/step_ab_side_dec.log
