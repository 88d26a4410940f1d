#!/bin/bash
# Round 5: micro-batch samples in order of decreasing encoder length (AFM_SORT_BATCH=1) on the padded workloads.
mkdir -p gpurun_out/r5
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity --warmup 3"
for rep in 1 2; do
  for f in 2 0; do
    AFM_SORT_BATCH=$f python bench.py $Q --steps 8 --workload c3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 sort $f rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
    AFM_SORT_BATCH=$f python bench.py $Q --steps 5 --workload c4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 sort $f rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
  done
done | tee gpurun_out/r5/sort_batch.log
