#!/bin/bash
# Round 5: the weight gradients of n consecutive layers in one grouped launch (AFM_WGRAD_LAYERS=n) against one group per layer.
mkdir -p gpurun_out/r5
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity --warmup 3"
for rep in 1 2; do
  for n in 1 2 3 6; do
    AFM_WGRAD_LAYERS=$n python bench.py $Q --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 layers $n rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
  done
  for n in 1 2 3; do
    AFM_WGRAD_LAYERS=$n python bench.py $Q --steps 4 --workload c4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 layers $n rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
  done
done | tee gpurun_out/r5/wgrad_layers.log
