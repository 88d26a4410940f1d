#!/bin/bash
# Round 5: a layer's FFN weight gradients launched right behind the FFN backward (AFM_WGRAD_FLUSH_FFN=1) against one group per layer.
mkdir -p gpurun_out/r5
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity --warmup 3"
for rep in 1 2; do
  for f in 1 0; do
    AFM_WGRAD_FLUSH_FFN=$f python bench.py $Q --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 flush_ffn $f rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
    AFM_WGRAD_FLUSH_FFN=$f python bench.py $Q --steps 4 --workload c4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 flush_ffn $f rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
  done
done | tee gpurun_out/r5/flush_ffn.log
