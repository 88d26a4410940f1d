#!/bin/bash
# Round 5: samples/s of the c2 step against the micro-batch size at a fixed global batch of 512 (B x accumulate): do activations that fit
# the 256-MB memory-side cache (B = 64: 67 ... 268 MB per tensor) pay more than the smaller launches cost?
mkdir -p gpurun_out/r5
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity --warmup 2 --steps 6"
for rep in 1 2; do
  for ba in "128 4" "64 8" "32 16" "256 2"; do
    set -- $ba
    python bench.py $Q --batch $1 --acc $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B $1 acc $2 rep $rep', d['value'], d['ms_per_step'], d.get('step_hbm_gbs'))"
  done
done | tee gpurun_out/r5/batch_sweep.log
