#!/bin/bash
# Round 5: the 32x32x16 dQ kernel compiled for two workgroups per CU (-DAFM_DQ_OCC=2: 256 registers, no spill) against three (168, the
# shipped build), alternating processes; then the step.
mkdir -p gpurun_out/r5
V=tools/experiments/_abl/libafm_dqocc2.so
for rep in 1 2; do
  for lib in "" $V; do
    echo "== lib [$lib] rep $rep"
    AFM_LIB_OVERRIDE=$lib python tools/experiments/attn_m16.py --time-only 2>&1 | grep "round 1" | grep -v "dkv\|fwd" | cut -c1-200
  done
done | tee gpurun_out/r5/dqocc.log
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity --warmup 3 --steps 8"
for rep in 1 2; do
  for lib in "" $V; do
    AFM_LIB_OVERRIDE=$lib python bench.py $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 lib [$lib] rep $rep', d['value'], d['ms_per_step'])"
  done
done | tee -a gpurun_out/r5/dqocc.log
