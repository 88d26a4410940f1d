#!/bin/bash
# Round 5: step-level A/B of the 16x16x32 dK/dV kernel (engine flag AFM_ATTN_BWD_FLAGS), then the step statistics / counters with the
# set-up cut off (tools/rocpd_stats.py --from, tools/prof_step_reduce.py).
mkdir -p gpurun_out/r5
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --steps 8 --warmup 3"
for rep in 1 2; do
  for f in 0 16384; do
    AFM_ATTN_BWD_FLAGS=$f python bench.py $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags $f rep $rep', d['value'], d['ms_per_step'])"
  done
done | tee gpurun_out/r5/step_ab_pipe16.log
bash tools/prof_r05.sh stats steppmc > gpurun_out/prof_r05_run2.log 2>&1
tail -2 gpurun_out/prof_r05/step_c2_fp16_total.txt gpurun_out/prof_r05/step_c4_fp16_total.txt
