#!/bin/bash
# Round 5: padded stored-factor rows of the gated FFN (engine._empty_factors, AFM_FACTOR_PAD=0 / 1): tests, then the c4 / c5 step A/B.
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_fp16.py tests/test_gpu_shapes.py tests/test_gpu_x3.py -m gpu -x -q -k "glu or shape or c4 or c5 or gated" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
timeout 300 python tools/experiments/ld_pad.py --c4 2>&1 | grep "glu"
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --warmup 3"
for rep in 1 2; do
  for f in 1 0; do
    AFM_FACTOR_PAD=$f python bench.py $Q --steps 5 --workload c4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 factor pad $f rep $rep', d['value'], d['ms_per_step'])"
  done
done | tee gpurun_out/r5/step_ab_factorpad.log
for f in 1 0; do
  AFM_FACTOR_PAD=$f python bench.py $Q --steps 8 --workload c5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 factor pad $f', d['value'], d['ms_per_step'])"
done | tee -a gpurun_out/r5/step_ab_factorpad.log
