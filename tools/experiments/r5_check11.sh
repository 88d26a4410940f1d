#!/bin/bash
# Round 5: the padded all-layers dK | dV buffer (engine._empty_dkv_all, AFM_DKV_PAD=0 / 1): model-level tests, then c2 / c4 step A/B.
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_shapes.py tests/test_gpu_model.py tests/test_gpu_branches.py -m gpu -x -q 2>&1 | tail -4 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --warmup 3"
for rep in 1 2; do
  for f in 1 0; do
    AFM_DKV_PAD=$f python bench.py $Q --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 pad $f rep $rep', d['value'], d['ms_per_step'])"
    AFM_DKV_PAD=$f python bench.py $Q --steps 4 --workload c4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 pad $f rep $rep', d['value'], d['ms_per_step'])"
  done
done | tee gpurun_out/r5/step_ab_dkvpad.log
