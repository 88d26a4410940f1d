#!/bin/bash
# Round 5: the group planner without its 256-tile limit: GEMM fuzz (any plan must be correct), then the multi-layer probe again.
mkdir -p gpurun_out/r5
timeout 900 python tools/experiments/gemm_fuzz.py 5 50 > gpurun_out/r5/gemm_fuzz2.log 2>&1; grep -v "^ok" gpurun_out/r5/gemm_fuzz2.log | tail -5
python -m pytest tests/test_gpu_fp16.py -m gpu -x -q -k "group or wgrad" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity --warmup 3"
for rep in 1 2; do
  for n in 1 2; do
    AFM_WGRAD_LAYERS=$n python bench.py $Q --steps 4 --workload c4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 layers $n rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
    AFM_WGRAD_LAYERS=$n python bench.py $Q --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 layers $n rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
  done
done | tee gpurun_out/r5/wgrad_layers2.log
