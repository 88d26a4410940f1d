#!/bin/bash
# Round 5: LayerNorm grids = resident workgroups for the kernel's register count (by d) against the d <= 512 caps for every d (the old rule).
mkdir -p gpurun_out/r5
for rep in 1 2; do
  for d in 768 1024 512; do
    python tools/bench_ln.py --d=$d 2>&1 | grep -v amdgpu
    AFM_LN_FWD_BLOCKS=1280 AFM_LN_BWD_BLOCKS=768 python tools/bench_ln.py --d=$d 2>&1 | grep -v amdgpu
  done
done | tee gpurun_out/r5/ln_grid.log
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "layernorm or ln" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity --warmup 3"
for rep in 1 2; do
  python bench.py $Q --steps 5 --workload c4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 new rep $rep', d['value'], d['ms_per_step'])"
  AFM_LN_FWD_BLOCKS=1280 AFM_LN_BWD_BLOCKS=768 python bench.py $Q --steps 5 --workload c4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 old rep $rep', d['value'], d['ms_per_step'])"
done | tee -a gpurun_out/r5/ln_grid.log
