#!/bin/bash
# Round 5: LayerNorm backward with its padded-row flags one iteration ahead: tests, then the padded workloads' steps and the kernel's rate.
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_ops.py tests/test_gpu_shapes.py tests/test_gpu_branches.py -m gpu -x -q -k "layernorm or ln or padded or c3 or c4 or skip or hint" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity --warmup 3"
for rep in 1 2; do
  python bench.py $Q --steps 8 --workload c3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
  python bench.py $Q --steps 5 --workload c4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
done | tee gpurun_out/r5/ln_flags.log
bash tools/experiments/r5_c3_prof.sh > /dev/null 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/prof_r05/r05_c3_fp16_step_pmc.json"))
print({k: v for k, v in d.items() if k in ("kernel_ms_per_step", "hbm_gbs")})
for k, e in d["by_kernel"].items():
    if "k_ln" in k: print(k[:50], e, round(e["hbm_gb_per_step"] / e["ms_per_step"], 2), "TB/s")
PY
