#!/bin/bash
# Round 5: the gated data gradient (EPI 9) with contiguous read-once loads: tests, time, HBM bytes, c4 / c5 step.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
python -m pytest tests/test_gpu_fp16.py tests/test_gpu_shapes.py tests/test_gpu_ops.py -m gpu -x -q -k "glu or gated or c4 or c5" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
timeout 200 python tools/experiments/xgc_time.py 2>&1 | grep "EPI"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/x9f -o pmc -- python3 $R/tools/experiments/xgc_time.py > $O/x9f.log 2>&1
  python3 $R/tools/rocpd_pmc.py $(find $O/x9f -name "*.db" | head -1) k_gemm_nt > $O/epi9_fix_$c.json
  rm -rf $O/x9f
done
python3 - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r5"
rd = json.load(open(f"{O}/epi9_fix_FETCH_SIZE.json")); wr = json.load(open(f"{O}/epi9_fix_WRITE_SIZE.json"))
for k, e in rd.items():
    if "Li8E" in k or "Li9E" in k:
        print("fixed", k[:75], "read MB", round(e["FETCH_SIZE"] * 2048 / 1e6), "write MB", round(wr[k]["WRITE_SIZE"] * 1024 / 1e6), "us", round(e["avg_ns"] / 1e3))
PY
cd $R
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity --warmup 3"
for rep in 1 2; do
  python bench.py $Q --steps 5 --workload c4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 rep $rep', d['value'], d['ms_per_step'], d['final_loss'])"
done
python bench.py $Q --steps 8 --workload c5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5', d['value'], d['ms_per_step'], d['final_loss'])"
