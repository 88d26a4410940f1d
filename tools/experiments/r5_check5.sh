#!/bin/bash
# Round 5: GPU suite + benchmark line + the attention profile passes (all three kernels in both MFMA shapes).
mkdir -p gpurun_out/r5
rm -f gpurun_out/parity_records.jsonl
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5/gputest_check5.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r5/gputest_check5.log | tail -4
python bench.py > gpurun_out/r5/bench_check5.json 2> gpurun_out/r5/bench_check5.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/bench_check5.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v.get("value") for k, v in d.get("workloads", {}).items()}, d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"))
PY
bash tools/prof_r05.sh attn > gpurun_out/prof_r05_attn.log 2>&1
tail -3 gpurun_out/prof_r05_attn.log
