#!/bin/bash
# Round 5: the GPU suite and the benchmark line on the last build of the round.
mkdir -p gpurun_out/r5
rm -f gpurun_out/parity_records.jsonl
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 > gpurun_out/r5/gputest_final2.log
cat gpurun_out/r5/gputest_final2.log
python bench.py > gpurun_out/r5/bench_final2.json 2> gpurun_out/r5/bench_final2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/bench_final2.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v.get("value") for k, v in d.get("workloads", {}).items()}, d["modes"]["fp16"]["logits_vs_cpu_reference"]["logits_rel_err"])
PY
