#!/bin/bash
# Round 5: the 16x16x32 attention-backward defaults -- experiment checks, the GPU suite, the benchmark line.
mkdir -p gpurun_out/r5
python tools/experiments/attn_m16.py > gpurun_out/r5/attn_m16_default.log 2>&1
grep -c ok gpurun_out/r5/attn_m16_default.log; grep -n "FAIL\|ALL OK" gpurun_out/r5/attn_m16_default.log | head
rm -f gpurun_out/parity_records.jsonl
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r5/gputest_check3.log
tail -3 gpurun_out/r5/gputest_check3.log
python bench.py > gpurun_out/r5/bench_check3.json 2> gpurun_out/r5/bench_check3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/bench_check3.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v.get("value") for k, v in d.get("workloads", {}).items()})
PY
