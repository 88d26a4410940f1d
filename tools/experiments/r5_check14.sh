#!/bin/bash
# Round 5: the benchmark line with `micro_batch_alt`; HBM bytes of the cross-attention launches under the counters (item 9).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
python bench.py > $O/bench_final3.json 2> $O/bench_final3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/bench_final3.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v.get("value") for k, v in d.get("workloads", {}).items()}, d.get("micro_batch_alt"))
PY
cd /tmp && export TMPDIR=/tmp
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d $O/xattn_$i -o pmc -- python3 $R/tools/experiments/attn_cross.py --time-only > $O/xattn_$i.log 2>&1
  python3 $R/tools/rocpd_pmc.py $(find $O/xattn_$i -name "*.db" | head -1) k_attn > $O/attn_cross_pmc_$i.json
  rm -rf $O/xattn_$i
done
ls $O | grep attn_cross_pmc
