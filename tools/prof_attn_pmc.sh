#!/bin/bash
# PMC passes (one rocprofv3 run per counter group) of the attention kernels at the C2 encoder shape.
#   bash tools/prof_attn_pmc.sh <mode> <tag> [extra bench_attn_x3.py args]  -> gpurun_out/prof/<tag>_attn_<mode>_pmc_<i>.json
set -u
R=$GRAFT_REPO_ROOT
m=${1:-fp16}; tag=${2:-r03}; shift 2
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
i=0
for g in "$G1" "$G2" ${PMC_MORE:-}; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d $O/pmc_$i -o pmc -- python3 $R/tools/bench_attn_x3.py --mode $m "$@" > $O/${tag}_attn_${m}_$i.log 2>&1
  python3 $R/tools/rocpd_pmc.py $(find $O/pmc_$i -name "*.db" | head -1) k_attn > $O/${tag}_attn_${m}_pmc_$i.json
  rm -rf $O/pmc_$i
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/${tag}_attn_${m}_pmc_*.json")):
    d = json.load(open(f))
    for k, v in d.items():
        print(k[:60], {c: round(x) for c, x in v.items()})
PY
