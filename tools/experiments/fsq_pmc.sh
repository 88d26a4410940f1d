#!/bin/bash
# PMC passes of the fused short-query backward at the c2 cross shape:  bash tools/experiments/fsq_pmc.sh  -> gpurun_out/r6/fsq_pmc_<i>.json
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
G3="SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
i=0
for g in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d $O/pmc_$i -o pmc -- python3 $R/tools/experiments/xattn_fused.py --pmc > $O/fsq_pmc_$i.log 2>&1
  python3 $R/tools/rocpd_pmc.py $(find $O/pmc_$i -name "*.db" | head -1) k_attn_bwd > $O/fsq_pmc_$i.json
  rm -rf $O/pmc_$i
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/fsq_pmc_*.json")):
    d = json.load(open(f))
    for k, v in d.items():
        print(k[:70], {c: round(x) for c, x in v.items()})
PY
