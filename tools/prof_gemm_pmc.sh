#!/bin/bash
# PMC passes (one rocprofv3 run per counter group, --kernel-trace --pmc only) of the step's dominant NT GEMM launches.
#   bash tools/prof_gemm_pmc.sh <tag> [c2|c4]  -> gpurun_out/prof/<tag>_gemm_fp16_pmc.json / <tag>_c4_gemm_fp16_pmc.json (merged, with derived figures)
set -u
R=$GRAFT_REPO_ROOT
tag=${1:-r04}
SET=${2:-c2}
OUTN=${tag}_gemm_fp16_pmc.json
[ "$SET" != "c2" ] && OUTN=${tag}_${SET}_gemm_fp16_pmc.json
REPS=${REPS:-12}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
G3="FETCH_SIZE"
G4="WRITE_SIZE"
G5="TCC_HIT_sum TCC_MISS_sum"
i=0
for g in "$G1" "$G2" "$G3" "$G4" "$G5"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $g -d $O/gpmc_$i -o pmc -- python3 $R/tools/prof_gemm_run.py --reps $REPS --set $SET > $O/${tag}_gemm_$i.log 2>&1
  cp $(find $O/gpmc_$i -name "*.db" | head -1) $O/${tag}_gemm_pass$i.db 2>/dev/null
  rm -rf $O/gpmc_$i
done
python3 $R/tools/prof_gemm_reduce.py $O $tag $REPS $SET > $O/$OUTN
rm -f $O/${tag}_gemm_pass*.db
cat $O/$OUTN | head -150
