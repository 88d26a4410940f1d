#!/bin/bash
# cycles (GRBM_GUI_ACTIVE) and pipe-busy counters of every ablation of the pipelined dK/dV kernel: do the parts add up in CYCLES or only in time?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AFM_LIB_OVERRIDE=$R/tools/experiments/_abl/libafm_attnabl.so
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/pmc_abl -o pmc -- python3 $R/tools/experiments/attn_pipe.py --abl > $O/attn_pipe_pmc.log 2>&1
python3 $R/tools/rocpd_pmc.py $(find $O/pmc_abl -name "*.db" | head -1) dkv_pipe > $O/attn_pipe_abl_pmc.json
rm -rf $O/pmc_abl
python3 - <<PY
import json
d = json.load(open("$O/attn_pipe_abl_pmc.json"))
for k, v in sorted(d.items(), key=lambda kv: kv[1]["avg_ns"]):
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    tot = cyc * 1024
    print(f"{k[-16:]:16s} {v['avg_ns']/1e6:.4f} ms  {cyc/1e3:8.1f} kcyc  {cyc/v['avg_ns']:.3f} GHz  mfma {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/tot:.3f} valu {v.get('SQ_ACTIVE_INST_VALU',0)*4/tot:.3f} lds {v.get('SQ_ACTIVE_INST_LDS',0)*4/tot:.3f}")
PY
