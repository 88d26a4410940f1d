#!/bin/bash
# Round 5: HBM read bytes of c4's gated data gradient (EPI 9) and up-projection (EPI 8) under the tile-walk settings (AFM_NT_XGC).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in auto 1 4 8; do
  if [ $g = auto ]; then unset AFM_NT_XGC; else export AFM_NT_XGC=$g; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/x9_$g -o pmc -- python3 $R/tools/experiments/xgc_time.py > $O/x9_$g.log 2>&1
    python3 $R/tools/rocpd_pmc.py $(find $O/x9_$g -name "*.db" | head -1) k_gemm_nt > $O/epi9_${g}_$c.json
    rm -rf $O/x9_$g
  done
done
python3 - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r5"
for g in ("auto", "1", "4", "8"):
    rd = json.load(open(f"{O}/epi9_{g}_FETCH_SIZE.json")); wr = json.load(open(f"{O}/epi9_{g}_WRITE_SIZE.json"))
    for k, e in rd.items():
        if "Li8E" in k or "Li9E" in k:
            print(g, k[:75], "read MB", round(e["FETCH_SIZE"] * 2048 / 1e6), "write MB", round(wr[k]["WRITE_SIZE"] * 1024 / 1e6), "us", round(e["avg_ns"] / 1e3))
PY
