#!/bin/bash
# which library kernels torch picks for the c2 step's plain NT products (the Tensile kernel name spells out tile, MFMA shape, LDS use)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/blaslt_tr -o tr -- python3 $R/tools/experiments/blaslt_ref.py > $O/blaslt_names.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $O/blaslt_tr -name "*.db" | head -1) $O/blaslt_kernel_stats.csv 2> /dev/null
rm -rf $O/blaslt_tr
grep -v "afm_\|at6native\|rocclr" $O/blaslt_kernel_stats.csv | head -20
