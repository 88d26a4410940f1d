#!/bin/bash
# Round 5: kernel statistics of greedy / beam decode in the timed mode (fp16), c2 shapes, B = 128.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/dec -o st -- python3 $R/tools/bench_decode.py c2 128 fp16 > $O/dec_fp16.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $O/dec -name "*.db" | head -1) $O/decode_fp16_kernel_stats.csv 2> $O/decode_fp16_total.txt
rm -rf $O/dec
cat $O/dec_fp16.log | grep "ms/token"; head -24 $O/decode_fp16_kernel_stats.csv | cut -c1-150; cat $O/decode_fp16_total.txt
