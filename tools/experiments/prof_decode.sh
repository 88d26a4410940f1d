set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/dec -o st -- python3 $R/tools/bench_decode.py c2 128 bf16x3 > $O/dec.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $O/dec -name "*.db" | head -1) $O/decode_bf16x3_kernel_stats.csv 2> $O/decode_total.txt
rm -rf $O/dec
head -14 $O/decode_bf16x3_kernel_stats.csv | cut -c1-160; cat $O/decode_total.txt
