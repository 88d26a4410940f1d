# rocprofv3 --kernel-trace --stats of the command that times the roofline kernels alone (tools/bench_attn_x3.py): the per-kernel average
# must agree with bench.py's live HIP-event time of the same launches
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in bf16x3 bf16; do
  rocprofv3 --kernel-trace --stats -d $O/attnstats_$m -o st -- python3 $R/tools/bench_attn_x3.py --mode $m > $O/attnstats_$m.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find $O/attnstats_$m -name "*.db" | head -1) $O/attn_${m}_kernel_stats.csv 2> $O/attn_${m}_kernel_stats_total.txt
  rm -rf $O/attnstats_$m
  grep k_attn $O/attn_${m}_kernel_stats.csv
done
