set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
m=bf16x3-mixed
rocprofv3 --kernel-trace --stats -d $O/step_$m -o step -- python3 $R/bench.py --dtype $m --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline > $O/step_$m.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $O/step_$m -name "*.db" | head -1) $O/step_${m}_kernel_stats.csv 2> $O/step_${m}_total.txt
rm -rf $O/step_$m
head -12 $O/step_${m}_kernel_stats.csv; cat $O/step_${m}_total.txt
