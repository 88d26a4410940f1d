#!/bin/bash
# Round-6 profiles (run on the GPU box through gpurun; outputs under gpurun_out/prof_r06/, copied into profiles/r06_* by hand):
#   stats    step_{c2,c3,c4}_fp16_kernel_stats.csv (+ _by_grid.csv): rocprofv3 --kernel-trace of the benchmark command, 2 timed steps
#   ddp      {c2,c4}_ddp_timeline.txt: `bench.py --gpus 1 --force-ddp` as rank 0 of a 1-rank job (the launcher environment is set
#            here, so bench.py runs in place: no process is started under the profiler), tools/rocpd_overlap.py over the trace
#   steppmc  r06_{c2,c3,c4}_fp16_step_pmc.json (tools/prof_step_pmc.sh)
#   bash tools/prof_r06.sh [stats] [ddp] [steppmc]
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r06
mkdir -p $O $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
what="${*:-stats ddp}"
COMMON="--dtype fp16 --steps 2 --warmup 1 --other-modes  --extra-workloads  --no-roofline --no-cpu-baseline --no-input-compare --no-eval"
if [[ " $what " == *" stats "* ]]; then
  for wl in ${WLS:-c2 c3 c4}; do
    timeout 600 rocprofv3 --kernel-trace -d $O/step_$wl -o step -- python3 $R/bench.py --workload $wl --dtype fp16 --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline --no-input-compare --no-eval > $O/step_${wl}_fp16.log 2>&1
    db=$(find $O/step_$wl -name "*.db" | head -1)
    python3 $R/tools/rocpd_stats.py $db $O/step_${wl}_fp16_kernel_stats.csv --from k_patch_ k_gather_rows 2> $O/step_${wl}_fp16_total.txt
    python3 $R/tools/rocpd_stats.py $db $O/step_${wl}_fp16_kernel_stats_by_grid.csv --by-grid --cluster 1.3 --from k_patch_ k_gather_rows 2>> $O/step_${wl}_fp16_total.txt
    rm -rf $O/step_$wl
  done
fi
if [[ " $what " == *" ddp "* ]]; then
  export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 AFM_DDP_STANDIN=1
  for wl in ${WLS:-c2 c4}; do
    timeout 600 rocprofv3 --kernel-trace -d $O/ddp_$wl -o ddp -- python3 $R/bench.py --gpus 1 --force-ddp --workload $wl --dtype fp16 --steps 2 --warmup 1 --other-modes "" --extra-workloads "" --no-roofline --no-cpu-baseline --no-input-compare --no-eval > $O/ddp_${wl}.log 2>&1
    db=$(find $O/ddp_$wl -name "*.db" | head -1)
    python3 $R/tools/rocpd_overlap.py $db --standin > $O/${wl}_ddp_timeline.txt 2>&1
    grep -o '"rccl_ranks": [0-9]*' $O/ddp_${wl}.log | head -1 >> $O/${wl}_ddp_timeline.txt
    rm -rf $O/ddp_$wl
  done
  unset RANK WORLD_SIZE LOCAL_RANK AFM_DDP_STANDIN
fi
if [[ " $what " == *" steppmc "* ]]; then
  for wl in ${WLS:-c2 c3 c4}; do
    bash $R/tools/prof_step_pmc.sh r06 $wl fp16 > $O/step_pmc_$wl.log 2>&1
    cp $R/gpurun_out/prof/r06_${wl}_fp16_step_pmc.json $O/ 2>/dev/null
  done
fi
ls $O
