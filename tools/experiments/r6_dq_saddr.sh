#!/bin/bash
# Round 6: the dQ kernel's ring pieces in saddr form (dma_piece_s: no spilled lane addresses, no scratch reload inside the tile loop) against
# the 64-bit lane addresses of rounds 2-5 (-DAFM_DQ_OLD_DMA build), alternating processes; then the c2 step.
#   AFM_BUILD_VARIANT=dqold AFM_EXTRA_FLAGS=-DAFM_DQ_OLD_DMA python -m multimodalanalytical_amd.csrc.build     (before gpurun)
mkdir -p gpurun_out/r6
V=$PWD/tools/experiments/_abl/libafm_dqold.so
for rep in 1 2; do
  for lib in "" $V; do
    echo "== lib [${lib##*/}] rep $rep"
    AFM_LIB_OVERRIDE=$lib python tools/experiments/dq_time.py 2>&1 | grep "^dQ"
  done
done | tee gpurun_out/r6/dq_saddr.log
Q="--other-modes= --extra-workloads= --no-roofline --no-cpu-baseline --no-input-compare --no-eval --no-parity --warmup 2 --steps 6"
for rep in 1 2; do
  for lib in "" $V; do
    AFM_LIB_OVERRIDE=$lib python bench.py $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 lib [${lib##*/}] rep $rep', d['value'], d['ms_per_step'])"
  done
done | tee -a gpurun_out/r6/dq_saddr.log
