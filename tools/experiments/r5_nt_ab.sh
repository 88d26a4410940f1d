#!/bin/bash
# Round 5, VERDICT r04 item 2: nontemporal loads of the read-once operands (GEMM epilogue multiplicands, the attention kernels' own rows),
# product library against the AFM_PRE_LOAD_NT=0 / AFM_ATTN_NT=0 build, alternating processes on one box.
#   AFM_BUILD_VARIANT=nt0 AFM_EXTRA_FLAGS="-DAFM_PRE_LOAD_NT=0 -DAFM_ATTN_NT=0" python -m multimodalanalytical_amd.csrc.build
OUT=gpurun_out/r5; mkdir -p $OUT
OLD=$PWD/tools/experiments/_abl/libafm_nt0.so
python -m pytest tests/test_gpu_fp16.py -x -q -m gpu 2>&1 | tail -3 > $OUT/nt_ab_tests.log
for r in 1 2; do
  echo "== new $r"; python tools/bench_gemm_step.py --only xsaved; python tools/bench_attn_x3.py --mode fp16
  echo "== old $r"; AFM_LIB_OVERRIDE=$OLD python tools/bench_gemm_step.py --only xsaved; AFM_LIB_OVERRIDE=$OLD python tools/bench_attn_x3.py --mode fp16
done > $OUT/nt_ab.log 2>&1
python bench.py > $OUT/bench_nt_new.json 2> $OUT/bench_nt_new.err
AFM_LIB_OVERRIDE=$OLD python bench.py --steps 6 --no-cpu-baseline --other-modes "" --extra-workloads c4 > $OUT/bench_nt_old.json 2> $OUT/bench_nt_old.err
tail -2 $OUT/nt_ab_tests.log; cat $OUT/nt_ab.log | grep -v "^$"
