#!/bin/bash
# Round 5: the whole GPU suite, the benchmark line, and the profiles of the final build.
mkdir -p gpurun_out/r5
rm -f gpurun_out/parity_records.jsonl
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r5/gputest_final.log
python bench.py > gpurun_out/r5/bench_final.json 2> gpurun_out/r5/bench_final.err
bash tools/prof_r05.sh stats steppmc gemm attn > gpurun_out/prof_r05_run.log 2>&1
tail -3 gpurun_out/r5/gputest_final.log; tail -c 300 gpurun_out/r5/bench_final.err
