#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_fp16.py tests/test_gpu_ops.py tests/test_gpu_shapes.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r5/gputest_b.log
python bench.py --no-cpu-baseline --other-modes "" --extra-workloads c4 > gpurun_out/r5/bench_b.json 2> gpurun_out/r5/bench_b.err
tail -4 gpurun_out/r5/gputest_b.log; tail -c 300 gpurun_out/r5/bench_b.err
