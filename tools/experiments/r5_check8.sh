#!/bin/bash
# Round 5: bench.py's self-launch path on one GPU (--gpus 1 --force-ddp: parent spawns torch.distributed.run, one RCCL rank), quick form.
mkdir -p gpurun_out/r5
python bench.py --gpus 1 --force-ddp --steps 6 --warmup 2 --other-modes= --extra-workloads= --no-cpu-baseline --no-input-compare --no-eval > gpurun_out/r5/bench_forceddp.json 2> gpurun_out/r5/bench_forceddp.err
echo "rc $?"
tail -1 gpurun_out/r5/bench_forceddp.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['config']['parallelism'], d['config']['rccl_ranks'], d['roofline']['kernel'])"
tail -3 gpurun_out/r5/bench_forceddp.err | cut -c1-200
