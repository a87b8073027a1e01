#!/bin/bash
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_callers.py -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log; tail -8 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err; tail -2 gpurun_out/bench_q.err; python -c "
import json; d=json.load(open('gpurun_out/bench_q.json')); print(json.dumps(d['other_workloads'], indent=0))"
