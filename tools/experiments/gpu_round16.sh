#!/bin/bash
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ed25519.py tests/test_gpu_msm.py -x -q > gpurun_out/pytest_ed.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_ed.log; tail -4 gpurun_out/pytest_ed.log
timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other > gpurun_out/bench_ed.json 2> gpurun_out/bench_ed.err; tail -2 gpurun_out/bench_ed.err; python -c "
import json; d=json.load(open('gpurun_out/bench_ed.json')); print(d['value'], d['detail'])"
