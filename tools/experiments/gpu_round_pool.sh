#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -6
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other > gpurun_out/pool_bench.json 2> gpurun_out/pool_bench.err
python -c "
import json; d=json.load(open('gpurun_out/pool_bench.json')); print(d['value'], d['detail'])"
