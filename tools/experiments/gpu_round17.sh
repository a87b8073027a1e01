#!/bin/bash
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_bn256.py -x -q > gpurun_out/pytest_p.log 2>&1; tail -3 gpurun_out/pytest_p.log
for n in 65536 262144; do for s in bls12381 bn256; do timeout 600 python tools/pair_probe.py $s $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['suite'], d['n'], 'pair/s %.4g check/s %.4g' % (d['pair_per_s'], d['pair_check_per_s']))"; done; done
