#!/bin/bash
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bls12381.py -x -q > gpurun_out/pytest_bls.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_bls.log; tail -15 gpurun_out/pytest_bls.log
timeout 600 python tools/bls_probe.py 16384 > gpurun_out/bls_probe.json 2> gpurun_out/bls_probe.err; cat gpurun_out/bls_probe.json; tail -3 gpurun_out/bls_probe.err
