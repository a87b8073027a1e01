#!/bin/bash
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_callers.py -x -q > gpurun_out/pytest_callers.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_callers.log; tail -30 gpurun_out/pytest_callers.log
