#!/bin/bash
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_msm.py -x -q > gpurun_out/pytest_msm.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_msm.log; tail -25 gpurun_out/pytest_msm.log
