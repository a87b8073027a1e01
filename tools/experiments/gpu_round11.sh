#!/bin/bash
set -x
mkdir -p gpurun_out/prof4; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof4 -o msm16 -- python tools/msm_probe.py 65536 > gpurun_out/prof4/msm16.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof4 -o msm20 -- python tools/msm_probe.py 1048576 > gpurun_out/prof4/msm20.log 2>&1
