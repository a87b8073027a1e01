#!/bin/bash
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_msm.py -x -q > gpurun_out/pytest_msm.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_msm.log; tail -4 gpurun_out/pytest_msm.log
timeout 900 python tools/msm_probe.py 1048576 > gpurun_out/msm_probe.json 2> gpurun_out/msm_probe.err; cat gpurun_out/msm_probe.json; tail -2 gpurun_out/msm_probe.err
timeout 900 python tools/msm_probe.py 65536 > gpurun_out/msm_probe_2p16.json 2>/dev/null; cat gpurun_out/msm_probe_2p16.json
