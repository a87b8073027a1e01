#!/bin/bash
set -x
mkdir -p gpurun_out/prof; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_bn256.py tests/test_gpu_msm.py -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
for s in bls12381 bn256; do timeout 600 python tools/pair_probe.py $s 65536 > gpurun_out/probe_$s.json 2> gpurun_out/probe_$s.err; cat gpurun_out/probe_$s.json; done
timeout 600 python tools/pair_probe.py bls12381 262144 > gpurun_out/probe_bls12381_2p18.json 2>/dev/null; cat gpurun_out/probe_bls12381_2p18.json
