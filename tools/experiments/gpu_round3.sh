#!/bin/bash
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log; tail -8 gpurun_out/pytest_gpu.log
for s in bls12381 bn256; do timeout 600 python tools/pair_probe.py $s 65536 > gpurun_out/probe_$s.json 2> gpurun_out/probe_$s.err; cat gpurun_out/probe_$s.json; tail -2 gpurun_out/probe_$s.err; done
