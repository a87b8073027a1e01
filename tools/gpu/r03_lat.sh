#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_lat; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/latency_probe.py 2>/dev/null | tail -1 > $O/single_call_latency.json; head -c 1500 $O/single_call_latency.json; echo
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
