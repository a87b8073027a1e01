#!/bin/bash
# GPU parity suite only (all -m gpu tests), log under gpurun_out/r02_tests/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_tests; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q "$@" > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
