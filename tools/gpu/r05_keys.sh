#!/bin/bash
# Round 5: the key-line cache of the same-key verification (eight slots per stream), tests + the full GPU suite on this tree.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_keys; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_verify_same_key.py -m gpu -q -x > $O/pytest_keys.log 2>&1; echo "rc=$?" >> $O/pytest_keys.log; tail -6 $O/pytest_keys.log
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
