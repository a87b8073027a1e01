#!/bin/bash
# bn254 bring-up: the new suite's GPU tests, then the whole GPU suite (tower_vm.cuh and the BN library changed under bn256 too).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_bn254; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bn254.py -x -q > $O/pytest_bn254.log 2>&1; echo "rc=$?" >> $O/pytest_bn254.log; tail -15 $O/pytest_bn254.log
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
timeout 300 python tools/pair_probe.py bn254 262144 2>/dev/null | tail -1 | tee $O/probe_bn254.json
