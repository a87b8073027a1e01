#!/bin/bash
# thirty seeds of fresh inputs through every suite against the oracles, on the round's final binaries (lazy-limb ladders included)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r05_soak
KYB_SOAK_SEEDS=30 timeout 2400 python -m pytest tests/test_gpu_soak.py -m gpu -q --durations=5 > gpurun_out/r05_soak/soak.txt 2>&1; tail -12 gpurun_out/r05_soak/soak.txt
