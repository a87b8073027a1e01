#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_callers.py tests/test_gpu_full_size.py tests/test_gpu_ed25519.py -m gpu -x -q 2>&1 | tail -5
bash tools/experiments/gpu_round_msm3.sh
