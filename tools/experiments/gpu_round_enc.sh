#!/bin/bash
# deferred-encoding check: parity + bench
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ed25519.py tests/test_gpu_full_size.py tests/test_gpu_msm.py tests/test_gpu_callers.py -m gpu -x -q > gpurun_out/enc_tests.log 2>&1
tail -5 gpurun_out/enc_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/enc_bench.json 2> gpurun_out/enc_bench.err
cat gpurun_out/enc_bench.json
