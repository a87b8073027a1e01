#!/bin/bash
# Round 4: same-key verification (program VERIFYK): tests, then timing against the general fused verification.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_vk; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_verify_same_key.py tests/test_gpu_bls12381.py tests/test_gpu_devices.py tests/test_scalar_poly.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 300 python tools/verify_probe.py 65536 2>/dev/null | tail -1 | tee $O/verify_probe.json
