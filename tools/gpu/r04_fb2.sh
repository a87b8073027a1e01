#!/bin/bash
# Round 4: fixed-base path with parked results + one inversion per eight elements (encode_kernel).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_fb2; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fixed_base.py tests/test_gpu_switches.py tests/test_gpu_callers.py tests/test_gpu_soak.py tests/test_gpu_devices.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for s in bls12381 bn256 bn254; do timeout 300 python tools/fb_probe.py $s 1048576 2>/dev/null | tail -1 | tee -a $O/fb_probe.jsonl | cut -c1-420; done
