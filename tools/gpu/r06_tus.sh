#!/bin/bash
# round 6: the MSM adapters in translation units of their own -- parity, then the timings that showed the regression
set -u
O=gpurun_out/r06_tus; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_callers.py tests/test_gpu_full_size.py tests/test_gpu_bls12381.py tests/test_gpu_bn256.py tests/test_gpu_bn254.py tests/test_gpu_devices.py tests/test_gpu_soak.py tests/test_gpu_switches.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do timeout 300 python tools/msm_bls_probe.py 1048576 20 all | tail -1; done | tee $O/g1.jsonl
timeout 600 python tools/msm_g1_128_probe.py | tee $O/g1_128.jsonl
timeout 600 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tee $O/all.jsonl
timeout 600 python tools/msm_probe.py 262144 2>/dev/null | tail -1 | tee -a $O/all.jsonl
