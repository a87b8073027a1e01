#!/bin/bash
# round 6: BLS12-381 G1 MSM of short scalars (KYB_F_SCALAR_BITS(b), b <= 160) on plain windows instead of GLV halves
set -u
O=gpurun_out/r06_g1plain; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_callers.py tests/test_gpu_full_size.py tests/test_gpu_bls12381.py tests/test_gpu_devices.py tests/test_gpu_soak.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python tools/msm_g1_128_probe.py | tee $O/probe.jsonl
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o t -- python tools/msm_g1_128_probe.py > $O/t.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "BlsG1MsmPlain" $O/t.txt | cut -c1-100,140-200
