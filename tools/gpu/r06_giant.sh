#!/bin/bash
# round 6: giant buckets -- the sort's giant bins in slices, the long join as two launches of the cooperative fold
set -u
O=gpurun_out/r06_giant; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_switches.py -k "msm" tests/test_gpu_msm.py tests/test_gpu_callers.py tests/test_gpu_full_size.py tests/test_gpu_soak.py tests/test_gpu_bn256.py tests/test_gpu_ed25519.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python tools/msm_g1_128_probe.py | tee $O/g1_128.jsonl
for i in 1 2; do timeout 300 python tools/msm_bls_probe.py 1048576 20 affine | tail -1; done | tee $O/g1.jsonl
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o t -- python tools/msm_g1_128_probe.py > $O/t.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "giant|bucket_long|fine_sort" $O/t.txt | cut -d"|" -f1,2,4,5,6
