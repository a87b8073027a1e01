#!/bin/bash
# round 6: the MSM's light decode kernel (vouched-for uncompressed points) -- parity, A/B (KYB_MSM_DECODE=full), trace
set -u
O=gpurun_out/r06_light; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_callers.py "tests/test_gpu_full_size.py::test_msm_at_config_size_against_an_independent_expectation" tests/test_gpu_full_digest.py::test_bls12381_config2_msm_against_the_reference_shaped_sum -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
tag() { sed "s/^{/{\"run\": \"$1\", /"; }
for i in 1 2 3; do
  KYB_MSM_DECODE=full timeout 300 python tools/msm_bls_probe.py 1048576 20 affine | tag full >> $O/ab.jsonl 2>$O/err.log
  timeout 300 python tools/msm_bls_probe.py 1048576 20 affine | tag light >> $O/ab.jsonl 2>>$O/err.log
done
for n in 4096 65536 262144 4194304; do
  KYB_MSM_DECODE=full timeout 300 python tools/msm_bls_probe.py $n 20 affine | tag full >> $O/ab.jsonl 2>>$O/err.log
  timeout 300 python tools/msm_bls_probe.py $n 20 affine | tag light >> $O/ab.jsonl 2>>$O/err.log
done
cat $O/ab.jsonl
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o light_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/light_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -8 $O/light_trace.txt
