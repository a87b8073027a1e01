#!/bin/bash
# round 6: the two-stage cooperative reduce of the MSM tail (msm.cuh RS) -- parity (MSM tests at every size, full-size
# expectation test) and a same-box A/B against the one-stage kernel (KYB_MSM_REDUCE=1), with a per-stage trace of each
set -u
O=gpurun_out/r06_reduce2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_msm.py "tests/test_gpu_full_size.py::test_msm_at_config_size_against_an_independent_expectation" tests/test_gpu_full_digest.py::test_bls12381_config2_msm_against_the_reference_shaped_sum -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do
  KYB_MSM_REDUCE=1 timeout 300 python tools/msm_bls_probe.py 1048576 20 all >> $O/ab.jsonl 2>$O/err.log
  timeout 300 python tools/msm_bls_probe.py 1048576 20 all >> $O/ab.jsonl 2>>$O/err.log
done
for n in 65536 262144; do
  KYB_MSM_REDUCE=1 timeout 300 python tools/msm_bls_probe.py $n 20 affine >> $O/ab.jsonl 2>>$O/err.log
  timeout 300 python tools/msm_bls_probe.py $n 20 affine >> $O/ab.jsonl 2>>$O/err.log
done
cat $O/ab.jsonl
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o two_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/two_trace.log 2>&1
KYB_MSM_REDUCE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O -o one_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/one_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -14 $O/two_trace.txt; head -14 $O/one_trace.txt
