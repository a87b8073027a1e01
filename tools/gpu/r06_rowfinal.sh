#!/bin/bash
# round 6: the MSM's final doubling chains on the limb-per-lane arithmetic (rowfp.cuh, msm.cuh final_rows_kernel) -- parity
# (every MSM test, the full-size expectation and the whole-batch digest) and a same-box A/B against the three-lane
# final_kernel (KYB_MSM_FINAL=lanes), with a per-stage trace of each
set -u
O=gpurun_out/r06_rowfinal; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py "tests/test_gpu_full_size.py::test_msm_at_config_size_against_an_independent_expectation" tests/test_gpu_full_digest.py::test_bls12381_config2_msm_against_the_reference_shaped_sum tests/test_gpu_callers.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do
  KYB_MSM_FINAL=lanes timeout 300 python tools/msm_bls_probe.py 1048576 20 all >> $O/ab.jsonl 2>$O/err.log
  timeout 300 python tools/msm_bls_probe.py 1048576 20 all >> $O/ab.jsonl 2>>$O/err.log
done
for n in 4096 65536 262144; do
  KYB_MSM_FINAL=lanes timeout 300 python tools/msm_bls_probe.py $n 20 affine >> $O/ab.jsonl 2>>$O/err.log
  timeout 300 python tools/msm_bls_probe.py $n 20 affine >> $O/ab.jsonl 2>>$O/err.log
done
cat $O/ab.jsonl
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o rows_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/rows_trace.log 2>&1
KYB_MSM_FINAL=lanes timeout 300 rocprofv3 --kernel-trace --stats -d $O -o lanes_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/lanes_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -12 $O/rows_trace.txt; grep final $O/lanes_trace.txt
