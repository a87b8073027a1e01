#!/bin/bash
# round 6: the split tail with the first four tree levels inside reduce_coop_kernel, single-piece buckets stored by
# accumulate_kernel, the last fold two inputs per group -- parity, A/B by switch, trace
set -u
O=gpurun_out/${R06_OUT:-r06_split2}; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_msm.py "tests/test_gpu_full_size.py::test_msm_at_config_size_against_an_independent_expectation" tests/test_gpu_full_digest.py::test_bls12381_config2_msm_against_the_reference_shaped_sum tests/test_gpu_callers.py tests/test_gpu_bls12381.py tests/test_gpu_bn256.py tests/test_gpu_ed25519.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
OLD=$PWD/kyber_amd/lib/libkyberhip_oldsort.so
tag() { sed "s/^{/{\"run\": \"$1\", /"; }
for i in 1 2 3; do
  KYBER_HIP_LIB=$OLD timeout 300 python tools/msm_bls_probe.py 1048576 20 affine | tag oldlib >> $O/ab.jsonl 2>$O/err.log
  KYB_MSM_REDUCE=mul timeout 300 python tools/msm_bls_probe.py 1048576 20 affine | tag reduce_mul >> $O/ab.jsonl 2>>$O/err.log
  KYB_MSM_REDUCE=nofuse timeout 300 python tools/msm_bls_probe.py 1048576 20 affine | tag nofuse >> $O/ab.jsonl 2>>$O/err.log
  timeout 300 python tools/msm_bls_probe.py 1048576 20 affine | tag new >> $O/ab.jsonl 2>>$O/err.log
done
timeout 300 python tools/msm_bls_probe.py 1048576 20 all | tag new >> $O/ab.jsonl 2>>$O/err.log
for n in 64 4096 65536 262144 4194304; do
  KYB_MSM_REDUCE=mul timeout 300 python tools/msm_bls_probe.py $n 20 affine | tag reduce_mul >> $O/ab.jsonl 2>>$O/err.log
  timeout 300 python tools/msm_bls_probe.py $n 20 affine | tag new >> $O/ab.jsonl 2>>$O/err.log
done
KYBER_HIP_LIB=$OLD timeout 300 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tag oldlib >> $O/ab.jsonl
timeout 300 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tag new >> $O/ab.jsonl
cat $O/ab.jsonl
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o new_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/new_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -18 $O/new_trace.txt
