#!/bin/bash
# round 6: the MSM's split tail (reduce without the per-chunk lo * run multiplication; the bit sums D_k out of the fold tree;
# one limb-per-lane doubling chain per term) and the XCD-aware scatter order -- parity, then same-box A/B by switch
# (KYB_MSM_REDUCE=mul = the tail of the previous commit, KYB_MSM_SORT_XCD=0 = tile-minor scatter order) and against the
# library of the commit before the sort change (libkyberhip_oldsort.so), traces of both tails
set -u
O=gpurun_out/r06_split; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_msm.py "tests/test_gpu_full_size.py::test_msm_at_config_size_against_an_independent_expectation" tests/test_gpu_full_digest.py::test_bls12381_config2_msm_against_the_reference_shaped_sum tests/test_gpu_callers.py tests/test_gpu_bls12381.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
OLD=$PWD/kyber_amd/lib/libkyberhip_oldsort.so
tag() { sed "s/^{/{\"run\": \"$1\", /"; }
for i in 1 2; do
  KYBER_HIP_LIB=$OLD timeout 300 python tools/msm_bls_probe.py 1048576 20 all | tag oldlib >> $O/ab.jsonl 2>$O/err.log
  KYB_MSM_REDUCE=mul timeout 300 python tools/msm_bls_probe.py 1048576 20 all | tag reduce_mul >> $O/ab.jsonl 2>>$O/err.log
  KYB_MSM_SORT_XCD=0 timeout 300 python tools/msm_bls_probe.py 1048576 20 all | tag xcd0 >> $O/ab.jsonl 2>>$O/err.log
  timeout 300 python tools/msm_bls_probe.py 1048576 20 all | tag new >> $O/ab.jsonl 2>>$O/err.log
done
for n in 64 4096 65536 262144 4194304; do
  KYB_MSM_REDUCE=mul timeout 300 python tools/msm_bls_probe.py $n 20 affine | tag reduce_mul >> $O/ab.jsonl 2>>$O/err.log
  timeout 300 python tools/msm_bls_probe.py $n 20 affine | tag new >> $O/ab.jsonl 2>>$O/err.log
done
cat $O/ab.jsonl
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o new_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/new_trace.log 2>&1
KYB_MSM_REDUCE=mul KYB_MSM_SORT_XCD=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O -o mul_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/mul_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -16 $O/new_trace.txt; grep -E "reduce|fold|final|scatter" $O/mul_trace.txt
