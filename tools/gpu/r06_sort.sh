#!/bin/bash
# round 6: the MSM's counting sort with tile-major counters (whole-line stores, a per-bucket walk over the tiles instead of a
# scan of nwin x nb x tiles counters) and a grid of whole rounds of the chip -- parity, then a same-box A/B against the
# library built from the previous msm.cuh (kyber_amd/lib/libkyberhip_oldsort.so), tile counts forced both ways, traces
set -u
O=gpurun_out/r06_sort; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_msm.py "tests/test_gpu_full_size.py::test_msm_at_config_size_against_an_independent_expectation" tests/test_gpu_full_digest.py::test_bls12381_config2_msm_against_the_reference_shaped_sum tests/test_gpu_callers.py tests/test_gpu_bn256.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
OLD=$PWD/kyber_amd/lib/libkyberhip_oldsort.so
for i in 1 2; do
  KYBER_HIP_LIB=$OLD timeout 300 python tools/msm_bls_probe.py 1048576 20 all | sed 's/^{/{"lib": "old", /' >> $O/ab.jsonl 2>$O/err.log
  timeout 300 python tools/msm_bls_probe.py 1048576 20 all | sed 's/^{/{"lib": "new", /' >> $O/ab.jsonl 2>>$O/err.log
done
for t in 14 24 28 32 56; do
  KYB_MSM_SORT_TILES=$t timeout 300 python tools/msm_bls_probe.py 1048576 20 affine | sed "s/^{/{\"lib\": \"new\", \"tiles\": $t, /" >> $O/ab.jsonl 2>>$O/err.log
done
for n in 4096 65536 262144 4194304; do
  KYBER_HIP_LIB=$OLD timeout 300 python tools/msm_bls_probe.py $n 20 affine | sed 's/^{/{"lib": "old", /' >> $O/ab.jsonl 2>>$O/err.log
  timeout 300 python tools/msm_bls_probe.py $n 20 affine | sed 's/^{/{"lib": "new", /' >> $O/ab.jsonl 2>>$O/err.log
done
KYBER_HIP_LIB=$OLD timeout 300 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | sed 's/^{/{"lib": "old", /' >> $O/ed.jsonl
timeout 300 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | sed 's/^{/{"lib": "new", /' >> $O/ed.jsonl
cat $O/ab.jsonl $O/ed.jsonl
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o new_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/new_trace.log 2>&1
KYBER_HIP_LIB=$OLD timeout 300 rocprofv3 --kernel-trace --stats -d $O -o old_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/old_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -24 $O/new_trace.txt; grep -E "hist|scatter|scan|offs" $O/old_trace.txt
