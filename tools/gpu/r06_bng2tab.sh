#!/bin/bash
# round 6: the BN G2 ladders' window tables in a lane-contiguous global slab instead of private scratch (bn_suite.inc
# g2_mul_gls_lz<TAB>) -- parity, then a same-box A/B against the scratch tables (libkyberhip_g2scratch.so =
# AB_TUS="bn256 bn254" tools/ab_build.sh g2scratch -DKYB_BN_G2_TAB_SCRATCH), trace + FETCH / WRITE counters of each
set -u
O=gpurun_out/r06_bng2tab; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bn256.py tests/test_gpu_bn254.py "tests/test_gpu_full_digest.py::test_bn256_config4_whole_batch_digest" -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
L=$PWD/kyber_amd/lib/libkyberhip_g2scratch.so
for i in 1 2; do
  for s in bn256 bn254; do
    KYBER_HIP_LIB=$L timeout 300 python tools/mul_probe.py $s 262144 7 2>/dev/null | tail -1 | sed 's/^{/{"lib": "scratch", /' >> $O/ab.jsonl
    timeout 300 python tools/mul_probe.py $s 262144 7 2>/dev/null | tail -1 | sed 's/^{/{"lib": "slab", /' >> $O/ab.jsonl
  done
done
cat $O/ab.jsonl
for v in slab scratch; do
  E=""; [ $v = scratch ] && E="KYBER_HIP_LIB=$L"
  env $E timeout 300 rocprofv3 --kernel-trace --stats -d $O -o ${v}_trace -- python tools/mul_probe.py bn256 262144 3 > $O/${v}_trace.log 2>&1
  env $E timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o ${v}_fetch -- python tools/mul_probe.py bn256 262144 3 > $O/${v}_fetch.log 2>&1
  env $E timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o ${v}_write -- python tools/mul_probe.py bn256 262144 3 > $O/${v}_write.log 2>&1
done
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep g2_mul $O/*_trace.txt $O/*_fetch.txt $O/*_write.txt | head -20
