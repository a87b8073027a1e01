#!/bin/bash
# round 6: a key's Miller lines walked by one wave on rowfp.cuh (bls12381_keylines.cuh g2_key_lines_rows) instead of one lane:
# the same-key verification tests, then the first-sight cost against the one-lane build (libkyberhip_keylane.so =
# AB_TUS=bls12381_pair tools/ab_build.sh keylane -DKYB_KEYLINES_LANE) and a trace
set -u
O=gpurun_out/r06_keylines; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_verify_same_key.py tests/test_gpu_bls12381.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
L=$PWD/kyber_amd/lib/libkyberhip_keylane.so
for i in 1 2; do
  KYBER_HIP_LIB=$L timeout 300 python tools/keyline_probe.py 64 24 2>/dev/null | tail -1 >> $O/ab.jsonl
  timeout 300 python tools/keyline_probe.py 64 24 2>/dev/null | tail -1 >> $O/ab.jsonl
done
cat $O/ab.jsonl
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o rows_trace -- python tools/keyline_probe.py 64 24 > $O/rows_trace.log 2>&1
KYBER_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d $O -o lane_trace -- python tools/keyline_probe.py 64 24 > $O/lane_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep key_lines $O/rows_trace.txt $O/lane_trace.txt
