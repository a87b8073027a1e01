#!/bin/bash
# round 6: fp_pow_words with sliding windows of five bits against the fixed four-bit windows (library built with
# -DKYB_POW_FIXED4 for the BLS12-381 units), where a square root per element is most of a throughput-bound call: the MSM over
# 48-byte points, UnmarshalBinary, hash-to-curve, Pair with flags = 0
set -u
O=gpurun_out/r06_pow; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_bn256.py tests/test_gpu_bn254.py tests/test_gpu_msm.py tests/test_gpu_full_digest.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
OLD=$PWD/kyber_amd/lib/libkyberhip_fixed4.so
tag() { sed "s/^{/{\"run\": \"$1\", /"; }
for i in 1 2 3; do
  KYBER_HIP_LIB=$OLD timeout 300 python tools/msm_bls_probe.py 1048576 10 all | tag fixed4 >> $O/ab.jsonl 2>$O/err.log
  timeout 300 python tools/msm_bls_probe.py 1048576 10 all | tag slide5 >> $O/ab.jsonl 2>>$O/err.log
done
for i in 1 2; do
  KYBER_HIP_LIB=$OLD timeout 300 python tools/pair_probe.py bls12381 65536 2>/dev/null | tail -1 | tag fixed4 >> $O/pair.jsonl
  timeout 300 python tools/pair_probe.py bls12381 65536 2>/dev/null | tail -1 | tag slide5 >> $O/pair.jsonl
  KYBER_HIP_LIB=$OLD timeout 300 python tools/verify_probe.py 65536 2>/dev/null | tail -1 | tag fixed4 >> $O/pair.jsonl
  timeout 300 python tools/verify_probe.py 65536 2>/dev/null | tail -1 | tag slide5 >> $O/pair.jsonl
done
cat $O/ab.jsonl; cut -c1-700 $O/pair.jsonl
