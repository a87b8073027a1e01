#!/bin/bash
# Same-box A/B of two library builds on the G1 / G2 multiplication kernels at 2^16 and 2^18 elements
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_ab3; mkdir -p $O; rm -f $O/*.jsonl
for round in 1 2; do for v in A B; do for s in "bls12381 65536" "bls12381 262144" "bn256 65536" "bn256 262144" "bn254 262144"; do set -- $s
KYBER_HIP_LIB=$PWD/kyber_amd/lib/libkyberhip_$v.so timeout 300 python tools/pair_probe.py $1 $2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.load(sys.stdin)
print(json.dumps({'v': '$v', 'suite': d['suite'], 'n': d['n'], 'g1_ms': round(d['g1_mul_ms'], 3), 'g2_ms': round(d['g2_mul_ms'], 3), 'g2_T_per_s': round(d['g2_mul_validated_per_s'])}))" | tee -a $O/ab.jsonl
done; done; done
