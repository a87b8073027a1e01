#!/bin/bash
# Same-box A/B of library builds (VARIANTS, default "O N") on the tower-machine entry points: pair / check probes of the
# three suites and the fused verification
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_ab4; mkdir -p $O; rm -f $O/*.jsonl
for round in 1 2; do for v in ${VARIANTS:-O N}; do
for s in "bls12381 65536" "bn256 262144" "bn254 262144"; do set -- $s
KYBER_HIP_LIB=$PWD/kyber_amd/lib/libkyberhip_$v.so timeout 300 python tools/pair_probe.py $1 $2 2>>$O/err.log | tail -1 | python -c "
import sys, json
d = json.load(sys.stdin)
print(json.dumps({'v': '$v', 'suite': d['suite'], 'pair_ms': round(d['pair_ms'], 3), 'pair_validated_ms': round(d['pair_validated_ms'], 3), 'check_ms': round(d['pair_check_ms'], 3)}))" | tee -a $O/ab.jsonl
done
KYBER_HIP_LIB=$PWD/kyber_amd/lib/libkyberhip_$v.so timeout 300 python tools/verify_probe.py 2>>$O/err.log | tail -1 | python -c "
import sys, json
d = json.load(sys.stdin)
print(json.dumps({'v': '$v', 'verify_ms': round(d['verify_ms'], 3), 'verify_known_keys_ms': round(d['verify_known_keys_ms'], 3), 'ok': d['all_valid_accepted'], 'rejected': d['wrong_signatures_rejected']}))" | tee -a $O/ab.jsonl
done; done
