#!/bin/bash
# BLS12-381 tests, then same-box A/B of the fused verification: G = general product check, V = fixed-generator lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_verify_ab; mkdir -p $O; rm -f $O/*.jsonl
[ -n "$SKIP_TESTS" ] || python -m pytest tests -x -q -m gpu -k "bls or pairing or verify or smoke" 2>&1 | tail -3 | tee $O/pytest.log
for round in 1 2; do for v in G V; do
KYBER_HIP_LIB=$PWD/kyber_amd/lib/libkyberhip_$v.so timeout 300 python tools/verify_probe.py 2>>$O/err.log | tail -1 | sed "s/^{/{\"v\": \"$v\", /" | tee -a $O/ab.jsonl
done; done
