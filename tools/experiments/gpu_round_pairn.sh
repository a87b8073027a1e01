#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for n in 32768 65536 131072 262144; do timeout 600 python tools/pair_probe.py bls12381 $n; done
