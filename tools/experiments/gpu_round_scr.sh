#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
p() { python tools/pair_probe.py bls12381 131072 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', 'pair/s %.3e'%d['pair_per_s'], 'g1 %.3e'%d['g1_mul_per_s'], 'g2 %.3e'%d['g2_mul_per_s'])"; }
p default
HSA_SCRATCH_SINGLE_LIMIT=8589934592 p single_limit_8G
HSA_SCRATCH_SINGLE_LIMIT_ASYNC=17179869184 p async_limit_16G
HSA_SCRATCH_SINGLE_LIMIT=8589934592 HSA_SCRATCH_SINGLE_LIMIT_ASYNC=17179869184 HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 p both_noasync
HSA_SCRATCH_MEM=8589934592 p scratch_mem_8G
