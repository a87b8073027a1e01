#!/bin/bash
# Round 5: G2 hash_to_curve with two powers per SSWU map instead of four: tests, then hash_g1 / hash_g2 timings at 2^16
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_hash2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_callers.py tests/test_gpu_group_conformance.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
timeout 300 python - <<'P' | tee $O/hash_timing.json
import json, numpy as np, torch, bench
from kyber_amd.pairing import bls12381 as m
n = 1 << 16
msgs = torch.from_numpy(bench.shake(b"h2c", n * 32).reshape(n, 32).copy()).cuda()
print(json.dumps({"n": n, "hash_g1_ms": bench.timed(lambda: m.batch_hash_g1(msgs)), "hash_g2_ms": bench.timed(lambda: m.batch_hash_g2(msgs))}))
P
KYBER_HIP_LIB=kyber_amd/lib/libkyberhip_oldh2.so timeout 300 python - <<'P' | tee $O/hash_timing_before.json
import json, numpy as np, torch, bench
from kyber_amd.pairing import bls12381 as m
n = 1 << 16
msgs = torch.from_numpy(bench.shake(b"h2c", n * 32).reshape(n, 32).copy()).cuda()
print(json.dumps({"lib": "four powers per G2 map (the previous commit)", "n": n, "hash_g1_ms": bench.timed(lambda: m.batch_hash_g1(msgs)), "hash_g2_ms": bench.timed(lambda: m.batch_hash_g2(msgs))}))
P
