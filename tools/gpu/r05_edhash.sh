#!/bin/bash
# Round 5: Ed25519 hash_to_curve with the straight-line Elligator 2 (one power, no inversion per map) against the previous commit's, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_edhash; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ed25519.py tests/test_gpu_group_conformance.py tests/test_gpu_callers.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
for lib in "" kyber_amd/lib/libkyberhip_oldedh.so "" kyber_amd/lib/libkyberhip_oldedh.so; do
KYBER_HIP_LIB=$lib timeout 300 python - <<P | tee -a $O/ab.jsonl
import json, numpy as np, torch, bench
from kyber_amd.group import edwards25519 as ed
n = 1 << 20
msgs = torch.from_numpy(bench.shake(b"edh2c", n * 32).reshape(n, 32).copy()).cuda()
dst = b"QUUX-V01-CS02-with-edwards25519_XMD:SHA-512_ELL2_RO_"; out = ed.batch_hash(msgs, dst)
import hashlib
print(json.dumps({"lib": "${lib:-straight-line}", "n": n, "hash_ms": bench.timed(lambda: ed.batch_hash(msgs, dst)), "sha256": hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]}))
P
done
