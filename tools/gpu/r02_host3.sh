#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_host; mkdir -p $O; export TMPDIR=/tmp
cat > /tmp/hp.py <<'PY'
import hashlib, numpy as np, torch, sys
sys.path.insert(0, ".")
from kyber_amd.group import edwards25519 as ed
n = 1 << 20
s = np.frombuffer(hashlib.shake_256(b"host/s").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); s[:, 31] &= 0x0F
P = ed.batch_mul_base(s)
for _ in range(4): ed.batch_mul(s, P)
PY
for c in 196608 262144; do
KYB_PIPE_CHUNK=$c timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O -o hp_$c -- python /tmp/hp.py > $O/hp_$c.log 2>&1
python tools/rocpd_summary.py $O/hp_${c}_results.db > $O/hp_$c.txt 2>&1; rm -f $O/hp_${c}_results.db
head -12 $O/hp_$c.txt
done
ls $O
