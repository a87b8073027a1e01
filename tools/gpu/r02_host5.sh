#!/bin/bash
# Ed25519 host-buffer path at the C ABI: (streams, chunk) sweep on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_host; mkdir -p $O; rm -f $O/sweep2.jsonl
for cfg in "1 262144" "1 131072" "2 131072" "2 262144" "3 131072" "3 65536"; do set -- $cfg
KYB_PIPE_STREAMS=$1 KYB_PIPE_CHUNK=$2 timeout 600 python - <<PY | tee -a $O/sweep2.jsonl
import json, time, hashlib, numpy as np, torch
from kyber_amd import _lib
from kyber_amd.group import edwards25519 as ed
lib = _lib.load()
n = 1 << 20
s = np.frombuffer(hashlib.shake_256(b"host/s").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); s[:, 31] &= 0x0F
P = np.ascontiguousarray(ed.batch_mul_base(s))
out = np.zeros((n, 32), dtype=np.uint8); st = np.zeros(n, dtype=np.uint8)
def med(fn, k=9):
    fn(); ts = []
    for _ in range(k):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return round(sorted(ts)[k // 2] * 1e3, 3)
res = {"streams": $1, "chunk": $2, "fixed_ms": med(lambda: lib.kyb_ed25519_mul_base(n, s.ctypes.data, out.ctypes.data, 0)),
       "var_ms": med(lambda: lib.kyb_ed25519_mul(n, s.ctypes.data, P.ctypes.data, out.ctypes.data, st.ctypes.data, 0))}
print(json.dumps(res))
PY
done
