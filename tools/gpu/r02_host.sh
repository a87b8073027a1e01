#!/bin/bash
# Ed25519 host-buffer path (page-locked staging pipeline): the parity test of the pipelined path, then its rate.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_host; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_full_size.py -k "pipelined or config2" tests/test_gpu_ed25519.py tests/test_gpu_devices.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python - <<'PY' | tee $O/host_rate.json
import json, time, hashlib, numpy as np, torch
from kyber_amd.group import edwards25519 as ed
n = 1 << 20
s = np.frombuffer(hashlib.shake_256(b"host/s").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); s[:, 31] &= 0x0F
P = ed.batch_mul_base(s)
res = {}
for name, fn in (("fixed", lambda: ed.batch_mul_base(s)), ("var", lambda: ed.batch_mul(s, P))):
    fn(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    res[name + "_ms"] = sorted(ts)[2] * 1e3
res["mix_scalar_muls_per_s"] = 2 * n / ((res["fixed_ms"] + res["var_ms"]) * 1e-3)
print(json.dumps(res))
PY
