#!/bin/bash
# Ed25519 host-buffer path: chunk-size sweep on one box (KYB_PIPE_CHUNK), medians of 7
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_host; mkdir -p $O; rm -f $O/sweep.jsonl
for c in 65536 131072 196608 262144; do
KYB_PIPE_CHUNK=$c timeout 600 python - <<PY | tee -a $O/sweep.jsonl
import json, time, hashlib, numpy as np, torch
from kyber_amd.group import edwards25519 as ed
n = 1 << 20
s = np.frombuffer(hashlib.shake_256(b"host/s").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); s[:, 31] &= 0x0F
P = ed.batch_mul_base(s)
d_s, d_P = torch.from_numpy(s).cuda(), torch.from_numpy(P).cuda()
res = {"chunk": $c}
for name, fn in (("fixed", lambda: ed.batch_mul_base(s)), ("var", lambda: ed.batch_mul(s, P)), ("var_resident", lambda: (ed.batch_mul(d_s, d_P), torch.cuda.synchronize()))):
    fn(); ts = []
    for _ in range(7):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    res[name + "_ms"] = round(sorted(ts)[3] * 1e3, 3)
res["mix_scalar_muls_per_s"] = 2 * n / ((res["fixed_ms"] + res["var_ms"]) * 1e-3)
print(json.dumps(res))
PY
done
