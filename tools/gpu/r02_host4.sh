#!/bin/bash
# Ed25519 host-buffer entry points timed at the C ABI with reused (already touched) buffers: the library's own time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_host; mkdir -p $O
timeout 600 python - <<'PY' | tee $O/host_cabi.json
import json, time, hashlib, numpy as np, torch
from kyber_amd import _lib
from kyber_amd.group import edwards25519 as ed
lib = _lib.load()
n = 1 << 20
s = np.frombuffer(hashlib.shake_256(b"host/s").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); s[:, 31] &= 0x0F
P = np.ascontiguousarray(ed.batch_mul_base(s))
out = np.zeros((n, 32), dtype=np.uint8); st = np.zeros(n, dtype=np.uint8)
d_s, d_P = torch.from_numpy(s).cuda(), torch.from_numpy(P).cuda()
def med(fn, k=9):
    fn(); ts = []
    for _ in range(k):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return round(sorted(ts)[k // 2] * 1e3, 3)
res = {"fixed_ms": med(lambda: lib.kyb_ed25519_mul_base(n, s.ctypes.data, out.ctypes.data, 0)),
       "var_ms": med(lambda: lib.kyb_ed25519_mul(n, s.ctypes.data, P.ctypes.data, out.ctypes.data, st.ctypes.data, 0)),
       "var_python_wrapper_ms": med(lambda: ed.batch_mul(s, P)),
       "var_resident_ms": med(lambda: (ed.batch_mul(d_s, d_P), torch.cuda.synchronize()))}
res["mix_scalar_muls_per_s"] = 2 * n / ((res["fixed_ms"] + res["var_ms"]) * 1e-3)
print(json.dumps(res))
PY
