#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_vt; mkdir -p $O; export TMPDIR=/tmp
python - <<'PY' 2>&1 | tee $O/vt_timing.json
import json, hashlib, numpy as np, torch
from kyber_amd.group import edwards25519 as ed
from kyber_amd.pairing import bn256 as bn
n = 1 << 20
raw = np.frombuffer(hashlib.shake_256(b"vt").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()
raw[:, 31] &= 0x0F
P = ed.batch_mul_base(torch.from_numpy(raw).cuda())
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
full = torch.from_numpy(raw).cuda()
short = raw.copy(); short[:, 16:] = 0; short = torch.from_numpy(short).cuda()
res = {"n": n,
       "ed25519_mul_ms_253bit": t(lambda: ed.batch_mul(full, P)),
       "ed25519_mul_ms_253bit_vartime": t(lambda: ed.batch_mul(full, P, vartime=True)),
       "ed25519_mul_ms_128bit": t(lambda: ed.batch_mul(short, P)),
       "ed25519_mul_ms_128bit_vartime": t(lambda: ed.batch_mul(short, P, vartime=True))}
m = 1 << 18
k = np.zeros((m, 32), dtype=np.uint8); k[:, 16:] = raw[:m, :16]
kk = torch.from_numpy(k).cuda()
hh = torch.from_numpy(raw[:m] & 0x3F).cuda()
P1 = bn._mul(1, hh, torch.from_numpy(np.frombuffer(bn.G1_BASE, dtype=np.uint8).copy()).cuda(), True)[0]
P2 = bn._mul(2, hh, torch.from_numpy(np.frombuffer(bn.G2_BASE, dtype=np.uint8).copy()).cuda(), True)[0]
for g, Pg, f in ((1, P1, bn.g1_msm), (2, P2, bn.g2_msm)):
    res[f"bn256_g{g}_msm_ms_2p18_128bit_scalars"] = t(lambda: f(kk, Pg, bn.F_TRUSTED(0)))
    res[f"bn256_g{g}_msm_ms_2p18_128bit_scalars_flagged"] = t(lambda: f(kk, Pg, bn.F_TRUSTED(0) | bn.F_SCALAR_BITS(128)))
print(json.dumps(res))
PY
