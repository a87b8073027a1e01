#!/usr/bin/env python3
"""MSM timing probe: Ed25519, BLS12-381 G1/G2, bn256 G1/G2 at n points (device-resident inputs)."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kyber_amd.group import edwards25519 as ed
from kyber_amd.pairing import bls12381 as bls, bn256 as bn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
def sc(label, n, mask_idx, mask):
    a = np.frombuffer(hashlib.shake_256(label).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); a[:, mask_idx] &= mask
    return a
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = {"n": n}
s = torch.from_numpy(sc(b"e/s", n, 31, 0x0F)).cuda(); h = torch.from_numpy(sc(b"e/h", n, 31, 0x0F)).cuda()
P = ed.batch_mul_base(h)
res["ed25519_msm_ms"] = timeit(lambda: ed.msm(s, P))
for name, m, n2 in (("bls12381", bls, n), ("bn256", bn, n)):
    k = torch.from_numpy(sc(b"k" + name.encode(), n2, 0, 0x3F)).cuda(); h = torch.from_numpy(sc(b"h" + name.encode(), n2, 0, 0x3F)).cuda()
    g1b = torch.from_numpy(np.frombuffer(m.G1_BASE, dtype=np.uint8).copy()).cuda()
    g2b = torch.from_numpy(np.frombuffer(m.G2_BASE, dtype=np.uint8).copy()).cuda()
    P1, _ = m._mul(1, h, g1b, True)
    res[name + "_g1_msm_ms"] = timeit(lambda: m.g1_msm(k, P1))
    nq = min(n2, 1 << 18)
    P2, _ = m._mul(2, h[:nq], g2b, True)
    res[name + "_g2_msm_ms_n%d" % nq] = timeit(lambda: m.g2_msm(k[:nq], P2))
print(json.dumps(res))
