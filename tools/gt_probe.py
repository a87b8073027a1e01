#!/usr/bin/env python3
"""GT exponentiation probe (device-resident inputs, HIP-event timing, median): pointGT.Mul / GTElt.Mul on the tower
machine at n elements per suite.  usage: gt_probe.py [n] [reps]"""
import hashlib, importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
def scalars(label, n):
    a = np.frombuffer(hashlib.shake_256(label).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); a[:, 0] &= 0x3F
    return a
def timeit(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
res = {"n": n}
for suite in ("bls12381", "bn256", "bn254"):
    m = importlib.import_module("kyber_amd.pairing." + suite)
    k = torch.from_numpy(scalars(b"k" + suite.encode(), n)).cuda(); h = torch.from_numpy(scalars(b"h" + suite.encode(), n)).cuda()
    g2b = torch.from_numpy(np.frombuffer(m.G2_BASE, dtype=np.uint8).copy()).cuda()
    P, _ = m.g1_commit(h); Q = g2b.repeat(n, 1)
    T = m.F_TRUSTED(0) | m.F_TRUSTED(1)
    gt, _ = m.batch_pair(P, Q, T)
    ms = timeit(lambda: m.gt_batch_mul(k, gt))
    out, st = m.gt_batch_mul(k, gt)
    kP, _ = m.g1_batch_mul(k[:512].contiguous(), P[:512].contiguous())
    ek, _ = m.batch_pair(kP, Q[:512].contiguous(), T)
    res[suite] = {"gt_mul_ms": ms, "gt_muls_per_s": n / ms * 1e3, "status_clear": not bool(st.any().item()),
                  "equals_pairing_of_multiple": bool((out[:512] == ek).all().item())}
print(json.dumps(res))
