#!/usr/bin/env python3
"""Quick throughput probe of the BLS12-381 kernels (device-resident inputs, HIP-event timing)."""
import hashlib, json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kyber_amd.pairing import bls12381 as bls
R = bls.ORDER
def scalars(label, n):
    raw = hashlib.shake_256(label).digest(n * 32)
    a = np.frombuffer(raw, dtype=np.uint8).reshape(n, 32).copy(); a[:, 0] &= 0x3F
    return a
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 14
k = torch.from_numpy(scalars(b"k", n)).cuda(); h = torch.from_numpy(scalars(b"h", n)).cuda()
g1b = torch.from_numpy(np.frombuffer(bls.G1_BASE, dtype=np.uint8).copy()).cuda()
g2b = torch.from_numpy(np.frombuffer(bls.G2_BASE, dtype=np.uint8).copy()).cuda()
P, st = bls._mul(1, h, g1b, True); Q, st2 = bls._mul(2, k, g2b, True)
torch.cuda.synchronize(); assert not st.any().item() and not st2.any().item()
res = {"n": n}
ms = timeit(lambda: bls.g1_batch_mul(k, P)); res["g1_mul_per_s"] = n / ms * 1e3
ms = timeit(lambda: bls.g2_batch_mul(k, Q)); res["g2_mul_per_s"] = n / ms * 1e3
ms = timeit(lambda: bls.batch_pair(P, Q)); res["pair_per_s"] = n / ms * 1e3
G2 = g2b.repeat(n, 1)
ms = timeit(lambda: bls.batch_validate_pairing(P, Q, P, Q)); res["pair_check_per_s"] = n / ms * 1e3
print(json.dumps(res))
