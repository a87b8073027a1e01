#!/usr/bin/env python3
"""Throughput probe of a pairing suite's kernels (device-resident inputs, HIP-event timing).
usage: pair_probe.py {bls12381|bn256|bn254} [n]"""
import hashlib, importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
suite = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
m = importlib.import_module("kyber_amd.pairing." + suite)
def scalars(label, n):
    a = np.frombuffer(hashlib.shake_256(label).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); a[:, 0] &= 0x3F
    return a
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
k = torch.from_numpy(scalars(b"k", n)).cuda(); h = torch.from_numpy(scalars(b"h", n)).cuda()
g1b = torch.from_numpy(np.frombuffer(m.G1_BASE, dtype=np.uint8).copy()).cuda()
g2b = torch.from_numpy(np.frombuffer(m.G2_BASE, dtype=np.uint8).copy()).cuda()
P, st = m._mul(1, h, g1b, True); Q, st2 = m._mul(2, k, g2b, True)
torch.cuda.synchronize(); assert not st.any().item() and not st2.any().item()
res = {"suite": suite, "n": n}
ms = timeit(lambda: m.g1_batch_mul(k, P)); res["g1_mul_per_s"] = n / ms * 1e3; res["g1_mul_ms"] = ms
ms = timeit(lambda: m.g2_batch_mul(k, Q)); res["g2_mul_per_s"] = n / ms * 1e3; res["g2_mul_ms"] = ms
ms = timeit(lambda: m.batch_pair(P, Q)); res["pair_per_s"] = n / ms * 1e3; res["pair_ms"] = ms
ms = timeit(lambda: m.batch_validate_pairing(P, Q, P, Q)); res["pair_check_per_s"] = n / ms * 1e3; res["pair_check_ms"] = ms
T = m.F_TRUSTED(0) | m.F_TRUSTED(1)
ms = timeit(lambda: m.batch_pair(P, Q, T)); res["pair_validated_per_s"] = n / ms * 1e3; res["pair_validated_ms"] = ms
ms = timeit(lambda: m.g2_batch_mul(k, Q, m.F_TRUSTED(0))); res["g2_mul_validated_per_s"] = n / ms * 1e3
print(json.dumps(res))
