#!/usr/bin/env python3
"""Latency of the HOST-BUFFER entry points (what a cgo caller pays: upload, launch, download, synchronise) as a function
of the batch size -- where a single Point.Mul / Suite.Pair should stay on the CPU ("keep n = 1 on
the CPU") and from which size a batch belongs on the device.  Median of 9 calls after 3 warm-ups, microseconds."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kyber_amd.group import edwards25519 as ed
from kyber_amd.pairing import bls12381 as bls, bn256 as bn
def shake(label, n): return np.frombuffer(hashlib.shake_256(label).digest(n), dtype=np.uint8)
def med(fn, reps=9, warm=3):
    for _ in range(warm): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e6
N = 4096
s = shake(b"lat/s", N * 32).reshape(N, 32).copy(); s[:, 31] &= 0x0F
P = ed.batch_mul_base(s)
k = shake(b"lat/k", N * 32).reshape(N, 32).copy(); k[:, 0] &= 0x3F
res = {"unit": "microseconds per CALL (host buffers in, host buffers out)", "sizes": [1, 4, 16, 64, 256, 1024, 4096]}
rows = {}
G1 = np.asarray(bls.g1_commit(k)[0]); G2 = np.asarray(bls.g2_commit(k)[0])
B1 = np.asarray(bn.g1_commit(k)[0]); B2 = np.asarray(bn.g2_commit(k)[0])
T = bls.F_TRUSTED(0) | bls.F_TRUSTED(1)
for name, fn in (("ed25519_mul", lambda n: ed.batch_mul(s[:n], P[:n])),
                 ("ed25519_mul_base", lambda n: ed.batch_mul_base(s[:n])),
                 ("bls12381_g1_mul_validated", lambda n: bls.g1_batch_mul(k[:n], G1[:n], bls.F_TRUSTED(0))),
                 ("bls12381_g2_mul_validated", lambda n: bls.g2_batch_mul(k[:n], G2[:n], bls.F_TRUSTED(0))),
                 ("bls12381_pair_validated", lambda n: bls.batch_pair(G1[:n], G2[:n], T)),
                 ("bls12381_validate_pairing_validated", lambda n: bls.batch_validate_pairing(G1[:n], G2[:n], G1[:n], G2[:n], T | bls.F_TRUSTED(2) | bls.F_TRUSTED(3))),
                 ("bn256_g1_mul", lambda n: bn.g1_batch_mul(k[:n], B1[:n])),
                 ("bn256_pair", lambda n: bn.batch_pair(B1[:n], B2[:n]))):
    rows[name] = [round(med(lambda: fn(n)), 1) for n in res["sizes"]]
res["latency_us"] = rows
# the reference's published single-core times per operation (BASELINE.md / docs/benchmark-app data.json, README.md)
ref = {"ed25519_mul": 349.4, "ed25519_mul_base": 60.7, "bls12381_g1_mul_validated": 120.0, "bls12381_g2_mul_validated": 275.0,
       "bls12381_pair_validated": 1600.0, "bls12381_validate_pairing_validated": 3300.0, "bn256_g1_mul": 145.0, "bn256_pair": 1630.0}
res["reference_single_core_us_per_op"] = ref
res["break_even_batch"] = {}
for name, lat in rows.items():
    be = None
    for n, l in zip(res["sizes"], lat):
        if l < ref[name] * n:
            be = n
            break
    res["break_even_batch"][name] = be
print(json.dumps(res))
