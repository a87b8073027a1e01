#!/usr/bin/env python3
"""BLS12-381 G1 MSM at n points with validated, uncompressed inputs (the form a resident pipeline keeps): per-stage
kernel times come from running this under rocprofv3 --kernel-trace."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kyber_amd.pairing import bls12381 as bls
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
def sc(label, n):
    a = np.frombuffer(hashlib.shake_256(label).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); a[:, 0] &= 0x3F
    return a
k = torch.from_numpy(sc(b"k", n)).cuda(); h = torch.from_numpy(sc(b"h", n)).cuda()
g1b = torch.from_numpy(np.frombuffer(bls.G1_BASE, dtype=np.uint8).copy()).cuda()
P, _ = bls._mul(1, h, g1b, True, bls.F_UNCOMPRESSED_OUT)
fl = bls.F_TRUSTED(0) | bls.F_UNCOMPRESSED
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print(json.dumps({"n": n, "bls12381_g1_msm_trusted_unc_ms": timeit(lambda: bls.g1_msm(k, P, fl))}))
