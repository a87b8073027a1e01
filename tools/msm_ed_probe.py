#!/usr/bin/env python3
"""Ed25519 MSM at n points: per-stage kernel times come from running this under rocprofv3 --kernel-trace."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kyber_amd.group import edwards25519 as ed
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
def sc(label, n):
    a = np.frombuffer(hashlib.shake_256(label).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); a[:, 31] &= 0x0F
    return a
s = torch.from_numpy(sc(b"e/s", n)).cuda(); h = torch.from_numpy(sc(b"e/h", n)).cuda()
P = ed.batch_mul_base(h)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print(json.dumps({"n": n, "ed25519_msm_ms": timeit(lambda: ed.msm(s, P))}))
