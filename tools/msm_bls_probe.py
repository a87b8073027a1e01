#!/usr/bin/env python3
"""BLS12-381 G1 MSM at n points in the three calling conventions -- validated uncompressed points (the form a resident
pipeline keeps), validated 48-byte points (SURVEY.md section 8d's 80 B per point), flags = 0 (every point re-validated) --
medians of `reps` HIP-event timings; per-stage kernel times come from running this under rocprofv3 --kernel-trace.
usage: msm_bls_probe.py [n] [reps] [which: all|affine]"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kyber_amd.pairing import bls12381 as bls
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
which = sys.argv[3] if len(sys.argv) > 3 else "affine"
def sc(label, n):
    a = np.frombuffer(hashlib.shake_256(label).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); a[:, 0] &= 0x3F
    return a
k = torch.from_numpy(sc(b"k", n)).cuda(); h = torch.from_numpy(sc(b"h", n)).cuda()
g1b = torch.from_numpy(np.frombuffer(bls.G1_BASE, dtype=np.uint8).copy()).cuda()
P, _ = bls._mul(1, h, g1b, True, bls.F_UNCOMPRESSED_OUT)
fl = bls.F_TRUSTED(0) | bls.F_UNCOMPRESSED
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
res = {"n": n, "groups": os.environ.get("KYB_MSM_GROUPS", "default"), "bls12381_g1_msm_trusted_unc_ms": timeit(lambda: bls.g1_msm(k, P, fl))}
if which == "all":
    Pc, _ = bls._mul(1, h, g1b, True)
    res["bls12381_g1_msm_trusted_48B_ms"] = timeit(lambda: bls.g1_msm(k, Pc, bls.F_TRUSTED(0)))
    res["bls12381_g1_msm_checked_48B_ms"] = timeit(lambda: bls.g1_msm(k, Pc))
    a, b, c = bls.g1_msm(k, P, fl)[0], bls.g1_msm(k, Pc, bls.F_TRUSTED(0))[0], bls.g1_msm(k, Pc)[0]
    res["conventions_agree"] = bool(torch.equal(a, b) and torch.equal(a, c))
print(json.dumps(res))
