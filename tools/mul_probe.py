#!/usr/bin/env python3
"""G1 / G2 scalar-multiplication probe (device-resident inputs, HIP-event timing): every calling convention.
usage: mul_probe.py {bls12381|bn256|bn254} [n] [reps]"""
import hashlib, importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
suite = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
m = importlib.import_module("kyber_amd.pairing." + suite)
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
k = torch.from_numpy(scalars(b"k", n)).cuda(); h = torch.from_numpy(scalars(b"h", n)).cuda()
g1b = torch.from_numpy(np.frombuffer(m.G1_BASE, dtype=np.uint8).copy()).cuda()
g2b = torch.from_numpy(np.frombuffer(m.G2_BASE, dtype=np.uint8).copy()).cuda()
P, st = m._mul(1, h, g1b, True); Q, st2 = m._mul(2, k, g2b, True)
res = {"suite": suite, "n": n, "lvm_min": os.environ.get("KYB_LVM_MIN", "default")}
T = m.F_TRUSTED(0)
for g, pts, fn in ((1, P, m.g1_batch_mul), (2, Q, m.g2_batch_mul)):
    ms = timeit(lambda: fn(k, pts)); res["g%d_checked_ms" % g] = ms; res["g%d_checked_per_s" % g] = n / ms * 1e3
    ms = timeit(lambda: fn(k, pts, T)); res["g%d_trusted_ms" % g] = ms; res["g%d_trusted_per_s" % g] = n / ms * 1e3
    if suite == "bls12381":
        U = m.F_UNCOMPRESSED
        pu, _ = m._mul(g, h if g == 1 else k, g1b if g == 1 else g2b, True, m.F_UNCOMPRESSED_OUT)
        ms = timeit(lambda: fn(k, pu, T | U | m.F_UNCOMPRESSED_OUT)); res["g%d_trusted_unc_ms" % g] = ms; res["g%d_trusted_unc_per_s" % g] = n / ms * 1e3
print(json.dumps(res))
