#!/usr/bin/env python3
"""A/B probe of the per-lane kernels of one pairing suite (device-resident inputs, HIP-event timing, median of `reps`):
UnmarshalBinary, checked / validated Pair, G1 / G2 / GT Mul, G1 / G2 MSM.  Run once per library build
(KYBER_HIP_LIB=<path> selects the build under test); the outputs are hashed so that builds can be compared.
usage: ab_probe.py {bls12381|bn256|bn254} [n] [reps]"""
import hashlib, importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
suite = sys.argv[1]; m = importlib.import_module("kyber_amd.pairing." + suite)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
def scalars(label, n):
    a = np.frombuffer(hashlib.shake_256(label).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); a[:, 0] &= 0x0F
    return a
def timeit(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return round(sorted(ts)[len(ts) // 2], 3)
k = torch.from_numpy(scalars(b"k", n)).cuda(); h = torch.from_numpy(scalars(b"h", n)).cuda()
g1b = torch.from_numpy(np.frombuffer(m.G1_BASE, dtype=np.uint8).copy()).cuda()
g2b = torch.from_numpy(np.frombuffer(m.G2_BASE, dtype=np.uint8).copy()).cuda()
P, _ = m._mul(1, h, g1b, True); Q, _ = m._mul(2, k, g2b, True)
T = m.F_TRUSTED(0); E = m.ENGINE
res = {"lib": os.path.basename(os.environ.get("KYBER_HIP_LIB", "libkyberhip.so")), "suite": suite, "n": n}
sha = hashlib.sha256()
def run(name, fn, keep=True):
    res[name + "_ms"] = timeit(fn)
    if keep:
        sha.update(name.encode())
        for o in fn(): sha.update(o.cpu().numpy().tobytes()) if hasattr(o, "cpu") else sha.update(bytes(o))
run("g1_unmarshal", lambda: E.batch_unmarshal(1, P)); run("g2_unmarshal", lambda: E.batch_unmarshal(2, Q))
run("pair_checked", lambda: m.batch_pair(P, Q)); run("pair_validated", lambda: m.batch_pair(P, Q, T), False)
run("g1_mul_checked", lambda: m.g1_batch_mul(k, P)); run("g1_mul_validated", lambda: m.g1_batch_mul(k, P, T), False)
run("g2_mul_checked", lambda: m.g2_batch_mul(k, Q)); run("g2_mul_validated", lambda: m.g2_batch_mul(k, Q, T), False)
run("g1_msm_checked", lambda: E.g1_msm(k, P)); run("g1_msm_validated", lambda: E.g1_msm(k, P, T), False)
run("g2_msm_checked", lambda: E.g2_msm(k, Q)); run("g2_msm_validated", lambda: E.g2_msm(k, Q, T), False)
ng = min(n, 1 << 14); gt, _ = m.batch_pair(P[:ng], Q[:ng], T)
run("gt_mul_%d" % ng, lambda: E.gt_batch_mul(k[:ng], gt))
res["outputs_sha256"] = sha.hexdigest()
print(json.dumps(res))
