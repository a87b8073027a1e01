#!/usr/bin/env python3
"""Same-base batch (share.PriPoly.Commit shape) timing: n scalars times ONE base, device-resident, HIP events.
KYB_FB_MIN=0 disables the fixed-base table (the variable-base kernels run: the before figure); one JSON line.
usage: fb_probe.py {bls12381|bn256|bn254} [n]"""
import hashlib, importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
suite = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
m = importlib.import_module("kyber_amd.pairing." + suite)
s = torch.from_numpy(np.frombuffer(hashlib.shake_256(b"fb").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()).cuda()
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = {"suite": suite, "n": n, "fb_min": os.environ.get("KYB_FB_MIN", "default")}
h = np.frombuffer((987654321).to_bytes(32, "big"), dtype=np.uint8).reshape(1, 32)
for grp, commit in ((1, m.g1_commit), (2, m.g2_commit)):
    nn = n if grp == 1 else min(n, 1 << 18)
    P = torch.from_numpy(np.asarray(commit(h, flags=0)[0])[0].copy()).cuda()
    Q = torch.from_numpy(np.asarray(commit(h[:, ::-1].copy(), flags=0)[0])[0].copy()).cuda()
    ms = timeit(lambda: commit(s[:nn], P)); res[f"g{grp}_same_base_ms"] = ms; res[f"g{grp}_same_base_per_s"] = nn / ms * 1e3
    # alternating bases: every call rebuilds the table
    t = [P, Q]; i = [0]
    def alt():
        i[0] ^= 1
        return commit(s[:nn], t[i[0]])
    ms = timeit(alt, reps=4); res[f"g{grp}_alternating_bases_ms"] = ms
    res[f"g{grp}_n"] = nn
print(json.dumps(res))
