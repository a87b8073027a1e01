import hashlib, json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from kyber_amd.pairing import bls12381 as m
def sc(label, n):
    a = np.frombuffer(hashlib.shake_256(label).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); a[:, 0] &= 0x3F
    return a
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for n in (1 << 20, 1 << 16, 3000):
    k = sc(b"k", n); h = torch.from_numpy(sc(b"h", n)).cuda()
    k128 = k.copy(); k128[:, :16] = 0
    dk, dk128 = torch.from_numpy(k).cuda(), torch.from_numpy(k128).cuda()
    g1b = torch.from_numpy(np.frombuffer(m.G1_BASE, dtype=np.uint8).copy()).cuda()
    Pu, _ = m._mul(1, h, g1b, True, m.F_UNCOMPRESSED_OUT)
    fl = m.F_TRUSTED(0) | m.F_UNCOMPRESSED
    print(json.dumps({"n": n, "full_ms": t(lambda: m.g1_msm(dk, Pu, fl)), "k128_noflag_ms": t(lambda: m.g1_msm(dk128, Pu, fl)),
                      "k128_flag_ms": t(lambda: m.g1_msm(dk128, Pu, fl | m.F_SCALAR_BITS(128)))}))
