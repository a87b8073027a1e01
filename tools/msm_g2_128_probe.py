import hashlib, json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from kyber_amd.pairing import bls12381 as m
n = int(sys.argv[1])
def sc(label, n):
    a = np.frombuffer(hashlib.shake_256(label).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); a[:, 0] &= 0x3F
    return a
k = sc(b"k", n); h = torch.from_numpy(sc(b"h", n)).cuda()
k128 = k.copy(); k128[:, :16] = 0
dk, dk128 = torch.from_numpy(k).cuda(), torch.from_numpy(k128).cuda()
g2b = torch.from_numpy(np.frombuffer(m.G2_BASE, dtype=np.uint8).copy()).cuda()
Pu, _ = m._mul(2, h, g2b, True, m.F_UNCOMPRESSED_OUT)
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
fl = m.F_TRUSTED(0) | m.F_UNCOMPRESSED
a = m.g2_msm(dk128, Pu, fl)[0]; b = m.g2_msm(dk128, Pu, fl | m.F_SCALAR_BITS(128))[0]
print(json.dumps({"n": n, "mode": os.environ.get("KYB_BLS_G2_MSM_GLS", "default"), "full_ms": t(lambda: m.g2_msm(dk, Pu, fl)),
                  "k128_flag_ms": t(lambda: m.g2_msm(dk128, Pu, fl | m.F_SCALAR_BITS(128))), "flag_equals_noflag": bool((a == b).all().item())}))
