"""Throughput of the batch UnmarshalBinary kernels on device-resident inputs (points/s), one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from kyber_amd.group import edwards25519 as ed
from kyber_amd.pairing import bls12381 as bls, bn256 as bn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rng = np.random.default_rng(5)
ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
ks[:, 0] &= 0x0F
res = {"n": n}


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


kd = torch.from_numpy(ks).cuda()
pe = ed.batch_mul_base(kd)
res["ed25519"] = n / timed(lambda: ed.batch_unmarshal(pe))
for name, m in (("bls12381", bls), ("bn256", bn)):
    b1 = torch.frombuffer(bytearray(m.G1_BASE), dtype=torch.uint8).cuda()
    b2 = torch.frombuffer(bytearray(m.G2_BASE), dtype=torch.uint8).cuda()
    g1, _ = m.g1_commit(kd, b1)
    g2, _ = m.g2_commit(kd[: n // 4], b2)
    res[f"{name}_g1"] = n / timed(lambda: m.g1_batch_unmarshal(g1))
    res[f"{name}_g2"] = (n // 4) / timed(lambda: m.g2_batch_unmarshal(g2))
    if name == "bls12381":
        res["bls12381_g1_to_affine"] = n / timed(lambda: m.g1_batch_unmarshal(g1, m.F_UNCOMPRESSED_OUT))
        a1, _ = m.g1_batch_unmarshal(g1, m.F_UNCOMPRESSED_OUT)
        res["bls12381_g1_affine_trusted"] = n / timed(lambda: m.g1_batch_unmarshal(a1, m.F_UNCOMPRESSED | m.F_TRUSTED(0)))
print(json.dumps({k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in res.items()}))
