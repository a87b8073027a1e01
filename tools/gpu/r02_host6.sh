#!/bin/bash
# Host-buffer entry points of the pairing suites and MSMs at the C ABI / Python mirror (PCIe-inclusive) next to resident
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_host; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_bn256.py tests/test_gpu_bls12381.py tests/test_gpu_devices.py -x -q 2>&1 | tail -3
timeout 600 python - <<'PY' | tee $O/host_pairing.json
import json, time, hashlib, numpy as np, torch
from kyber_amd.pairing import bls12381 as bls, bn256 as bn
def med(fn, k=5):
    fn(); ts = []
    for _ in range(k):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(sorted(ts)[k // 2] * 1e3, 3)
res = {}
for name, m, n in (("bls12381", bls, 1 << 16), ("bn256", bn, 1 << 18)):
    k = np.frombuffer(hashlib.shake_256(b"hp/" + name.encode()).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); k[:, 0] &= 0x3F
    P, _ = m.g1_commit(k); Q, _ = m.g2_commit(k)
    dP, dQ, dk = torch.from_numpy(np.asarray(P)).cuda(), torch.from_numpy(np.asarray(Q)).cuda(), torch.from_numpy(k).cuda()
    res[name] = {"pair_host_ms": med(lambda: m.batch_pair(P, Q)), "pair_resident_ms": med(lambda: m.batch_pair(dP, dQ)),
                 "g1_mul_host_ms": med(lambda: m.g1_batch_mul(k, P)), "g1_mul_resident_ms": med(lambda: m.g1_batch_mul(dk, dP))}
n = 1 << 20
k = np.frombuffer(hashlib.shake_256(b"hp/msm").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); k[:, 0] &= 0x3F
P, _ = bls._mul(1, k, bls.G1_BASE, True, bls.F_UNCOMPRESSED_OUT)
P = np.asarray(P); dP, dk = torch.from_numpy(P).cuda(), torch.from_numpy(k).cuda()
fl = bls.F_TRUSTED(0) | bls.F_UNCOMPRESSED
res["bls12381_g1_msm_2p20"] = {"host_ms": med(lambda: bls.g1_msm(k, P, fl)), "resident_ms": med(lambda: bls.g1_msm(dk, dP, fl))}
print(json.dumps(res))
PY
