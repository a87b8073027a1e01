#!/bin/bash
# Ed25519: the kernels without a budget of their own on two waves (shipped: ed25519.hip KYB_TU_WAVES 2) against the same
# unit compiled loose (libkyberhip_edw1.so) -- UnmarshalBinary, Add, Hash, the MSM (its one-lane reduce kernel is among them)
cd /root/repo; mkdir -p gpurun_out/r04_edaux; O=gpurun_out/r04_edaux
for lib in "" libkyberhip_edw1.so "" libkyberhip_edw1.so; do
  KYBER_HIP_LIB=${lib:+/root/repo/kyber_amd/lib/$lib} timeout 300 python - <<PY | tee -a $O/ab.jsonl
import json, time, numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from kyber_amd.group import edwards25519 as ed
n = 1 << 20
rng = np.random.default_rng(5)
ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); ks[:, 31] &= 0x0F
P = ed.batch_mul_base(torch.from_numpy(ks).cuda()); Q = ed.batch_mul_base(torch.from_numpy(ks[::-1].copy()).cuda())
msgs = torch.from_numpy(rng.integers(0, 256, size=(n, 32), dtype=np.uint8)).cuda()
def timed(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
S = torch.from_numpy(ks).cuda()
msm_ms = round(timed(lambda: ed.msm(S, P)), 3)
print(json.dumps({"lib": "${lib:-shipped}", "msm_2p20_ms": msm_ms, "unmarshal_ms": round(timed(lambda: ed.batch_unmarshal(P)), 3), "add_ms": round(timed(lambda: ed.batch_add(P, Q)), 3), "hash_ms": round(timed(lambda: ed.batch_hash(msgs, b"r04-edaux")), 3)}))
PY
done
timeout 900 python -m pytest tests/test_gpu_ed25519.py tests/test_gpu_msm.py tests/test_gpu_callers.py -q -x 2>&1 | tail -2
