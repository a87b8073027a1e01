#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04_h2cw; O=gpurun_out/r04_h2cw
timeout 900 python -m pytest tests/test_gpu_switches.py -q -x -k "hashw2" 2>&1 | tail -2
for sw in 1 0 1 0; do
  KYB_UNM_W2=$sw timeout 300 python - <<PY | tee -a $O/shipped_ab.jsonl
import json, numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from kyber_amd.pairing import bls12381 as B
rng = np.random.default_rng(7)
def timed(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
out = {"unm_w2": $sw}
for n in (1 << 17, 1 << 18):
    m = torch.from_numpy(rng.integers(0, 256, size=(n, 32), dtype=np.uint8)).cuda()
    out["hash_g1_%d_ms" % n] = round(timed(lambda: B.batch_hash_g1(m)), 3)
    out["hash_g2_%d_ms" % n] = round(timed(lambda: B.batch_hash_g2(m)), 3)
print(json.dumps(out))
PY
done
