#!/bin/bash
# BLS12-381 hash-to-curve kernels (kyb_bls12381_hash_g1 / _g2: G1Elt.Hash / G2Elt.Hash, kilic/g1.go:161-170) on a two-wave
# register budget (libkyberhip_h2cw2.so) against the shipped ones (295 / 512 registers): same box
cd /root/repo; mkdir -p gpurun_out/r04_h2cw; O=gpurun_out/r04_h2cw
for lib in "" libkyberhip_h2cw2.so "" libkyberhip_h2cw2.so; do
  KYBER_HIP_LIB=${lib:+/root/repo/kyber_amd/lib/$lib} timeout 300 python - <<PY | tee -a $O/ab.jsonl
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
out = {"lib": "${lib:-shipped}"}
for n in (1 << 16, 1 << 18):
    m = torch.from_numpy(rng.integers(0, 256, size=(n, 32), dtype=np.uint8)).cuda()
    out["hash_g1_%d_ms" % n] = round(timed(lambda: B.batch_hash_g1(m)), 3)
    out["hash_g2_%d_ms" % n] = round(timed(lambda: B.batch_hash_g2(m)), 3)
print(json.dumps(out))
PY
done
