#!/bin/bash
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log; tail -8 gpurun_out/pytest_gpu.log
python - <<'PY'
import sys, os, time, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from kyber_amd.pairing import bls12381 as bls
n = 1 << 16
msgs = torch.from_numpy(np.frombuffer(hashlib.shake_256(b"m").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()).cuda()
for name, fn in (("hash_g1", bls.batch_hash_g1), ("hash_g2", bls.batch_hash_g2)):
    fn(msgs); torch.cuda.synchronize()
    t = time.perf_counter(); out, st = fn(msgs); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("bls12381 %s: %.3g hashes/s" % (name, n / dt), "ok" if not st.any().item() else "FAIL")
PY
