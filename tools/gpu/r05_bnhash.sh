#!/bin/bash
# Round 5: bn256 pointG1.Hash with the pending candidates queued per wave (default for n >= 2^17) against one message per lane
# (KYB_BN_HASH_QUEUE=0), same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_bnhash; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bn256.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for rep in 1 2; do for qon in 1 0; do
KYB_BN_HASH_QUEUE=$qon timeout 300 python - <<P | tee -a $O/ab.jsonl
import json, numpy as np, torch, bench
from kyber_amd.pairing import bn256 as m
res = {"queue": $qon}
for lg in (17, 18, 20):
    n = 1 << lg
    msgs = torch.from_numpy(bench.shake(b"bnh", n * 32).reshape(n, 32).copy()).cuda()
    res["hash_g1_2p%d_ms" % lg] = bench.timed(lambda: m.batch_hash_g1(msgs))
print(json.dumps(res))
P
done; done
