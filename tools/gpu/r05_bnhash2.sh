#!/bin/bash
# messages per wave of the queued bn256 hash (KYB_BN_HASH_HQ) at 2^17 .. 2^20
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_bnhash; mkdir -p $O
for hq in 128 256 512; do
KYB_BN_HASH_HQ=$hq timeout 300 python - <<P | tee -a $O/hq.jsonl
import json, numpy as np, torch, bench
from kyber_amd.pairing import bn256 as m
res = {"hq": $hq}
for lg in (17, 18, 19, 20):
    n = 1 << lg
    msgs = torch.from_numpy(bench.shake(b"bnh", n * 32).reshape(n, 32).copy()).cuda()
    res["2p%d_ms" % lg] = round(bench.timed(lambda: m.batch_hash_g1(msgs)), 3)
print(json.dumps(res))
P
done
