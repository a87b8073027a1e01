#!/usr/bin/env python3
"""share.PriPoly.Commit at 2^20 coefficients EXACTLY as bench.py's other_workloads() calls it -- the same SHAKE-256
scalars (label kyberhip/v1/msm/k), the same base 0x1234567 G, device-resident tensors, bench.timed()'s median of 20
HIP-event timings after 5 warm-ups -- so that a rocprofv3 trace / PMC pass of THIS command is the profile of the figure
the bench line quotes (the earlier profile was of tools/bls_probe.py's inputs).
usage: commit_probe.py [n]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from kyber_amd.pairing import bls12381 as m
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ks = torch.from_numpy(bench.be_scalars(b"kyberhip/v1/msm/k", n)).cuda()
cb = torch.from_numpy(np.asarray(m.g1_commit((0x1234567).to_bytes(32, "big"))[0])[0].copy()).cuda()
fn = lambda: m.g1_commit(ks, cb)
ms = bench.timed(fn)
print(json.dumps({"n": n, "g1_commit_ms": ms, "commits_per_s": n / ms * 1e3, "mads_per_commit": bench.MADS_G1_COMMIT,
                  "calls_timed": 20, "calls_warm": 5}))
