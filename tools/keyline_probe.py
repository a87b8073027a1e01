#!/usr/bin/env python3
"""First sight of a public key by kyb_bls12381_verify_g1_same_key: every call brings a key the stream's cache has never
seen (its 68 Miller lines are walked on the device before the batch is verified), against calls that repeat one key.
usage: keyline_probe.py [n signatures per call] [calls]; KYBER_HIP_LIB selects the build (A/B); one JSON line"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kyber_amd.pairing import bls12381 as bls
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 24
def sc(label, k):
    a = np.frombuffer(hashlib.shake_256(label).digest(k * 32), dtype=np.uint8).reshape(k, 32).copy(); a[:, 0] &= 0x3F
    return a
msgs = torch.from_numpy(sc(b"kl/m", n)).cuda()
Hm, _ = bls.batch_hash_g1(msgs)
xs = torch.from_numpy(sc(b"kl/x", calls + 1)).cuda()
keys = bls.g2_commit(xs)[0]
sigs = [bls.g1_batch_mul(xs[j:j + 1].repeat(n, 1), Hm)[0] for j in range(calls + 1)]
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), r
ok_all = True
t, (ok, st) = timed(lambda: bls.batch_verify_g1_same_key(keys[calls].contiguous(), msgs, sigs[calls]))  # warm-up (workspaces, code)
new, rep = [], []
for j in range(calls):
    t, (ok, st) = timed(lambda: bls.batch_verify_g1_same_key(keys[j].contiguous(), msgs, sigs[j]))
    ok_all &= bool(ok.all().item()) and not bool(st.any().item())
    new.append(t)
    t, (ok, st) = timed(lambda: bls.batch_verify_g1_same_key(keys[j].contiguous(), msgs, sigs[j]))
    rep.append(t)
new.sort(); rep.sort()
print(json.dumps({"n": n, "calls": calls, "new_key_call_ms_median": new[len(new) // 2], "same_key_call_ms_median": rep[len(rep) // 2],
                  "first_sight_cost_ms": new[len(new) // 2] - rep[len(rep) // 2], "all_verified": ok_all, "lib": os.environ.get("KYBER_HIP_LIB", "default")}))
