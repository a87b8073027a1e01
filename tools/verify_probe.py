"""Times the fused BLS12-381 verification (kyb_bls12381_verify_g1: hash to G1, both unmarshal checks, product of two
Miller loops, final exponentiation) on n valid (key, message, signature) triples; one JSON line.  KYBER_HIP_LIB selects
the library build (same-box A/B)."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kyber_amd.pairing import bls12381 as m

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16


def shake(label, nbytes):
    return np.frombuffer(hashlib.shake_256(label).digest(nbytes), dtype=np.uint8)


k = shake(b"verify-probe/k", n * 32).reshape(n, 32).copy()
k[:, 0] &= 0x3F
k = torch.from_numpy(k).cuda()
msgs = torch.from_numpy(shake(b"verify-probe/m", n * 32).reshape(n, 32).copy()).cuda()
g2b = torch.from_numpy(np.frombuffer(m.G2_BASE, dtype=np.uint8).copy()).cuda()
Q, _ = m._mul(2, k, g2b, True)
Hm, _ = m.batch_hash_g1(msgs)
sig, _ = m.g1_batch_mul(k, Hm)
bad = sig.clone()
bad[1::2] = sig[0:-1:2]  # every other signature belongs to the neighbouring key


def t(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


ok, st = m.batch_verify_g1(Q, msgs, sig)
okb, _ = m.batch_verify_g1(Q, msgs, bad)
res = {"n": n, "all_valid_accepted": bool(ok.all().item()) and not bool(st.any().item()),
       "wrong_signatures_rejected": int((~okb.bool()).sum().item()), "wrong_signatures": n // 2,
       "verify_ms": t(lambda: m.batch_verify_g1(Q, msgs, sig)),
       "verify_known_keys_ms": t(lambda: m.batch_verify_g1(Q, msgs, sig, flags=m.F_TRUSTED(0)))}
# one public key for every message (a drand chain): program VERIFYK, both Miller loops from line tables
x1 = k[:1].contiguous()
X1 = m.g2_commit(x1)[0][0].contiguous()
sig1, _ = m.g1_batch_mul(x1.repeat(n, 1), Hm)
ok1, st1 = m.batch_verify_g1_same_key(X1, msgs, sig1)
okg, _ = m.batch_verify_g1(X1.repeat(n, 1), msgs, sig1)
X2 = m.g2_commit(k[1:2].contiguous())[0][0].contiguous()
alt = [X1, X2]
i = [0]
def alternating():  # every call changes the key: every call pays the key's table
    i[0] ^= 1
    return m.batch_verify_g1_same_key(alt[i[0]], msgs, sig1)
res["same_key_all_valid_accepted"] = bool(ok1.all().item()) and not bool(st1.any().item()) and bool(okg.all().item())
res["same_key_ms"] = t(lambda: m.batch_verify_g1_same_key(X1, msgs, sig1))
res["same_key_known_key_ms"] = t(lambda: m.batch_verify_g1_same_key(X1, msgs, sig1, flags=m.F_TRUSTED(0)))
res["same_key_general_program_ms"] = t(lambda: m.batch_verify_g1(X1.repeat(n, 1), msgs, sig1))
res["same_key_alternating_keys_ms"] = t(alternating)
res["same_key_per_s"] = n / res["same_key_ms"] * 1e3
res["verify_per_s"] = n / res["verify_ms"] * 1e3
res["verify_known_keys_per_s"] = n / res["verify_known_keys_ms"] * 1e3
print(json.dumps(res))
