#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_bnlazy_dbg; mkdir -p $O
cat > $O/dbg.py <<'P'
import sys, os, json, hashlib, numpy as np
sys.path.insert(0, os.getcwd())
from kyber_amd.pairing import bn256 as m
n = 4096
def sc(label, n):
    a = np.frombuffer(hashlib.shake_256(label).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); return a
k = sc(b"k", n); h = sc(b"h", n); h[:, 0] &= 0x3f
P = np.asarray(m.g1_commit(h)[0]); Q = np.asarray(m.g2_commit(h)[0])
o1, s1 = m.g1_batch_mul(k, P); o2, s2 = m.g2_batch_mul(k, Q); o3, s3 = m.g2_batch_mul(k, Q, m.F_TRUSTED(0))
G = json.load(open("tests/golden/bn256.json"))
privs = b"".join(bytes.fromhex(p) for p in G["bdn_privs"])
from oracle import bn256 as O
Hm = O.g1_marshal(O.hash_to_g1(G["bdn_msg"].encode()))
sg, _ = m.g1_commit(privs, Hm)
sg2, _ = m.g1_batch_mul(privs, Hm * 3)
np.savez(sys.argv[1], o1=np.asarray(o1), o2=np.asarray(o2), o3=np.asarray(o3), sg=np.asarray(sg), sg2=np.asarray(sg2))
print("fixture commit ok", [bytes(sg[i]).hex() == G["bdn_sigs"][i] for i in range(3)], "batch_mul ok", [bytes(sg2[i]).hex() == G["bdn_sigs"][i] for i in range(3)])
P
python $O/dbg.py $O/lazy.npz
KYBER_HIP_LIB=kyber_amd/lib/libkyberhip_bnpacked.so python $O/dbg.py $O/packed.npz
python - <<P
import numpy as np
a, b = np.load("$O/lazy.npz"), np.load("$O/packed.npz")
for k in ("o1", "o2", "o3"):
    d = np.where((a[k] != b[k]).any(axis=1))[0]
    print(k, "differing lanes", len(d), d[:20])
P
