#!/usr/bin/env python3
"""Extract the bn256 known answers held by the reference's own tests into tests/golden/bn256.json.
Run in the build container only (needs /root/reference).

Sources:
  sign/bdn/bdn_vartime_test.go:24-48    TestBDN_HashPointToR_BN256: coefficients of P, 2P, 3P (G2 base)
                                        and the aggregated public key  sum (c_i + 1) * P_i
                                        (mask.go:57-61 publicTerms, bdn.go:166-181)
  sign/bdn/bdn_vartime_test.go:90-135   TestBDNFixtures: 3 private scalars -> 3 G2 public keys,
                                        3 G1 signatures on "Hello many times Boneh-Lynn-Shacham"
                                        (sig = x * Hash(msg), sign/bls/bls.go:67-80)
                                        + aggregated signature / key for mask {0, 2} (needs the BLAKE2Xs
                                        coefficients of bdn.go:29-63)
  pairing/bn256/point_test.go:13-45     two pointG1.Hash outputs
  pairing/bn256/hash_test.go:11-20,45-57  TestKnownHashes: HashG1([]byte{i}, nil) for i = 0..10 (the Shallue-van de
                                        Woestijne map of hash.go:10-110 over the HKDF of gfp.go:46-68)
  pairing/bn256/constants.go:104-108    s = sqrt(-3) and (s - 1) / 2, de-Montgomerised here (which of the two roots the
                                        reference chose is data, not derivable)
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bn256.json")

bdn = open(os.path.join(REF, "sign/bdn/bdn_vartime_test.go")).read()
pt = open(os.path.join(REF, "pairing/bn256/point_test.go")).read()

coefs = re.findall(r'require\.Equal\(t, "([0-9a-f]{32})", coefs\[\d\]\.String\(\)\)', bdn)
agg = re.search(r'ref := "([0-9a-f]{256})"', bdn).group(1)
pubs = re.findall(r'public\d := unmarshalHex\(t, suite\.G2\(\)\.Point\(\), "([0-9a-f]{256})"\)', bdn)
privs = re.findall(r'private\d := unmarshalHex\(t, suite\.G2\(\)\.Scalar\(\), "([0-9a-f]{64})"\)', bdn)
sigs = re.findall(r'sig\dExp, err := hex\.DecodeString\("([0-9a-f]{128})"\)', bdn)
agg_sig = re.search(r'aggSigExp := unmarshalHex\(t, suite\.G1\(\)\.Point\(\), "([0-9a-f]{128})"\)', bdn).group(1)
agg_key = re.search(r'aggKeyExp := unmarshalHex\(t, suite\.G2\(\)\.Point\(\), "([0-9a-f]{256})"\)', bdn).group(1)
msg = re.search(r'msg := \[\]byte\("([^"]+)"\)', bdn).group(1)
assert len(coefs) == 3 and len(pubs) == 3 and len(privs) == 3 and len(sigs) == 3
hashes = []
m1 = re.search(r'Hash\(\[\]byte\("abc"\)\).*?DecodeString\("([0-9a-f]{128})"\)', pt, re.S)
hashes.append({"msg_hex": b"abc".hex(), "point": m1.group(1)})
m2 = re.search(r'buf2, err := hex\.DecodeString\("([0-9a-f]{64})"\).*?refBuf2, err := hex\.DecodeString\("([0-9a-f]{128})"\)', pt, re.S)
hashes.append({"msg_hex": m2.group(1), "point": m2.group(2)})
ht = open(os.path.join(REF, "pairing/bn256/hash_test.go")).read()
rows = re.findall(r"\[64\]byte\{([0-9, ]+)\}", ht[ht.index("var marshaledHashes"):])
hash_svdw = [bytes(int(x) for x in r.split(",")).hex() for r in rows]
assert len(hash_svdw) == 11 and all(len(h) == 128 for h in hash_svdw)
cg = open(os.path.join(REF, "pairing/bn256/constants.go")).read()
_U = 6518589491078791937
_P = 36 * _U**4 + 36 * _U**3 + 24 * _U**2 + 6 * _U + 1


def _demont(name):
    w = re.search(r"var %s = &gfP\{(0x[0-9a-f]+), (0x[0-9a-f]+), (0x[0-9a-f]+), (0x[0-9a-f]+)\}" % name, cg).groups()
    v = sum(int(x, 16) << (64 * i) for i, x in enumerate(w))
    return v * pow(1 << 256, -1, _P) % _P


svdw_s, svdw_h = _demont("s"), _demont("sMinus1Over2")
assert svdw_s * svdw_s % _P == _P - 3 and (2 * svdw_h + 1) % _P == svdw_s
json.dump({"hash_g1_svdw": [{"msg_hex": bytes([i]).hex(), "dst_hex": "", "point": h} for i, h in enumerate(hash_svdw)],
           "svdw_s": "%064x" % svdw_s, "svdw_s_minus_1_over_2": "%064x" % svdw_h,
           "bdn_coefs": coefs, "bdn_agg_key": agg, "bdn_pubs": pubs, "bdn_privs": privs, "bdn_sigs": sigs,
           "bdn_msg": msg, "bdn_fixture_agg_sig_mask101": agg_sig, "bdn_fixture_agg_key_mask101": agg_key, "hash_g1": hashes}, open(OUT, "w"), indent=1)
print("ok", OUT)
