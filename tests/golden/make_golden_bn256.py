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
json.dump({"bdn_coefs": coefs, "bdn_agg_key": agg, "bdn_pubs": pubs, "bdn_privs": privs, "bdn_sigs": sigs,
           "bdn_msg": msg, "bdn_fixture_agg_sig_mask101": agg_sig, "bdn_fixture_agg_key_mask101": agg_key, "hash_g1": hashes}, open(OUT, "w"), indent=1)
print("ok", OUT)
