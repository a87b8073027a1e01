#!/usr/bin/env python3
"""Collect the BLS12-381 fixtures the reference's own tests replay into tests/golden/bls12381_zcash.json.
Run in the build container only (needs /root/reference).

Source: pairing/bls12381/deserialization_tests/{G1,G2}/*.yaml, replayed by
TestZKCryptoVectorsG1Compressed / G2Compressed (pairing/bls12381/bls12381_test.go:74-186):
`output: true` => UnmarshalBinary must succeed, `output: null` => it must fail.
"""
import glob
import json
import os
import re

REF = "/root/reference/pairing/bls12381/deserialization_tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bls12381_zcash.json")

res = {"G1": [], "G2": []}
for grp, key in (("G1", "pubkey"), ("G2", "signature")):
    for path in sorted(glob.glob(os.path.join(REF, grp, "*.yaml"))):
        txt = open(path).read()
        m = re.search(key + r":\s*'?\"?([0-9a-fA-Fx]*)", txt)
        hexstr = m.group(1)
        valid = re.search(r"output:\s*true", txt) is not None
        res[grp].append({"name": os.path.basename(path)[:-5], "hex": hexstr, "valid": valid})
json.dump(res, open(OUT, "w"), indent=1)
print({k: len(v) for k, v in res.items()})
