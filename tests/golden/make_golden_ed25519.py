#!/usr/bin/env python3
"""Extract the Ed25519 golden vectors held by the reference's own tests into
small fixtures under tests/golden/.  Run in the build container only (needs
/root/reference); the fixtures it writes are committed and are what travels to
the GPU box.

Sources (all under /root/reference):
  sign/eddsa/eddsa_test.go:24-52        RFC 8032 section 7.1 vectors
  sign/eddsa/testdata/sign.input.gz     1024 SUPERCOP KATs (TestGolden, eddsa_test.go:285)
  sign/eddsa/testdata/ed25519_test.json 150 Wycheproof cases (TestWycheProof, eddsa_test.go:355)
  group/edwards25519/point_test.go:369-445  RFC 9380 hash-to-field / hash-to-curve values
  group/edwards25519/const.go:1453-1473 small-order encodings (weakKeys)

Derived columns (a, r, h) are computed here with hashlib exactly as
sign/eddsa/eddsa.go:45-59,91-142,207-227 derive them, so the fixture holds pure
(scalar, point) -> point known answers for the hot path:
  pub = a*B   (fixed-base)         R = r*B   (fixed-base)
  S*B = R + h*A                    (variable-base, through the verify equation)
"""
import gzip
import hashlib
import json
import os
import re

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
L = 2**252 + 27742317777372353535851937790883648493


def clamp(h):
    b = bytearray(h[:32])
    b[0] &= 0xF8
    b[31] &= 0x7F
    b[31] |= 0x40
    return bytes(b)


def kat_row(seed, pub, msg, sig):
    dig = hashlib.sha512(seed).digest()
    a = clamp(dig)
    prefix = dig[32:]
    r = int.from_bytes(hashlib.sha512(prefix + msg).digest(), "little") % L
    R, S = sig[:32], sig[32:]
    h = int.from_bytes(hashlib.sha512(R + pub + msg).digest(), "little") % L
    return [a, pub, r.to_bytes(32, "little"), R, h.to_bytes(32, "little"), S]


def main():
    rows = []
    with gzip.open(f"{REF}/sign/eddsa/testdata/sign.input.gz", "rt") as f:
        for line in f:
            parts = line.strip().split(":")
            sk, pub, msg, sigmsg = (bytes.fromhex(x) for x in parts[:4])
            assert sk[32:] == pub
            rows.append(kat_row(sk[:32], pub, msg, sigmsg[:64]))
    arr = np.frombuffer(b"".join(b"".join(r) for r in rows), dtype=np.uint8)
    arr = arr.reshape(len(rows), 6, 32)
    np.save(f"{OUT}/ed25519_sign_input.npy", arr)

    misc = {}
    src = open(f"{REF}/sign/eddsa/eddsa_test.go").read()
    blk = src[src.index("EdDSATestVectors = "):src.index("// Tests if marshalling")]
    strs = re.findall(r'"([0-9a-f]*)"', blk)
    assert len(strs) == 20
    misc["rfc8032"] = [dict(seed=strs[i], pub=strs[i + 1], msg=strs[i + 2], sig=strs[i + 3])
                       for i in range(0, 20, 4)]

    src = open(f"{REF}/group/edwards25519/point_test.go").read()
    f1 = src[src.index("func TestHashToField"):src.index("func TestHashToPoint")]
    f2 = src[src.index("func TestHashToPoint"):]
    us = re.findall(r'"([0-9a-f]{40,64})"', f1)
    ps = re.findall(r'"([0-9a-f]{40,64})"', f2)[:10]
    assert len(us) == 10 and len(ps) == 10
    # the five RFC 9380 input messages (point_test.go:21-38) and the suite DST (point_test.go:370)
    blk = src[src.index("inputsTestVectRFC9380 = []string{"):]
    blk = blk[:blk.index("\n\t}")]
    msgs = ["".join(re.findall(r'"([^"]*)"', part)) for part in re.split(r",\n\t\t(?=\")", blk[blk.index("{") + 1:])]
    msgs = [m for m in msgs][:5]
    assert [len(m) for m in msgs] == [0, 3, 16, 133, 517], [len(m) for m in msgs]
    dst = re.search(r'func TestHashToPoint.*?dst := "([^"]+)"', src, re.S).group(1)
    misc["rfc9380_dst"] = dst
    misc["rfc9380"] = [dict(msg=msgs[i], u0=us[2 * i], u1=us[2 * i + 1], x=ps[2 * i], y=ps[2 * i + 1])
                       for i in range(5)]

    src = open(f"{REF}/group/edwards25519/const.go").read()
    blk = src[src.index("var weakKeys"):]
    blk = blk[:blk.index("}}") + 2]
    groups = re.findall(r"\{((?:\s*0x[0-9a-f]{2},?)+)\s*\}", blk)
    weak = [bytes(int(x, 16) for x in re.findall(r"0x([0-9a-f]{2})", g)).hex() for g in groups]
    assert len(weak) == 5 and all(len(w) == 64 for w in weak)
    misc["small_order"] = weak

    wp = json.load(open(f"{REF}/sign/eddsa/testdata/ed25519_test.json"))
    cases = []
    for g in wp["testGroups"]:
        for t in g["tests"]:
            cases.append(dict(pk=g["publicKey"]["pk"], msg=t["msg"], sig=t["sig"],
                              valid=(t["result"] == "valid"), id=t["tcId"]))
    assert len(cases) == 150
    misc["wycheproof"] = cases
    # point_test.go:66-96 TestPointIsCanonical: of the 38 encodings p+i (i<19, both
    # sign bits) exactly 24 decode successfully.
    misc["noncanonical_decodable_count"] = 24
    json.dump(misc, open(f"{OUT}/ed25519_misc.json", "w"), indent=0)
    print("wrote", arr.shape, len(cases))


if __name__ == "__main__":
    main()
