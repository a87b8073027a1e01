#!/usr/bin/env python3
"""Extract the BLS12-381 signature fixtures of the reference's tests into tests/golden/bls12381_drand.json
(needs /root/reference; run in the build container).

  pairing/bls12381/kilic/suite_test.go:17-46   TestVerifySigOnG1WithG2Domain  sig on G1, key on G2, msg = SHA-256(BE64(round));
                                               must FAIL with the G1 DST and PASS with the G2 DST used for G1 hashing
  pairing/bls12381/kilic/suite_test.go:48-72   TestVerifySigOnG2 (same vector in gnark/suite_test.go:16-40): sig on G2,
                                               key on G1, msg = SHA-256(prevSig || BE64(round)), G2 DST
  pairing/bls12381/bls12381_test.go:877-904    TestSignatureEdgeCase: bls.Verify (sig on G1, default G1 DST) must pass
"""
import json
import os
import re

REF = "/root/reference/pairing/bls12381"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bls12381_drand.json")
k = open(os.path.join(REF, "kilic/suite_test.go")).read()
t = open(os.path.join(REF, "bls12381_test.go")).read()
g1dst = re.search(r'var domainG1 = \[\]byte\("([^"]+)"\)', open(os.path.join(REF, "kilic/g1.go")).read()).group(1)
g2dst = re.search(r'var domainG2 = \[\]byte\("([^"]+)"\)', open(os.path.join(REF, "kilic/g2.go")).read()).group(1)
f1 = re.search(r'func TestVerifySigOnG1WithG2Domain.*?pk := "([0-9a-f]+)".*?sig := "([0-9a-f]+)".*?round := uint64\((\d+)\)', k, re.S)
f2 = re.search(r'func TestVerifySigOnG2\(.*?pk := "([0-9a-f]+)".*?sig := "([0-9a-f]+)".*?prevSig := "([0-9a-f]+)".*?round := uint64\((\d+)\)', k, re.S)
gn = open(os.path.join(REF, "gnark/suite_test.go")).read()
assert f2.group(2) in gn and f2.group(1) in gn  # the gnark test replays the same vector


def gobytes(name):
    body = re.search(name + r" := \[\]byte\{([^}]*)\}", t).group(1)
    return bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]+)", body)).hex()


edge = re.search(r"func TestSignatureEdgeCase.*", t, re.S).group(0)
t = edge
json.dump({
    "dst_g1": g1dst, "dst_g2": g2dst,
    "sig_on_g1": {"pk_g2": f1.group(1), "sig_g1": f1.group(2), "round": int(f1.group(3))},
    "sig_on_g2": {"pk_g1": f2.group(1), "sig_g2": f2.group(2), "prev_sig": f2.group(3), "round": int(f2.group(4))},
    "edge_case": {"pk_g2": gobytes("publicBytes"), "msg": gobytes("message"), "sig_g1": gobytes("sig")},
}, open(OUT, "w"), indent=1)
print("ok", OUT)
