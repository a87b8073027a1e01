"""BLS12-381 hash-to-curve of the oracle (isogeny constants derived by tools/derive_bls12381_isogenies.py) pinned
by the reference's drand fixtures: together with the pairing check they must reproduce the pass / fail results
of pairing/bls12381/kilic/suite_test.go:17-72, gnark/suite_test.go:16-40 and bls12381_test.go:877-904."""
import hashlib
import json
import os
import struct

import pytest

from oracle import bls12381 as O
from oracle import bls12381_h2c_consts as K


@pytest.fixture(scope="module")
def D(golden_dir):
    return json.load(open(os.path.join(golden_dir, "bls12381_drand.json")))


def test_rfc9380_k_1_0_constant():
    # the first isogeny coefficient of RFC 9380 appendix E.2, as the derivation reproduces it
    assert K.G1_XNUM[0] == 0x11A05F2B1E833340B809101DD99815856B303E88A2D7005FF2627B56CDB4E2C85610C2D5F2E62D6EAEAC1662734649B7
    assert [len(K.G1_XNUM), len(K.G1_XDEN), len(K.G1_YNUM), len(K.G1_YDEN)] == [12, 11, 16, 16]
    assert [len(K.G2_XNUM), len(K.G2_XDEN), len(K.G2_YNUM), len(K.G2_YDEN)] == [4, 3, 4, 4]


def test_sig_on_g1_passes_with_g2_domain_fails_with_g1_domain(D):
    f = D["sig_on_g1"]
    pk, sig = O.g2_decompress(bytes.fromhex(f["pk_g2"])), O.g1_decompress(bytes.fromhex(f["sig_g1"]))
    msg = hashlib.sha256(struct.pack(">Q", f["round"])).digest()
    h_bad = O.hash_to_g1(msg, D["dst_g1"].encode())
    h_ok = O.hash_to_g1(msg, D["dst_g2"].encode())
    assert O.g1_in_subgroup(h_ok) and O.g1_in_subgroup(h_bad)
    assert not O.pair_check(h_bad, pk, sig, O.G2_GEN)  # suite_test.go:36-38
    assert O.pair_check(h_ok, pk, sig, O.G2_GEN)  # suite_test.go:41-45, :84-106


def test_sig_on_g2(D):
    f = D["sig_on_g2"]
    pk, sig = O.g1_decompress(bytes.fromhex(f["pk_g1"])), O.g2_decompress(bytes.fromhex(f["sig_g2"]))
    msg = hashlib.sha256(bytes.fromhex(f["prev_sig"]) + struct.pack(">Q", f["round"])).digest()
    h = O.hash_to_g2(msg, D["dst_g2"].encode())
    assert O.g2_in_subgroup(h)
    assert O.pair_check(O.G1_GEN, sig, pk, h)  # ValidatePairing(base, sigP, pubkeyP, MsgP), suite_test.go:68-71


def test_signature_edge_case(D):
    f = D["edge_case"]
    pk, sig = O.g2_decompress(bytes.fromhex(f["pk_g2"])), O.g1_decompress(bytes.fromhex(f["sig_g1"]))
    h = O.hash_to_g1(bytes.fromhex(f["msg"]), D["dst_g1"].encode())
    assert O.pair_check(h, pk, sig, O.G2_GEN)  # bls.Verify, sign/bls/bls.go:82-96
