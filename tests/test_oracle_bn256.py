"""Pin the bn256 oracle (oracle/bn256.py) against the known answers the reference's tests hold
(tests/golden/bn256.json, extracted by tests/golden/make_golden_bn256.py) and against an independent
textbook pairing."""
import json
import os
import random

import pytest

from oracle import bn256 as O


@pytest.fixture(scope="module")
def G(golden_dir):
    return json.load(open(os.path.join(golden_dir, "bn256.json")))


def test_generators():
    assert O.g1_on_curve(O.G1_GEN) and O.g2_on_curve(O.G2_GEN)
    assert O.g1_mul(O.ORDER, O.G1_GEN) is None and O.g2_mul(O.ORDER, O.G2_GEN) is None


def test_hash_to_g1_fixtures(G):
    for h in G["hash_g1"]:
        assert O.g1_marshal(O.hash_to_g1(bytes.fromhex(h["msg_hex"]))).hex() == h["point"]


def test_bdn_public_keys_and_signatures(G):
    Hm = O.hash_to_g1(G["bdn_msg"].encode())
    for priv, pub, sig in zip(G["bdn_privs"], G["bdn_pubs"], G["bdn_sigs"]):
        x = int(priv, 16)
        assert O.g2_marshal(O.g2_mul(x, O.G2_GEN)).hex() == pub  # G2 fixed-base mul + wire format
        assert O.g1_marshal(O.g1_mul(x, Hm)).hex() == sig  # G1 variable-base mul
        assert O.g2_unmarshal(bytes.fromhex(pub)) == O.g2_mul(x, O.G2_GEN)
        # and the signature verifies: e(H, X) == e(sig, G2)   (sign/bls/bls.go:36-38)
    x = int(G["bdn_privs"][0], 16)
    assert O.validate_pairing(Hm, O.g2_unmarshal(bytes.fromhex(G["bdn_pubs"][0])),
                              O.g1_unmarshal(bytes.fromhex(G["bdn_sigs"][0])), O.G2_GEN)


def test_bdn_aggregated_key(G):
    # sum (c_i + 1) * (i + 1) * G2   (mask.go:57-61, bdn.go:166-181)
    acc = None
    for i, c in enumerate(G["bdn_coefs"]):
        Pi = O.g2_mul(i + 1, O.G2_GEN)
        acc = O.g2_add(acc, O.g2_add(O.g2_mul(int(c, 16), Pi), Pi))
    assert O.g2_marshal(acc).hex() == G["bdn_agg_key"]


def test_wire_edge_cases():
    assert O.g1_unmarshal(bytes(64)) is None and O.g2_unmarshal(bytes(128)) is None
    assert O.g1_marshal(None) == bytes(64)
    # coordinates >= p are reduced, not rejected (montEncode, point.go:218-221)
    x, y = O.G1_GEN
    assert O.g1_unmarshal((x + O.P).to_bytes(32, "big") + y.to_bytes(32, "big")) == O.G1_GEN
    assert O.g1_unmarshal(O.P.to_bytes(32, "big") * 2) is None
    with pytest.raises(O.DecodeError):
        O.g1_unmarshal((5).to_bytes(32, "big") + (5).to_bytes(32, "big"))


def test_pairing_restatement_equals_textbook_and_is_bilinear():
    rng = random.Random(9)
    a, b = rng.randrange(1, O.ORDER), rng.randrange(1, O.ORDER)
    Pa, Qb = O.g1_mul(a, O.G1_GEN), O.g2_mul(b, O.G2_GEN)
    e = O.pair(Pa, Qb)
    assert e == O.pair_textbook(Pa, Qb)
    base = O.pair(O.G1_GEN, O.G2_GEN)
    assert e == O.f12_pow(base, a * b % O.ORDER)
    assert O.f12_pow(base, O.ORDER) == O.F12_ONE and base != O.F12_ONE
    assert len(O.gt_marshal(e)) == 384


def test_pairing_off_subgroup_g2_matches_textbook():
    # UnmarshalBinary never checks the subgroup on G2 (point.go:466-499): such inputs must still be
    # processed deterministically; restatement and textbook agree on them too.
    rng = random.Random(10)
    for _ in range(64):
        x = (rng.randrange(O.P), rng.randrange(O.P))
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(x), x), O.TWIST_B))
        if y is None:
            continue
        q = (x, y)
        assert O.g2_on_curve(q) and O.g2_mul(O.ORDER, q) is not None
        assert O.pair(O.G1_GEN, q) == O.pair_textbook(O.G1_GEN, q)
        return
    raise AssertionError("no twist point found")


def test_hash_g1_svdw_known_hashes(G):
    """TestKnownHashes (pairing/bn256/hash_test.go:11-20, 45-57): HashG1([]byte{i}, nil) for i = 0..10 -- pins hashToBase's
    HKDF (gfp.go:46-68) and mapToCurve (hash.go:14-110) of the restatement, including which root of -3 the reference uses"""
    assert len(G["hash_g1_svdw"]) == 11
    for i, v in enumerate(G["hash_g1_svdw"]):
        assert bytes.fromhex(v["msg_hex"]) == bytes([i]) and v["dst_hex"] == ""
        assert O.g1_marshal(O.hash_g1_svdw(bytes([i]), b"")).hex() == v["point"]
    # the other root of -3 swaps x1 and x2, which changes the point whenever both are abscissae of the curve: the
    # reference's choice (constants.go:105) is data, and the vectors fix it
    s0, h0 = O.SVDW_S, O.SVDW_S_MINUS_1_OVER_2
    try:
        O.SVDW_S = O.P - s0
        O.SVDW_S_MINUS_1_OVER_2 = (O.SVDW_S - 1) * pow(2, -1, O.P) % O.P
        other = [O.g1_marshal(O.hash_g1_svdw(bytes([i]))).hex() for i in range(11)]
    finally:
        O.SVDW_S, O.SVDW_S_MINUS_1_OVER_2 = s0, h0
    diff = sum(a != v["point"] for a, v in zip(other, G["hash_g1_svdw"]))
    assert 0 < diff < 11
