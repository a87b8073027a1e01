"""Pin the bn254 oracle (oracle/bn254.py) against the known answers and constants the reference holds
(tests/golden/bn254.json, extracted by tests/golden/make_golden_bn254.py) and against an independent
textbook pairing."""
import json
import os
import random

import pytest

from oracle import bn254 as O


@pytest.fixture(scope="module")
def G(golden_dir):
    return json.load(open(os.path.join(golden_dir, "bn254.json")))


def test_constants_are_the_references(G):
    assert [int(v) for v in G["curve_gen"]] == list(O.G1_GEN)                        # curve.go:19-23
    xx, xy, yx, yy = (int(v) for v in G["twist_gen"])                                 # gfP2{x, y} = x i + y
    assert O.G2_GEN == ((xy, xx), (yy, yx))                                           # twist.go:21-33
    assert O.TWIST_B == (int(G["twist_b"][1]), int(G["twist_b"][0]))                  # twist.go:16-19
    assert (O.SVDW_C1, O.SVDW_C2, O.SVDW_C3, O.SVDW_C4) == tuple(int(G["svdw"][k]) for k in ("c1", "c2", "c3", "c4"))
    assert O.g1_on_curve(O.G1_GEN) and O.g2_on_curve(O.G2_GEN)
    assert O.g1_mul(O.ORDER, O.G1_GEN) is None and O.g2_mul(O.ORDER, O.G2_GEN) is None
    assert sum(d << i for i, d in enumerate(O.SIXU_PLUS_2_NAF)) == 6 * O.U + 2        # optate.go:117-120


def test_keccak_and_expand_message(G):
    assert O.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    e = G["expand"]                                                                    # point_test.go:50-79
    assert O.expand_message_xmd(e["dst"].encode(), bytes.fromhex(e["msg_hex"]), 96).hex() == e["out"]  # (two absorbed blocks: Z_pad alone is one rate)


def test_hash_to_field_vectors(G):
    dst = G["h2f_dst"].encode()                                                        # point_test.go:81-101
    for v in G["hash_to_field"]:
        x, y = O.hash_to_field(dst, bytes.fromhex(v["msg"]))
        assert (x, y) == (int(v["x"], 16), int(v["y"], 16))


def test_map_to_point_vectors(G):
    for u, x, y in G["map_to_point"]:                                                  # point_test.go:103-118, 1000 vectors
        assert O.map_to_point(int(u)) == (int(x), int(y))
    # u = 0 and the exceptional inputs of the map still land on the curve
    for u in (0, 1, O.P - 1):
        assert O.g1_on_curve(O.map_to_point(u))


def test_hash_to_g1_fixtures(G):
    dst = G["hash_g1_dst"].encode()                                                    # point_test.go:14-48
    for h in G["hash_g1"]:
        assert O.g1_marshal(O.hash_to_g1(bytes.fromhex(h["msg_hex"]), dst)).hex() == h["point"]


def test_wire_formats_reject_what_the_reference_rejects():
    assert O.g1_unmarshal(bytes(64)) is None and O.g2_unmarshal(bytes(128)) is None
    x, y = O.G1_GEN
    be = lambda v: v.to_bytes(32, "big")
    assert O.g1_unmarshal(be(x) + be(y)) == O.G1_GEN
    for bad in (be(x + O.P) + be(y), be(O.P) + be(O.P), be(5) + be(5)):               # gfp.go:101-118; off the curve
        with pytest.raises(O.DecodeError):
            O.g1_unmarshal(bad)
    g2 = O.g2_marshal(O.G2_GEN)
    assert O.g2_unmarshal(g2) == O.G2_GEN
    with pytest.raises(O.DecodeError):
        O.g2_unmarshal(be(O.P) + g2[32:])
    # a twist point outside the order-n subgroup is rejected (twist.go:62-65); bn256 would accept it
    rng = random.Random(12)
    while True:
        xx = (rng.randrange(O.P), rng.randrange(O.P))
        yy = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(xx), xx), O.TWIST_B))
        if yy is not None:
            break
    assert O.g2_on_curve((xx, yy)) and not O.g2_in_subgroup((xx, yy))
    with pytest.raises(O.DecodeError):
        O.g2_unmarshal(O.g2_marshal((xx, yy)))
    gt = O.gt_marshal(O.pair(O.G1_GEN, O.G2_GEN))
    assert O.gt_marshal(O.gt_unmarshal(gt)) == gt
    with pytest.raises(O.DecodeError):
        O.gt_unmarshal(be(O.P) + gt[32:])


def test_pairing_restatement_equals_textbook_and_is_bilinear():
    rng = random.Random(9)
    a, b = rng.randrange(1, O.ORDER), rng.randrange(1, O.ORDER)
    Pa, Qb = O.g1_mul(a, O.G1_GEN), O.g2_mul(b, O.G2_GEN)
    e = O.pair(Pa, Qb)
    assert e == O.pair_textbook(Pa, Qb)
    base = O.pair(O.G1_GEN, O.G2_GEN)
    assert e == O.f12_pow(base, a * b % O.ORDER)                                       # suite_test.go:240-251
    assert O.f12_pow(base, O.ORDER) == O.F12_ONE and base != O.F12_ONE
    assert len(O.gt_marshal(e)) == 384
    assert O.validate_pairing(Pa, Qb, O.g1_mul(a * b % O.ORDER, O.G1_GEN), O.G2_GEN)
    assert not O.validate_pairing(Pa, Qb, O.G1_GEN, O.G2_GEN)


def _random_twist_point(rng):
    while True:
        x = (rng.randrange(O.P), rng.randrange(O.P))
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(x), x), O.TWIST_B))
        if y is not None:
            return (x, y)


def test_fast_subgroup_criterion_is_the_references_test():
    """[Order]Q == infinity (twist.go:62-65) against the endomorphism criterion of the device library, on the cases that
    could tell them apart: the group is cyclic of order n * q1 q2 q3 q4; points of every prime order, G2 points with a
    small-order component added, random points."""
    h = 2 * O.P - O.ORDER
    prod = 1
    for q in O.G2_COFACTOR_PRIMES:
        prod *= q
        assert O.ORDER % q != 0
    assert prod == h
    # psi satisfies X^2 - t X + p on the twist; on the prime-order components the criterion's polynomial is non-zero
    t = 6 * O.U * O.U + 1
    for q in O.G2_COFACTOR_PRIMES[:3]:
        roots = [m for m in range(q) if (m * m - t * m + O.P) % q == 0] if q < 20000 else None
        if roots is not None:
            assert roots and all(((O.U + 1) + O.U * m + O.U * m * m - 2 * O.U * m**3) % q != 0 for m in roots)
    rng = random.Random(21)
    R = _random_twist_point(rng)
    assert O.g2_mul(O.ORDER * h, R) is None                                            # the group order
    assert not O.g2_in_subgroup(R) and not O.g2_in_subgroup_fast(R)
    g = O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN)
    assert O.g2_in_subgroup(g) and O.g2_in_subgroup_fast(g) and O.g2_in_subgroup_fast(O.G2_GEN)
    assert O.g2_psi(g) == O.g2_mul(O.P % O.ORDER, g)                                   # psi = [p] on G2
    for q in O.G2_COFACTOR_PRIMES:
        small = O.g2_mul(O.ORDER * h // q, R)                                          # a point of order q (or infinity)
        if small is None:
            continue
        assert O.g2_mul(q, small) is None
        assert not O.g2_in_subgroup(small) and not O.g2_in_subgroup_fast(small)
        mixed = O.g2_add(g, small)                                                     # G2 point + small-order component
        assert not O.g2_in_subgroup(mixed) and not O.g2_in_subgroup_fast(mixed)
