"""The BN device library instantiated at pairing/bn254's constants (kyber_amd/csrc/bn254.cuh), compiled for the host,
against the bn254 oracle and the reference's own vectors (tests/golden/bn254.json)."""
import json
import os
import random

import pytest

from oracle import bn254 as O
from tests import _host_harness as H


@pytest.fixture(scope="module")
def G(golden_dir):
    return json.load(open(os.path.join(golden_dir, "bn254.json")))


def _be(v):
    return v.to_bytes(32, "big")


def test_keccak_expand_and_hash_vectors(G):
    for n in (0, 1, 55, 135, 136, 137, 272, 500):
        m = bytes((i * 7 + 3) & 0xFF for i in range(n))
        assert H.call("hh_bn4_keccak256", m, n, out_sizes=(32,))[1] == O.keccak256(m)
    dst = G["hash_g1_dst"].encode()
    for h in G["hash_g1"]:                                   # point_test.go:14-48
        m = bytes.fromhex(h["msg_hex"])
        assert H.call("hh_bn4_hash_g1", m, len(m), dst, len(dst), out_sizes=(64,)) == (0, bytes.fromhex(h["point"]))
    d2 = G["h2f_dst"].encode()
    for v in G["hash_to_field"][:12]:                        # hash_to_field feeds the map: compare whole hashes with the oracle
        m = bytes.fromhex(v["msg"])
        assert H.call("hh_bn4_hash_g1", m, len(m), d2, len(d2), out_sizes=(64,)) == (0, O.g1_marshal(O.hash_to_g1(m, d2)))


def test_map_to_point_vectors(G):
    for u, x, y in G["map_to_point"][::5]:                   # 200 of the 1000 reference vectors
        assert H.call("hh_bn4_map_to_point", _be(int(u)), out_sizes=(64,))[1] == _be(int(x)) + _be(int(y))
    for u in (0, 1, O.P - 1):
        assert H.call("hh_bn4_map_to_point", _be(u), out_sizes=(64,))[1] == O.g1_marshal(O.map_to_point(u))


def test_scalar_multiplication_and_strict_decoding():
    rng = random.Random(41)
    g1, g2 = O.g1_marshal(O.G1_GEN), O.g2_marshal(O.G2_GEN)
    for k in [0, 1, 2, O.ORDER - 1, O.ORDER, O.ORDER + 5, (1 << 256) - 1] + [rng.getrandbits(256) for _ in range(6)]:
        kb = k.to_bytes(32, "big")
        assert H.call("hh_bn4_g1_mul", kb, g1, out_sizes=(64,)) == (0, O.g1_marshal(O.g1_mul(k, O.G1_GEN)))
    # G2 multiplication is the 4-dimensional GLS walk (psi = [6u^2] on the subgroup): edge scalars of the split
    lam = 6 * O.U * O.U
    q7 = O.g2_mul(7, O.G2_GEN)
    for k in [0, 1, 2, 15, 16, 17, O.ORDER - 1, O.ORDER, O.ORDER + 1, (1 << 256) - 1, 1 << 255, lam, lam * lam % O.ORDER, lam - 1,
              O.U, 2 * O.U + 1] + [rng.getrandbits(256) for _ in range(6)] + [rng.getrandbits(64) for _ in range(3)]:
        kb = k.to_bytes(32, "big")
        assert H.call("hh_bn4_g2_mul", kb, g2, 0, out_sizes=(128,)) == (0, O.g2_marshal(O.g2_mul(k, O.G2_GEN))), k
        assert H.call("hh_bn4_g2_mul", kb, O.g2_marshal(q7), 0x100, out_sizes=(128,)) == (0, O.g2_marshal(O.g2_mul(k, q7))), k
    p = O.g1_mul(rng.getrandbits(200), O.G1_GEN)
    q = O.g2_mul(rng.getrandbits(200), O.G2_GEN)
    assert H.call("hh_bn4_g1_add", O.g1_marshal(p), g1, out_sizes=(64,)) == (0, O.g1_marshal(O.g1_add(p, O.G1_GEN)))
    assert H.call("hh_bn4_g1_add", O.g1_marshal(p), O.g1_marshal(p), out_sizes=(64,)) == (0, O.g1_marshal(O.g1_add(p, p)))
    assert H.call("hh_bn4_g2_add", O.g2_marshal(q), g2, out_sizes=(128,)) == (0, O.g2_marshal(O.g2_add(q, O.G2_GEN)))
    # infinity, coordinates >= p (gfp.go:101-118), off the curve
    x, y = O.G1_GEN
    assert H.call("hh_bn4_g1_decode", bytes(64))[0] == 0
    assert H.call("hh_bn4_g1_decode", _be(x + O.P) + _be(y))[0] == 1
    assert H.call("hh_bn4_g1_decode", _be(O.P) + _be(O.P))[0] == 1
    assert H.call("hh_bn4_g1_decode", _be(5) + _be(5))[0] == 1
    assert H.call("hh_bn4_g1_mul", _be(7), bytes(64), out_sizes=(64,)) == (0, bytes(64))
    assert H.call("hh_bn4_g2_mul", _be(7), bytes(128), 0, out_sizes=(128,)) == (0, bytes(128))          # GLS walk on infinity
    assert H.call("hh_bn4_g2_mul", _be(O.ORDER), g2, 0x100, out_sizes=(128,)) == (0, bytes(128))
    assert H.call("hh_bn4_g1_mul", _be(7), _be(x + O.P) + _be(y), out_sizes=(64,)) == (1, bytes(64))
    assert H.call("hh_bn4_g2_decode", bytes(128), 1)[0] == 0
    assert H.call("hh_bn4_g2_decode", _be(O.P) + g2[32:], 1)[0] == 1
    # a twist point outside the order-n subgroup: rejected (twist.go:62-65) unless the caller vouches for the operand
    while True:
        xx = (rng.randrange(O.P), rng.randrange(O.P))
        yy = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(xx), xx), O.TWIST_B))
        if yy is not None:
            break
    off = O.g2_marshal((xx, yy))
    assert H.call("hh_bn4_g2_decode", off, 1)[0] == 2 and H.call("hh_bn4_g2_decode", off, 0)[0] == 0
    # points the endomorphism criterion could confuse if it were not exact: prime-order components of the cofactor
    h = 2 * O.P - O.ORDER
    g = O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN)
    assert H.call("hh_bn4_g2_decode", O.g2_marshal(g), 1)[0] == 0
    for q in O.G2_COFACTOR_PRIMES:
        small = O.g2_mul(O.ORDER * h // q, (xx, yy))
        if small is not None:
            assert H.call("hh_bn4_g2_decode", O.g2_marshal(small), 1)[0] == 2, q
            assert H.call("hh_bn4_g2_decode", O.g2_marshal(O.g2_add(g, small)), 1)[0] == 2, q
    assert H.call("hh_bn4_g2_mul", _be(3), off, 0, out_sizes=(128,)) == (2, bytes(128))
    # vouching for a point that is not in the subgroup is a broken precondition: a status-free result, but not the
    # plain multiple (the GLS walk relies on psi = [6u^2])
    assert H.call("hh_bn4_g2_mul", _be(3), off, 0x100, out_sizes=(128,))[0] == 0


def test_fixed_base_table_answers_bn254_g2_membership():
    """fixed_base.cuh round 4: bn254's G2 rule [Order] Q = infinity (twist.go:62-65) read off the finished plain table."""
    import ctypes

    rng = random.Random(8)
    lib = H.lib()

    def verdict(wire):
        m = ctypes.c_int(-1)
        return lib.hh_bn4_g2_fb_member(ctypes.c_char_p(wire), ctypes.byref(m)), m.value

    g = O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN)
    assert verdict(O.g2_marshal(g)) == (0, 1)
    while True:
        xx = (rng.randrange(O.P), rng.randrange(O.P))
        yy = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(xx), xx), O.TWIST_B))
        if yy is not None:
            break
    assert verdict(O.g2_marshal((xx, yy))) == (0, 0)
    h = 2 * O.P - O.ORDER
    small = O.g2_mul(O.ORDER * h // O.G2_COFACTOR_PRIMES[0], (xx, yy))
    if small is not None:
        assert verdict(O.g2_marshal(small)) == (0, 0) and verdict(O.g2_marshal(O.g2_add(g, small))) == (0, 0)


def test_field_inversion_by_division_steps():
    """fp_inv (mont.cuh: Bernstein-Yang division steps, thirty per batch): zero, one, p - 1, powers of two, values whose
    (f, g) walk is long or short, random values -- on the three fields, against pow(a, -1, p)."""
    from oracle import bls12381 as OB, bn256 as ON

    rng = random.Random(99)

    def cases(p, nbytes):
        vals = [0, 1, 2, 3, p - 1, p - 2, (p + 1) // 2, (p - 1) // 2, (1 << 30) - 1, 1 << 30, (1 << 60) + 1, (1 << 255) % p]
        vals += [(1 << k) % p for k in range(0, 8 * nbytes, 29)] + [p - ((1 << k) % p) for k in range(1, 8 * nbytes, 31)]
        vals += [rng.randrange(p) for _ in range(200)] + [rng.randrange(1 << 64) for _ in range(20)]
        return vals

    for a in cases(O.P, 32):
        exp = pow(a, -1, O.P) if a % O.P else 0
        assert H.call("hh_bn4_fp_inv", _be(a), out_sizes=(32,))[1] == _be(exp), a
    for a in cases(ON.P, 32):
        exp = pow(a, -1, ON.P) if a % ON.P else 0
        assert H.call("hh_bn_fp_op", 4, _be(a), _be(0), out_sizes=(32,))[1] == _be(exp), a
    for a in cases(OB.P, 48):
        exp = pow(a, -1, OB.P) if a % OB.P else 0
        assert H.call("hh_bls_fp_op", 4, a.to_bytes(48, "big"), bytes(48), out_sizes=(48,))[1] == exp.to_bytes(48, "big"), a
