"""The GT exponentiation programs of the tower machine (gen_tower_vm.py build_*_gtmul: GTElt.Mul, kilic/gt.go:79-84;
pointGT.Mul -> gfP12.Exp, pairing/bn256/point.go:613, gfp12.go:177-192) replayed on the CPU against the oracles: values,
the membership verdict of the BLS12-381 program on members / non-members of every kind its three comparisons separate,
the device's limb arithmetic with overflow assertions, worst-case bounds."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber_amd", "csrc"))
import gen_tower_vm as G  # noqa: E402

from oracle import bls12381 as O  # noqa: E402
from oracle import bn254 as O4  # noqa: E402
from oracle import bn256 as O6  # noqa: E402


def _inputs(f, a):
    """oracle Fp12 (six Fp2 coefficients of the w-basis) -> the twelve inputs the decode kernel leaves (a R1 mod p)"""
    return [c * f.R1 % f.p for j in range(6) for c in a[j]]


def _bytes(res, size, clen):
    out = bytearray(size)
    for off, (v, _) in res["gt"].items():
        out[off:off + clen] = v.to_bytes(clen, "big")
    return bytes(out)


@pytest.fixture(scope="module")
def bls_prog():
    return G.build_bls12381_gtmul()


@pytest.fixture(scope="module")
def bls_member():
    return O.pair(O.g1_mul(0xC0FFEE, O.G1_GEN), O.g2_mul(0xBADC0DE, O.G2_GEN))


def test_bls12381_gtmul_equals_the_oracle_power(bls_prog, bls_member):
    rng = random.Random(3)
    f = bls_prog.f
    for k in (0, 1, 2, O.R - 1, O.R, (1 << 256) - 1, rng.getrandbits(256), rng.getrandbits(255), rng.getrandbits(64)):
        _, res = bls_prog.simulate(_inputs(f, bls_member), expo=k)
        assert not res["not_one"], k
        assert _bytes(res, 576, 48) == O.gt_to_bytes(O.f12_pow(bls_member, k)), k
    assert _bytes(bls_prog.simulate(_inputs(f, O.F12_ONE), expo=5)[1], 576, 48) == O.gt_to_bytes(O.F12_ONE)


def test_bls12381_gtmul_membership_verdict(bls_prog, bls_member):
    """accepted <=> f^r = 1 (the oracle's gt_from_bytes): a random element (not unitary), zero, a unitary element outside
    the cyclotomic subgroup, an element of the cyclotomic subgroup outside GT, a member times an element of order
    dividing the cofactor, -1 (order 2: unitary, cyclotomic? no: (-1)^(p^4 - p^2 + 1) = -1)"""
    rng = random.Random(9)
    f = bls_prog.f
    p = f.p

    def verdict(a):
        return not bls_prog.simulate(_inputs(f, a), expo=3)[1]["not_one"]

    def member(a):
        return O.f12_pow(a, O.R) == O.F12_ONE

    x = [(rng.randrange(p), rng.randrange(p)) for _ in range(6)]
    zero = [(0, 0)] * 6
    unitary = O.f12_mul(O.f12_frob(x, 6), O.f12_inv(x))             # x^(p^6 - 1): f conj(f) = 1
    cyclo = O.f12_mul(O.f12_frob(unitary, 2), unitary)               # ^(p^2 + 1): order divides Phi_12(p) = r h
    minus_one = [((p - 1), 0)] + [(0, 0)] * 5
    cases = [x, zero, unitary, cyclo, O.f12_mul(cyclo, bls_member), minus_one, bls_member, O.f12_pow(bls_member, 12345),
             O.f12_pow(cyclo, (p ** 4 - p ** 2 + 1) // O.R)]         # the last: the cofactor power of a cyclotomic element IS a member
    want = [member(a) for a in cases]
    assert want == [False, False, False, False, False, False, True, True, True]
    assert [verdict(a) for a in cases] == want


def test_bls12381_gtmul_in_device_arithmetic_and_bounds(bls_prog, bls_member):
    f = bls_prog.f
    k = (1 << 256) - 0x1234567
    _, res = bls_prog.simulate_limbs(_inputs(f, bls_member), expo=k)
    assert not res["not_one"] and _bytes(res, 576, 48) == O.gt_to_bytes(O.f12_pow(bls_member, k))
    bls_prog.simulate_limbs([f.p - 1] * 12, expo=(1 << 256) - 1)   # garbage in: nothing may overflow
    col, val = bls_prog.check_bounds()
    assert col < 63 and val < 1024


@pytest.mark.parametrize("build,Ob", [(G.build_bn256_gtmul, O6), (G.build_bn254_gtmul, O4)])
def test_bn_gtmul_equals_the_oracle_power_for_any_element(build, Ob):
    """no membership in the reference (UnmarshalBinary takes any twelve residues): pairing values and arbitrary elements,
    the zero element and non-invertible-looking ones included"""
    rng = random.Random(17)
    prog = build()
    f = prog.f
    p = f.p
    gt = Ob.gt_unmarshal(Ob.pair_bytes(Ob.g1_marshal(Ob.G1_GEN), Ob.g2_marshal(Ob.G2_GEN)))
    elems = [[(rng.randrange(p), rng.randrange(p)) for _ in range(6)], [(0, 0)] * 6, [(1, 0)] + [(0, 0)] * 5, gt]
    for a in elems:
        for k in (0, 1, rng.getrandbits(256), (1 << 256) - 1):
            _, res = prog.simulate(_inputs(f, a), expo=k)
            assert _bytes(res, 384, 32) == Ob.gt_marshal(Ob.f12_pow(a, k)), k
    a, k = elems[0], rng.getrandbits(256) | (1 << 255)
    _, res = prog.simulate_limbs(_inputs(f, a), expo=k)
    assert _bytes(res, 384, 32) == Ob.gt_marshal(Ob.f12_pow(a, k))
    prog.simulate_limbs([p - 1] * 12, expo=(1 << 256) - 1)
    col, val = prog.check_bounds()
    assert col < 63 and val < 1024
