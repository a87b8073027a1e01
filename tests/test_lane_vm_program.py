"""The lane machine's programs (kyber_amd/csrc/gen_lane_vm.py -> lane_vm_bls12381.inc), replayed on the CPU with the
device's limb arithmetic (LProg.simulate: 28-bit balanced limbs, 64-bit columns, the signed Montgomery reduction, every
overflow an assertion) against the oracle's scalar multiplication -- kilic/g1.go:110-116, kilic/g2.go (G1Elt.Mul /
G2Elt.Mul).  The kernel runs these records unchanged; what the GPU tests add is the interpreter itself."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber_amd", "csrc"))
import gen_lane_vm as G  # noqa: E402

from oracle import bls12381 as O  # noqa: E402

EDGE = [1, 2, 3, 15, 16, 17, O.R - 1, O.R + 1, (1 << 256) - 1, 1 << 255, G.BLS_Z, G.BLS_Z - 1, G.BLS_Z * G.BLS_Z, G.BLS_Z ** 2 - 1,
        G.BLS_Z ** 3, (1 << 128) - 1]


@pytest.fixture(scope="module")
def progs():
    return G.build_bls12381_g1_mul(), G.build_bls12381_g2_mul()


def test_regular_recoding():
    rng = random.Random(1)
    for npos in (17, 33):
        for k in [1, 3, 16 ** npos - 1, 16 ** (npos - 1) + 1] + [rng.randrange(16 ** npos) | 1 for _ in range(200)]:
            d = G.regular_digits(k, npos)
            assert all(x & 1 and abs(x) <= 15 for x in d) and sum(x << (4 * i) for i, x in enumerate(d)) == k
    # the byte strings the prep kernel must produce: every sub-scalar, its correction included, sums back to k
    z = G.BLS_Z
    for k in EDGE + [0, O.R] + [rng.randrange(1 << 256) for _ in range(50)]:
        for digs, npos, base, nsub in ((G.bls_g1_digits(k), G.G1_NPOS, z * z, 2), (G.bls_g2_digits(k), G.G2_NPOS, z, 4)):
            tot = 0
            for j in range(nsub):
                b = digs[j * (npos + 1):(j + 1) * (npos + 1)]
                flip = bool(j & 1)
                v = 0
                for i, x in enumerate(b[:npos]):
                    m = 2 * (x & 15) + 1
                    v += (-m if bool(x >> 7) != flip else m) << (4 * i)
                c = 2 if (b[npos] & 15) == G.E2P else 1
                assert bool(b[npos] >> 7) != flip  # the correction subtracts
                tot += (v - c) * base ** j
            assert tot == k


def test_bounds_hold_for_any_input(progs):
    for P in progs:
        assert P.check_bounds() < 63


def test_g1_program_against_the_oracle(progs):
    P1 = progs[0]
    rng = random.Random(5)
    for k in EDGE[:10] + [rng.randrange(O.R) for _ in range(4)]:
        pt = O.g1_mul(rng.randrange(1, O.R), O.G1_GEN)
        outs, flags, _ = P1.simulate([[pt[0], pt[1]]], [G.bls_g1_digits(k)])
        assert not flags[0] and (outs[0][0], outs[0][1]) == O.g1_mul(k, pt), hex(k)


def test_g2_program_against_the_oracle(progs):
    P2 = progs[1]
    rng = random.Random(6)
    for k in EDGE[4:] + [rng.randrange(O.R) for _ in range(3)]:
        pt = O.g2_mul(rng.randrange(1, O.R), O.G2_GEN)
        d = G.bls_g2_digits(k)
        outs, flags, _ = P2.simulate([[pt[0][0], pt[1][0]], [pt[0][1], pt[1][1]]], [d, d])
        assert not (flags[0] & flags[1])
        assert ((outs[0][0], outs[1][0]), (outs[0][1], outs[1][1])) == O.g2_mul(k, pt), hex(k)


def test_infinity_and_exceptional_additions_leave_z_zero(progs):
    """k = 0 mod r (the result is the point at infinity) and a scalar that makes the accumulator meet a table entry
    both end with Z = 0 -- the flag the encode kernel turns into the per-lane recomputation."""
    P1, P2 = progs
    pt = O.g1_mul(77, O.G1_GEN)
    for k in (0, O.R, 2 * O.R):
        _, flags, _ = P1.simulate([[pt[0], pt[1]]], [G.bls_g1_digits(k)])
        assert flags[0] & 1
    q = O.g2_mul(78, O.G2_GEN)
    d = G.bls_g2_digits(O.R)
    _, flags, _ = P2.simulate([[q[0][0], q[1][0]], [q[0][1], q[1][1]]], [d, d])
    assert flags[0] & flags[1] & 1
