"""oracle/bn256_ref.c (the C restatement of the reference's bn256 pairing and N x (Mul + Add) sums that bench.py times as
the CPU baseline) against the big-integer restatement oracle/bn256.py, which the reference's own fixtures pin
(tests/test_oracle_bn256.py): GT bytes of Suite.Pair, pointG1.Mul bytes, the sum, infinity and malformed inputs."""
import random
import time

import numpy as np

from oracle import bn256 as O
from tests import _oracle_c as OC


def _pts(rng, n):
    g1 = [O.g1_marshal(O.g1_mul(rng.randrange(1, O.ORDER), O.G1_GEN)) for _ in range(n)]
    g2 = [O.g2_marshal(O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN)) for _ in range(n)]
    return g1, g2


def test_pairing_bytes_match_the_python_restatement():
    rng = random.Random(11)
    g1, g2 = _pts(rng, 6)
    g1[2] = bytes(64)                     # Pair(infinity, Q) = 1
    g2[3] = bytes(128)
    # an on-curve twist point outside the order-n subgroup: the reference accepts it (point.go:466-499)
    x0 = 1
    while True:
        x = (x0, 1)
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_mul(x, x), x), O.TWIST_B))
        if y is not None:
            g2[4] = O.g2_marshal((x, y))
            break
        x0 += 1
    bad1 = bytearray(g1[5])
    bad1[63] ^= 1                         # off the curve
    g1.append(bytes(bad1))
    g2.append(g2[0])
    gt, st = OC.bn256_pair(b"".join(g1), b"".join(g2), threads=2)
    assert list(st) == [0] * 6 + [1] and not gt[6].any()
    for i in range(6):
        assert bytes(gt[i]) == O.pair_bytes(g1[i], g2[i]), i
    # bilinearity through the C code alone: e(aP, bQ) == e(abP, Q)
    a, b = rng.randrange(O.ORDER), rng.randrange(O.ORDER)
    lhs, _ = OC.bn256_pair(O.g1_marshal(O.g1_mul(a, O.G1_GEN)), O.g2_marshal(O.g2_mul(b, O.G2_GEN)), threads=1)
    rhs, _ = OC.bn256_pair(O.g1_marshal(O.g1_mul(a * b % O.ORDER, O.G1_GEN)), O.g2_marshal(O.G2_GEN), threads=1)
    assert bytes(lhs[0]) == bytes(rhs[0])


def test_g1_mul_and_sum_match_the_python_restatement():
    rng = random.Random(12)
    n = 40
    ks = [rng.randrange(O.ORDER) for _ in range(n)]
    ks[:6] = [0, 1, 2, O.ORDER - 1, O.ORDER, (1 << 256) - 1]
    pts = [O.g1_mul(rng.randrange(1, O.ORDER), O.G1_GEN) for _ in range(n)]
    pts[7] = None
    kb = np.frombuffer(b"".join(k.to_bytes(32, "big") for k in ks), dtype=np.uint8)
    pb = np.frombuffer(b"".join(O.g1_marshal(p) for p in pts), dtype=np.uint8)
    out, st = OC.bn256_g1_mul(kb, pb, threads=3)
    assert not st.any()
    acc = None
    for i in range(n):
        e = O.g1_mul(ks[i], pts[i])
        assert bytes(out[i]) == O.g1_marshal(e), i
        acc = O.g1_add(acc, e)
    for th in (1, 3, 64):
        s, st = OC.bn256_g1_mul_sum(kb, pb, threads=th)
        assert not st.any() and bytes(s) == O.g1_marshal(acc), th


def test_throughput_is_that_of_compiled_code():
    """(a sanity bound, not a benchmark: one core must do a pairing in a few milliseconds -- the reference's assembler
    version needs 1.6 ms, README.md:34 -- or the 'CPU baseline' of bench.py would be measuring an interpreter)"""
    rng = random.Random(13)
    g1, g2 = _pts(rng, 4)
    t0 = time.perf_counter()
    OC.bn256_pair(b"".join(g1 * 8), b"".join(g2 * 8), threads=1)
    dt = (time.perf_counter() - t0) / 32
    assert dt < 0.02, dt


def test_bls12381_g1_mul_sum_matches_the_python_oracle():
    """oracle/bls12381_g1_ref.c (the N x (Mul + Add) baseline of the 2^20-point MSM config) against oracle/bls12381.py"""
    from oracle import bls12381 as OB

    rng = random.Random(14)
    n = 24
    ks = [rng.randrange(OB.R) for _ in range(n)]
    ks[:5] = [0, 1, OB.R - 1, OB.R, (1 << 256) - 1]
    pts = [OB.g1_mul(rng.randrange(1, OB.R), OB.G1_GEN) for _ in range(n)]
    pts[6] = None
    kb = np.frombuffer(b"".join(k.to_bytes(32, "big") for k in ks), dtype=np.uint8)
    pb = np.frombuffer(b"".join(OB.g1_serialize_unc(p) for p in pts), dtype=np.uint8)
    acc = None
    for k, p in zip(ks, pts):
        acc = OB.g1_add(acc, OB.g1_mul(k, p))
    for th in (1, 5):
        out, st = OC.bls12381_g1_mul_sum(kb, pb, threads=th)
        assert not st.any() and bytes(out) == OB.g1_serialize_unc(acc), th
    bad = bytearray(pb.tobytes())
    bad[96 * 3] |= 0x80   # compression flag on an uncompressed encoding
    out, st = OC.bls12381_g1_mul_sum(kb, np.frombuffer(bytes(bad), dtype=np.uint8), threads=2)
    assert st[3] == 1 and st.sum() == 1


def test_g2_mul_matches_the_python_restatement():
    """ora_bn256_g2_mul (twist.go:75-205 restated in C) against oracle/bn256.py: edge scalars as plain 256-bit integers,
    the point at infinity, a point of the twist outside the order-n subgroup (accepted by UnmarshalBinary, point.go:466-499),
    a point off the curve"""
    rng = random.Random(41)
    ks = [0, 1, 2, O.ORDER - 1, O.ORDER, O.ORDER + 1, (1 << 256) - 1, 1 << 255] + [rng.randrange(1 << 256) for _ in range(4)]
    pts = [O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN) for _ in ks]
    pts[2] = None
    enc = [O.g2_marshal(p) for p in pts]
    x0 = 1
    while True:  # a twist point outside the subgroup: any curve point is, with overwhelming probability
        x = (x0, 1)
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_mul(x, x), x), O.TWIST_B))
        if y is not None:
            pts[9] = (x, y)
            enc[9] = O.g2_marshal(pts[9])
            break
        x0 += 1
    assert O.g2_mul(O.ORDER, pts[9]) is not None
    k = b"".join(v.to_bytes(32, "big") for v in ks)
    for threads in (1, 4):
        out, st = OC.bn256_g2_mul(k, b"".join(enc), threads=threads)
        assert not st.any()
        for i, (v, p) in enumerate(zip(ks, pts)):
            assert bytes(out[i]) == O.g2_marshal(O.g2_mul(v, p)), (threads, i)
    bad = bytearray(enc[0])
    bad[127] ^= 1
    out, st = OC.bn256_g2_mul(k[:32], bytes(bad), threads=1)
    assert st[0] == 1 and not out.any()
