"""BLS12-381 Pair / ValidatePairing / fused Verify with unvouched-for G2 operands: their r-torsion test (kilic/g2.go
FromCompressed -> InCorrectSubgroup, UnmarshalBinary's last rule) is decided by the tower machine at the end of the
Miller loop (gen_tower_vm.py bls_g2_member_check) instead of per lane.  Statuses, their precedence in argument order
and the outputs must be what the per-lane decode gave: members, big-order non-members, points of every small prime
order of the cofactor (the loop's exceptional steps), member + small-order point; G1 operand fine / malformed / off its
own subgroup / at infinity; compressed and uncompressed encodings; trusted flags."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _twist_points(O):
    rng = random.Random(31)
    member = O.g2_mul(rng.randrange(1, O.R), O.G2_GEN)
    while True:
        x = (rng.randrange(O.P), rng.randrange(O.P))
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(x), x), (4, 4)))
        if y is not None:
            big = (x, y)
            break
    cof = O.g2_mul(O.R, big)
    small = []
    for q in (13, 23, 2713, 11953, 262069):
        e = O.H2
        while e % q == 0:
            e //= q
        pt = O.g2_mul(e, cof)
        while pt is not None and O.g2_mul(q, pt) is not None:
            pt = O.g2_mul(q, pt)
        if pt is not None:
            small.append(pt)
    return member, big, small


def _off_subgroup_g1(O):
    x = 1
    while True:
        y = O.fp_sqrt((x * x * x + 4) % O.P)
        if y is not None and not O.g1_in_subgroup((x, y)):
            return (x, y)
        x += 1


def test_pair_statuses_and_outputs_with_machine_decided_membership():
    from kyber_amd.pairing import bls12381 as bls
    from oracle import bls12381 as O

    member, big, small = _twist_points(O)
    g2s = [(member, 0), (big, 2), (O.g2_add(member, small[0]), 2), (None, 0)] + [(s, 2) for s in small] + [(O.g2_mul(5, member), 0)]
    p = O.g1_mul(11, O.G1_GEN)
    off1 = _off_subgroup_g1(O)
    for U in (0, bls.F_UNCOMPRESSED):
        s1 = O.g1_serialize_unc if U else O.g1_compress
        s2 = O.g2_serialize_unc if U else O.g2_compress
        good1 = s1(p)
        bad1 = bytes([good1[0] ^ (0x80 if not U else 0x80)]) + good1[1:]      # flag rule broken: status 1
        g1s = [(good1, 0, p), (bad1, 1, None), (s1(off1), 2, None), (s1(None), 0, None)]
        A, B, want_st, want = [], [], [], []
        for e1, st1, pv in g1s:
            for q, st2 in g2s:
                A.append(e1); B.append(s2(q))
                st = st1 or st2                                              # G1 is the first argument
                want_st.append(st)
                if st:
                    want.append(bytes(576))
                elif pv is None or q is None:
                    want.append(O.gt_to_bytes(O.F12_ONE))
                else:
                    want.append(None)                                        # checked below for a few
        reps = 3                                                             # 4 x 10 x 3 = 120 lanes: ragged second batch
        gt, st = bls.batch_pair(b"".join(A) * reps, b"".join(B) * reps, U)
        assert list(st) == want_st * reps, U
        for r in range(len(A) * reps):
            w = want[r % len(A)]
            if w is not None:
                assert bytes(gt[r]) == w, (U, r)
        assert bytes(gt[0]) == O.gt_to_bytes(O.pair(p, member)) and bytes(gt[len(g2s) - 1]) == O.gt_to_bytes(O.pair(p, O.g2_mul(5, member)))
        # vouched-for G2 operands are not examined (status 0 whatever they are); the G1 rules still apply
        gt_t, st_t = bls.batch_pair(b"".join(A), b"".join(B), U | bls.F_TRUSTED(1))
        assert list(st_t) == [s1_ for (_, s1_, _) in g1s for _ in g2s]


def test_validate_pairing_and_verify_status_precedence():
    from kyber_amd.pairing import bls12381 as bls
    from oracle import bls12381 as O

    member, big, small = _twist_points(O)
    k = 77
    p, kp = O.g1_mul(9, O.G1_GEN), O.g1_mul(9 * k, O.G1_GEN)
    q = O.g2_mul(k, O.G2_GEN)
    c1, c2 = O.g1_compress, O.g2_compress
    bad1 = bytes(48)                                                         # no compression flag: status 1
    rows = [  # p1, p2, inv1, inv2 -> (ok, status)
        (c1(p), c2(q), c1(kp), c2(O.G2_GEN), 1, 0),                          # e(P, kG2) == e(kP, G2)
        (c1(p), c2(q), c1(p), c2(O.G2_GEN), 0, 0),
        (c1(p), c2(big), c1(kp), c2(O.G2_GEN), 0, 2),                        # p2 outside the subgroup
        (c1(p), c2(q), c1(kp), c2(small[1]), 0, 2),                          # inv2 of small order
        (c1(p), c2(big), bad1, c2(O.G2_GEN), 0, 2),                          # p2 (argument 1) before inv1 (argument 2)
        (bad1, c2(big), c1(kp), c2(O.G2_GEN), 0, 1),                         # p1 (argument 0) before p2
        (c1(p), c2(q), bad1, c2(big), 0, 1),                                 # inv1 before inv2
        (c1(None), c2(big), c1(None), c2(O.G2_GEN), 0, 2),                   # G1 at infinity does not hide the G2 verdict
        (c1(None), c2(q), c1(None), c2(O.G2_GEN), 1, 0),                     # both pairs dead: 1 == 1
        (c1(p), c2(None), c1(kp), c2(small[0]), 0, 2),
    ]
    reps = 7                                                                 # 70 lanes
    cols = [b"".join(r[i] for r in rows) * reps for i in range(4)]
    ok, st = bls.batch_validate_pairing(*cols)
    assert list(st) == [r[5] for r in rows] * reps
    assert list(ok) == [r[4] for r in rows] * reps
    # every G2 operand vouched for: nothing about them is examined
    ok_t, st_t = bls.batch_validate_pairing(*cols, bls.F_TRUSTED(1) | bls.F_TRUSTED(3))
    assert list(st_t) == [1 if bad1 in (r[0], r[2]) else 0 for r in rows] * reps
    # fused Verify with signatures on G1: the key (a G2 point) is argument 0 of the call
    from kyber_amd.sign import bls as sbls  # noqa: F401  (module import keeps the DST constants in one place)
    msgs = [bytes([i]) * 32 for i in range(6)]
    Hm, _ = bls.batch_hash_g1(msgs)
    x = 123456789
    sigs, _ = bls.g1_batch_mul(x.to_bytes(32, "big") * 6, Hm)
    X = O.g2_mul(x, O.G2_GEN)
    keys = [c2(X), c2(big), c2(small[2]), c2(O.g2_add(X, small[0])), c2(X), c2(None)]
    v, st = bls.batch_verify_g1(b"".join(keys), msgs, sigs)
    assert list(st) == [0, 2, 2, 2, 0, 0] and list(v) == [1, 0, 0, 0, 1, 0]


def test_bn254_pair_and_validate_with_machine_decided_membership():
    """pairing/bn254's UnmarshalBinary rejects G2 points outside the order-n subgroup (twist.go:47-66); for the pairing
    entry points that verdict now comes from the end of the ate loop.  bn256, whose reference has no such rule, still
    pairs whatever is on the twist."""
    from kyber_amd.pairing import bn254 as bn4
    from kyber_amd.pairing import bn256 as bn6
    from oracle import bn254 as O
    from oracle import bn256 as O6

    rng = random.Random(41)
    member = O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN)
    while True:
        xx = (rng.randrange(O.P), rng.randrange(O.P))
        yy = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(xx), xx), O.TWIST_B))
        if yy is not None:
            break
    big = (xx, yy)
    h = 2 * O.P - O.ORDER
    small = [s for s in (O.g2_mul(O.ORDER * h // q, big) for q in O.G2_COFACTOR_PRIMES) if s is not None]
    g2s = [(member, 0), (big, 2), (O.g2_add(member, small[0]), 2), (None, 0)] + [(s, 2) for s in small]
    p = O.g1_mul(11, O.G1_GEN)
    good1 = O.g1_marshal(p)
    bad1 = (O.P).to_bytes(32, "big") + good1[32:]                              # coordinate >= p: status 1
    g1s = [(good1, 0, p), (bad1, 1, None), (O.g1_marshal(None), 0, None)]
    A, B, want_st, want = [], [], [], []
    for e1, st1, pv in g1s:
        for q, st2 in g2s:
            A.append(e1); B.append(O.g2_marshal(q)); want_st.append(st1 or st2)
            want.append(bytes(384) if (st1 or st2) else (O.gt_marshal(O.F12_ONE) if (pv is None or q is None) else O.gt_marshal(O.pair(pv, q))))
    reps = 3
    gt, st = bn4.batch_pair(b"".join(A) * reps, b"".join(B) * reps)
    assert list(st) == want_st * reps
    for r in range(len(A) * reps):
        assert bytes(gt[r]) == want[r % len(A)], r
    _, st_t = bn4.batch_pair(b"".join(A), b"".join(B), bn4.F_TRUSTED(1))
    assert list(st_t) == [s1 for (_, s1, _) in g1s for _ in g2s]
    # ValidatePairing: precedence in argument order
    k = 77
    kp, q = O.g1_mul(11 * k, O.G1_GEN), O.g2_mul(k, O.G2_GEN)
    m1, m2 = O.g1_marshal, O.g2_marshal
    rows = [(m1(p), m2(q), m1(kp), m2(O.G2_GEN), 1, 0), (m1(p), m2(q), m1(p), m2(O.G2_GEN), 0, 0),
            (m1(p), m2(big), m1(kp), m2(O.G2_GEN), 0, 2), (m1(p), m2(q), m1(kp), m2(small[1]), 0, 2),
            (m1(p), m2(big), bad1, m2(O.G2_GEN), 0, 2), (bad1, m2(big), m1(kp), m2(O.G2_GEN), 0, 1),
            (m1(p), m2(q), bad1, m2(big), 0, 1), (m1(None), m2(big), m1(None), m2(O.G2_GEN), 0, 2),
            (m1(None), m2(q), m1(None), m2(O.G2_GEN), 1, 0)]
    cols = [b"".join(r[i] for r in rows) * 8 for i in range(4)]               # 72 lanes
    ok, st = bn4.batch_validate_pairing(*cols)
    assert list(st) == [r[5] for r in rows] * 8 and list(ok) == [r[4] for r in rows] * 8
    # bn256: a point of the twist outside the subgroup is an ordinary operand (point.go:466-499 checks the curve only)
    while True:
        xx = (rng.randrange(O6.P), rng.randrange(O6.P))
        yy = O6.f2_sqrt(O6.f2_add(O6.f2_mul(O6.f2_sqr(xx), xx), O6.TWIST_B))
        if yy is not None:
            break
    p6 = O6.g1_mul(5, O6.G1_GEN)
    gt6, st6 = bn6.batch_pair(O6.g1_marshal(p6), O6.g2_marshal((xx, yy)))
    assert st6[0] == 0 and bytes(gt6[0]) == O6.gt_marshal(O6.pair(p6, (xx, yy)))
