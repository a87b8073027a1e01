"""GPU parity tests for the bn256 hot path (config 5), through the C ABI: the reference's golden
vectors (BDN fixtures, Hash outputs feed the signatures), bit-exact vs the oracle restatement of
pairing/bn256, and size-independent properties at batch scale."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

from oracle import bn256 as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bn():
    import torch

    assert torch.cuda.is_available()
    from kyber_amd.pairing import bn256 as bn

    return bn


@pytest.fixture(scope="module")
def G(golden_dir):
    return json.load(open(os.path.join(golden_dir, "bn256.json")))


def _scalars(label: bytes, n: int) -> np.ndarray:
    raw = hashlib.shake_256(label).digest(n * 64)
    out = np.empty((n, 32), dtype=np.uint8)
    for i in range(n):
        out[i] = np.frombuffer((int.from_bytes(raw[64 * i:64 * i + 64], "big") % O.ORDER).to_bytes(32, "big"), dtype=np.uint8)
    return out


def _fp(x):
    return x.to_bytes(32, "big")


def test_bdn_golden_fixtures(bn, G):
    Hm = O.g1_marshal(O.hash_to_g1(G["bdn_msg"].encode()))
    privs = b"".join(bytes.fromhex(p) for p in G["bdn_privs"])
    pubs, st = bn.g2_commit(privs)  # public keys = x * G2 base
    assert not st.any()
    sigs, st = bn.g1_commit(privs, Hm)  # signatures = x * Hash(msg)
    assert not st.any()
    for i in range(3):
        assert bytes(pubs[i]).hex() == G["bdn_pubs"][i]
        assert bytes(sigs[i]).hex() == G["bdn_sigs"][i]
    ok, st = bn.batch_validate_pairing(Hm * 3, pubs, sigs, bn.G2_BASE * 3)
    assert not st.any() and ok.all()
    ok, st = bn.batch_validate_pairing(Hm * 3, pubs, np.roll(sigs, 1, axis=0), bn.G2_BASE * 3)
    assert not ok.any()


def test_bdn_aggregated_key_terms(bn, G):
    # publicTerms[i] = c_i * P_i + P_i with P_i = (i+1) G2 (mask.go:57-61); their sum is the fixture
    ks = b"".join(_fp(i + 1) for i in range(3))
    P, _ = bn.g2_commit(ks)
    cp1 = b"".join(_fp(int(c, 16) + 1) for c in G["bdn_coefs"])
    terms, st = bn.g2_batch_mul(cp1, P)
    assert not st.any()
    acc = None
    for t in terms:
        acc = O.g2_add(acc, O.g2_unmarshal(bytes(t)))
    assert O.g2_marshal(acc).hex() == G["bdn_agg_key"]


def test_mul_vs_oracle_and_edges(bn):
    rng = random.Random(4)
    ks = [0, 1, 2, 16, 17, O.ORDER - 1, O.ORDER, O.ORDER + 3, (1 << 256) - 1] + [rng.randrange(O.ORDER) for _ in range(15)]
    hs = [rng.randrange(1, O.ORDER) for _ in ks]
    p1 = [O.g1_marshal(O.g1_mul(h, O.G1_GEN)) for h in hs]
    p2 = [O.g2_marshal(O.g2_mul(h, O.G2_GEN)) for h in hs]
    p1[2], p2[2] = bytes(64), bytes(128)
    x, y = O.G1_GEN
    p1[3] = _fp(x + O.P) + _fp(y)  # non-canonical coordinate accepted (point.go:218-221)
    kb = [_fp(k) for k in ks]
    out, st = bn.g1_batch_mul(b"".join(kb), b"".join(p1))
    assert not st.any()
    for i in range(len(ks)):
        assert bytes(out[i]) == O.g1_mul_bytes(kb[i], p1[i]), i
    out, st = bn.g2_batch_mul(b"".join(kb), b"".join(p2))
    assert not st.any()
    for i in range(len(ks)):
        assert bytes(out[i]) == O.g2_mul_bytes(kb[i], p2[i]), i
    out, st = bn.g1_batch_mul(kb[0] * 2, _fp(5) * 2 + p1[0])
    assert list(st) == [1, 0] and not out[0].any()


def test_pairing_bytes_vs_oracle(bn):
    rng = random.Random(5)
    n = 5
    g1 = [O.g1_marshal(O.g1_mul(rng.randrange(1, O.ORDER), O.G1_GEN)) for _ in range(n)]
    g2 = [O.g2_marshal(O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN)) for _ in range(n)]
    g1[3] = bytes(64)
    # an on-curve G2 point outside the order-n subgroup (accepted by the reference, point.go:466-499)
    while True:
        x = (rng.randrange(O.P), rng.randrange(O.P))
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(x), x), O.TWIST_B))
        if y is not None:
            g2[4] = O.g2_marshal((x, y))
            break
    gt, st = bn.batch_pair(b"".join(g1), b"".join(g2))
    assert not st.any()
    for i in range(n):
        assert bytes(gt[i]) == O.pair_bytes(g1[i], g2[i]), i


def test_bilinearity_at_scale(bn):
    """e(aP, bQ) == e(abP, Q) == e(P, abQ) (suite_test.go:231-259) for 2048 independent pairs."""
    n = 2048
    a, b = _scalars(b"bn/a", n), _scalars(b"bn/b", n)
    ab = np.empty_like(a)
    for i in range(n):
        v = int.from_bytes(bytes(a[i]), "big") * int.from_bytes(bytes(b[i]), "big") % O.ORDER
        ab[i] = np.frombuffer(_fp(v), dtype=np.uint8)
    aP, _ = bn.g1_commit(a)
    abP, _ = bn.g1_commit(ab)
    bQ, _ = bn.g2_commit(b)
    abQ, _ = bn.g2_commit(ab)
    G1 = np.tile(np.frombuffer(bn.G1_BASE, dtype=np.uint8), (n, 1))
    G2 = np.tile(np.frombuffer(bn.G2_BASE, dtype=np.uint8), (n, 1))
    e1, s1 = bn.batch_pair(aP, bQ)
    e2, s2 = bn.batch_pair(abP, G2)
    e3, s3 = bn.batch_pair(G1, abQ)
    assert not (s1.any() or s2.any() or s3.any())
    assert (e1 == e2).all() and (e1 == e3).all()
    assert len({bytes(r) for r in e1[:64]}) == 64


def test_suite_mirror(bn):
    s = bn.NewSuite()
    a, b = s.G1().Scalar().SetInt64(5), s.G1().Scalar().SetInt64(7)
    P, Q = s.G1().Point().Mul(a, None), s.G2().Point().Mul(b, None)
    ab = s.G1().Scalar().Mul(a, b)
    assert s.Pair(P, Q).Equal(s.Pair(s.G1().Point().Mul(ab, None), s.G2().Point().Base()))
    assert s.ValidatePairing(P, Q, s.G1().Point().Mul(ab, None), s.G2().Point().Base())


def test_gt_mul_vs_oracle_and_homomorphism(bn):
    rng = random.Random(8)
    n = 4
    g1 = [O.g1_marshal(O.g1_mul(rng.randrange(1, O.ORDER), O.G1_GEN)) for _ in range(n)]
    g2 = [O.g2_marshal(O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN)) for _ in range(n)]
    gt, _ = bn.batch_pair(b"".join(g1), b"".join(g2))
    ks = [0, 1, O.ORDER - 1, rng.randrange(O.ORDER)]
    kb = b"".join(_fp(k) for k in ks)
    out, st = bn.gt_batch_mul(kb, gt)
    assert not st.any()
    for i in range(n):
        assert bytes(out[i]) == O.gt_mul_bytes(_fp(ks[i]), bytes(gt[i])), i
    # e(P, Q)^k == e(kP, Q) at batch scale
    m = 512
    k = _scalars(b"bn/gt/k", m)
    P, _ = bn.g1_commit(_scalars(b"bn/gt/p", m))
    G2 = np.tile(np.frombuffer(bn.G2_BASE, dtype=np.uint8), (m, 1))
    e, _ = bn.batch_pair(P, G2)
    kP, _ = bn.g1_batch_mul(k, P)
    ek, _ = bn.batch_pair(kP, G2)
    out, st = bn.gt_batch_mul(k, e)
    assert not st.any() and (out == ek).all()


def test_hash_g1_fixtures_and_batch(bn, G):
    for h in G["hash_g1"]:
        out, st = bn.batch_hash_g1([bytes.fromhex(h["msg_hex"])])
        assert st[0] == 0 and bytes(out[0]).hex() == h["point"]  # point_test.go:13-45
    msgs = [hashlib.sha256(b"m%d" % i).digest() for i in range(500)]
    out, st = bn.batch_hash_g1(msgs)
    assert not st.any()
    for i in range(0, 500, 37):
        assert bytes(out[i]) == O.g1_marshal(O.hash_to_g1(msgs[i]))
    for ln in (0, 1, 55, 56, 64, 65, 119, 120):
        m = bytes((3 * i + ln) & 0xFF for i in range(ln))
        out, st = bn.batch_hash_g1([m, m])
        assert bytes(out[1]) == O.g1_marshal(O.hash_to_g1(m)), ln


def test_batch_unmarshal(bn):
    """kyb_bn256_g*_unmarshal = N x UnmarshalBinary + MarshalBinary (pairing/bn256/point.go:206-238, 466-499)."""
    rng = random.Random(41)
    x, y = O.G1_GEN
    fp = lambda v: v.to_bytes(32, "big")
    g1 = [O.g1_marshal(O.g1_mul(rng.randrange(1, O.ORDER), O.G1_GEN)) for _ in range(5)]
    batch = g1 + [bytes(64), fp(5) + fp(5), fp(x + O.P) + fp(y)]
    out, st = bn.g1_batch_unmarshal(b"".join(batch))
    assert list(st) == [0] * 6 + [1, 0]
    for i in range(6):
        assert bytes(out[i]) == batch[i]
    assert not out[6].any() and bytes(out[7]) == O.g1_marshal(O.G1_GEN)
    g2 = [O.g2_marshal(O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN)) for _ in range(3)]
    # on the twist, outside the order-r subgroup: accepted, as the reference does
    while True:
        X = (rng.randrange(O.P), rng.randrange(O.P))
        Y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(X), X), O.TWIST_B))
        if Y is not None:
            break
    assert O.g2_mul(O.ORDER, (X, Y)) is not None
    batch = g2 + [bytes(128), fp(1) * 4, O.g2_marshal((X, Y))]
    out, st = bn.g2_batch_unmarshal(b"".join(batch), bn.F_TRUSTED(0))  # bn256 has no check to skip
    assert list(st) == [0, 0, 0, 0, 1, 0]
    for i in (0, 1, 2, 3, 5):
        assert bytes(out[i]) == batch[i]
    assert not out[4].any()


def test_validate_pairing_product_form_under_trusted_g2(bn):
    """KYB_F_TRUSTED on both G2 operands selects the product-form program (one final exponentiation): same booleans as
    the reference's two pairings + Equal, on valid, forged and infinity rows."""
    n = 1500
    a, b = _scalars(b"bn/pf/a", n), _scalars(b"bn/pf/b", n)
    ab = np.empty_like(a)
    for i in range(n):
        v = int.from_bytes(bytes(a[i]), "big") * int.from_bytes(bytes(b[i]), "big") % O.ORDER
        ab[i] = np.frombuffer(_fp(v), dtype=np.uint8)
    aP, _ = bn.g1_commit(a)
    abP, _ = bn.g1_commit(ab)
    bQ, _ = bn.g2_commit(b)
    G2 = np.tile(np.frombuffer(bn.G2_BASE, dtype=np.uint8), (n, 1))
    forged = np.array(abP, copy=True)
    forged[::7] = aP[::7]
    p1, inv1 = np.array(aP, copy=True), forged
    p2, inv2 = np.array(bQ, copy=True), np.array(G2, copy=True)
    p1[5] = 0          # e(inf, Q) == e(abP, G2) is false
    p1[6] = 0
    inv1[6] = 0        # 1 == 1
    inv2[12] = 0       # e(aP, bQ) == e(., inf) = 1 is false
    ok0, st0 = bn.batch_validate_pairing(p1, p2, inv1, inv2)
    ok1, st1 = bn.batch_validate_pairing(p1, p2, inv1, inv2, bn.F_TRUSTED(1) | bn.F_TRUSTED(3))
    ok2, st2 = bn.batch_validate_pairing(p1, p2, inv1, inv2, bn.F_TRUSTED_ALL)
    assert not st0.any() and not st1.any() and not st2.any()
    exp = np.ones(n, dtype=np.uint8)
    exp[::7] = 0
    exp[5], exp[6], exp[12] = 0, 1, 0
    assert (np.asarray(ok0) == exp).all() and (np.asarray(ok1) == exp).all() and (np.asarray(ok2) == exp).all()


def test_g2_mul_gls_under_trusted_flag(bn):
    """bn256 G2: plain ladder by default (unchecked inputs), GLS when the caller vouches for the subgroup: same bytes."""
    n = 2000
    k = _scalars(b"bn/gls/k", n)
    k[0] = 0
    k[1] = np.frombuffer(_fp(O.ORDER), dtype=np.uint8)
    k[2] = 0xFF
    Q, _ = bn.g2_commit(_scalars(b"bn/gls/q", n))
    a, st = bn.g2_batch_mul(k, Q)
    b, st2 = bn.g2_batch_mul(k, Q, bn.F_TRUSTED(0))
    assert not st.any() and not st2.any() and (np.asarray(a) == np.asarray(b)).all()
    for i in (0, 1, 2, 3, n - 1):
        assert bytes(b[i]) == O.g2_mul_bytes(bytes(k[i]), bytes(Q[i]))


def test_same_base_commit_with_an_off_subgroup_g2_base_at_2p18(bn):
    """kyb_bn256_g2_mul_same_base with >= 2^18 coefficients validates the shared base once.  bn256's UnmarshalBinary
    checks the curve equation only (point.go:466-499, twist.go:49-60), so "decoded" must not become "vouched for the
    subgroup" (which selects the GLS walk): on a base with a cofactor component the result must still be the
    reference's plain double-and-add, whatever n is and however the batch is sharded (round-2 advisor finding)."""
    q = None
    x0 = 1
    while q is None:
        x = (x0, 1)
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_mul(x, x), x), O.TWIST_B))
        if y is not None:
            q = (x, y)
        x0 += 1
    assert O.g2_mul(O.ORDER, q) is not None
    base = O.g2_marshal(q)
    n = (1 << 18) + 3
    rng = random.Random(11)
    ks = [rng.randrange(O.ORDER) for _ in range(6)]
    k = np.zeros((n, 32), dtype=np.uint8)
    k[:, 31] = 1
    pos = [0, 1, n // 2, n - 2, n - 1, 77777]
    for p, v in zip(pos, ks):
        k[p] = np.frombuffer(v.to_bytes(32, "big"), dtype=np.uint8)
    out, st = bn.g2_commit(k, base)              # host buffers: the entry point the shortcut lives in
    assert not np.asarray(st).any()
    for p, v in zip(pos, ks):
        assert bytes(out[p]) == O.g2_marshal(O.g2_mul(v, q)), p
    assert bytes(out[5]) == base


def test_validate_pairing_product_form_with_the_zero_miller_fallback():
    """Round 4: bn256's ValidatePairing runs the product form (one final exponentiation) and hands the lanes whose joint
    Miller value was zero -- a G2 operand with a component of order 13 -- to the two-pairing program: booleans equal the
    oracle's two pairings + Equal on ordinary pairs, on pairs where one or BOTH pairings are zero (0 == 0 is true in the
    reference), on G2 operands carrying a small-order component, scattered through a batch of several workgroups."""
    import random

    from kyber_amd.pairing import bn256 as bn
    from oracle import bn256 as ON

    rng = random.Random(17)
    h = 2 * ON.P - ON.ORDER
    while True:
        x = (rng.randrange(ON.P), rng.randrange(ON.P))
        y = ON.f2_sqrt(ON.f2_add(ON.f2_mul(ON.f2_sqr(x), x), ON.TWIST_B))
        if y is not None:
            break
    Q13 = ON.g2_mul(ON.ORDER * h // 13, (x, y))
    p1, q1 = ON.g1_mul(5, ON.G1_GEN), ON.g2_mul(7, ON.G2_GEN)
    special = [
        (p1, q1, ON.g1_mul(35, ON.G1_GEN), ON.G2_GEN),
        (p1, q1, ON.g1_mul(36, ON.G1_GEN), ON.G2_GEN),
        (p1, Q13, ON.g1_mul(35, ON.G1_GEN), ON.G2_GEN),
        (p1, q1, ON.g1_mul(35, ON.G1_GEN), Q13),
        (p1, Q13, ON.g1_mul(3, ON.G1_GEN), ON.g2_mul(2, Q13)),
        (p1, ON.g2_add(q1, Q13), ON.g1_mul(7, ON.G1_GEN), ON.g2_add(ON.g2_mul(5, ON.G2_GEN), Q13)),
        (None, Q13, ON.g1_mul(3, ON.G1_GEN), ON.g2_mul(4, Q13)),   # pair A dead (infinity), pair B's value zero: 1 == 0 false
    ]
    n = 300
    a, b = ON.g1_marshal(p1), ON.g2_marshal(q1)
    c, d = ON.g1_marshal(ON.g1_mul(35, ON.G1_GEN)), ON.g2_marshal(ON.G2_GEN)
    P1, P2, I1, I2 = [a] * n, [b] * n, [c] * n, [d] * n
    want = [True] * n
    for j, (pa, qa, pb, qb) in enumerate(special):
        for pos in (j, 64 + 9 * j, n - 1 - j):
            P1[pos], P2[pos], I1[pos], I2[pos] = ON.g1_marshal(pa), ON.g2_marshal(qa), ON.g1_marshal(pb), ON.g2_marshal(qb)
            want[pos] = ON.validate_pairing(pa, qa, pb, qb)
    ok, st = bn.batch_validate_pairing(b"".join(P1), b"".join(P2), b"".join(I1), b"".join(I2))
    assert not np.asarray(st).any()
    assert [bool(v) for v in np.asarray(ok)] == want
    assert want[4] and want[64 + 36] and not want[2]  # (the both-zero case IS true in the reference)


@pytest.mark.parametrize("n", [(1 << 17) + 5, (1 << 18) + 77, (1 << 19) + 1])
def test_hash_g1_large_batches_take_the_queued_kernel_and_agree_with_the_per_lane_one(bn, n):
    """pointG1.Hash (try-and-increment, point.go:261-313) for n >= 2^17 runs with the pending candidates of 128 / 256 / 512
    messages queued per wave in LDS (bn256.hip bn256_hash_g1_queue_kernel); smaller batches keep one message per lane.
    Same bytes either way: the large batch against itself hashed in 2^16-message pieces, a ragged last workgroup
    included, and first / last / strided messages against the oracle."""
    import hashlib

    import torch

    msgs = np.frombuffer(hashlib.shake_256(b"bn256/hash/queue/%d" % n).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()
    d = torch.from_numpy(msgs).cuda()
    out, st = bn.batch_hash_g1(d)
    assert not st.any().item()
    step = 1 << 16
    for lo in range(0, n, step):
        o2, s2 = bn.batch_hash_g1(d[lo:lo + step].contiguous())
        assert not s2.any().item() and torch.equal(out[lo:lo + step], o2), lo
    got = out.cpu().numpy()
    for i in [0, 1, n - 2, n - 1] + list(range(4099, n - 2, n // 13)):
        assert bytes(got[i]) == O.g1_marshal(O.hash_to_g1(bytes(msgs[i]))), i


def test_hash_g1_svdw_known_hashes_and_batch(bn, G):
    """bn256.HashG1 (pairing/bn256/hash.go:10-110) on the engine: the 11 outputs of hash_test.go:45-57 one by one and as
    one batch, a 3000-message batch against the oracle on a stride, domain separation tags of every HMAC key class, host and
    device buffers"""
    import torch

    msgs = [bytes.fromhex(v["msg_hex"]) for v in G["hash_g1_svdw"]]
    out, st = bn.batch_hash_g1_svdw(msgs)
    assert not np.asarray(st).any()
    for i, v in enumerate(G["hash_g1_svdw"]):
        assert bytes(out[i]).hex() == v["point"], i
        assert bn.HashG1(msgs[i]).MarshalBinary().hex() == v["point"]
    n = 3000
    m = np.frombuffer(hashlib.shake_256(b"bn256/svdw").digest(n * 40), dtype=np.uint8).reshape(n, 40).copy()
    for dst in (b"", b"x", b"kyberhip-test-tag", bytes(range(64)), bytes(range(65)), bytes(255)):
        out_h, st_h = bn.batch_hash_g1_svdw(m, dst)
        out_d, st_d = bn.batch_hash_g1_svdw(torch.from_numpy(m).cuda(), dst)
        assert not np.asarray(st_h).any() and not st_d.any().item() and (out_d.cpu().numpy() == out_h).all()
        for i in list(range(0, n, 211)) + [n - 1]:
            assert bytes(out_h[i]) == O.g1_marshal(O.hash_g1_svdw(bytes(m[i]), dst)), (len(dst), i)
    # every output is a point of the group (cofactor 1: on the curve is in G1)
    pts, st = bn.g1_batch_unmarshal(out_h)
    assert not np.asarray(st).any() and (np.asarray(pts) == out_h).all()
