"""The reference's hot-path CALLERS replayed over the engine:
sign/bls Sign / Verify, sign/bdn aggregation against TestBDNFixtures, share/poly Commit / Eval / Check /
RecoverCommit against the sequential Mul + Add the reference performs."""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(golden_dir):
    return json.load(open(os.path.join(golden_dir, "bn256.json")))


def test_bls_sign_and_batch_verify_bn256_fixtures(G):
    from kyber_amd.sign import bls

    sch = bls.NewSchemeOnG1_bn256()
    msg = G["bdn_msg"].encode()
    sigs = [sch.sign(bytes.fromhex(p), msg) for p in G["bdn_privs"]]
    assert [s.hex() for s in sigs] == G["bdn_sigs"]  # TestBDNFixtures sig1..3 (bdn_vartime_test.go:104-118)
    pubs = [bytes.fromhex(p) for p in G["bdn_pubs"]]
    ok = sch.batch_verify(pubs, [msg] * 3, sigs)
    assert ok.all()
    ok = sch.batch_verify(pubs, [msg, b"other message", msg], [sigs[0], sigs[1], sigs[0]])
    assert list(ok) == [True, False, False]
    ok = sch.batch_verify(pubs[:1], [msg], [bytes(63) + b"\x05"])  # undecodable signature
    assert not ok[0]


def test_bdn_aggregation_fixtures(G):
    from kyber_amd.pairing import bn256
    from kyber_amd.sign import bdn, bls

    sch = bdn.Scheme(bn256)
    pubs = [bytes.fromhex(p) for p in G["bdn_pubs"]]
    sigs = [bytes.fromhex(s) for s in G["bdn_sigs"]]
    mask = [True, False, True]  # bdn_vartime_test.go:120-123
    agg_sig = sch.aggregate_signatures([sigs[0], sigs[2]], pubs, mask)
    agg_key = sch.aggregate_public_keys(pubs, mask)
    assert agg_sig.hex() == G["bdn_fixture_agg_sig_mask101"]
    assert agg_key.hex() == G["bdn_fixture_agg_key_mask101"]
    # and the aggregate verifies as a plain BLS signature under the aggregate key (bdn.go:107-124)
    assert bls.NewSchemeOnG1_bn256().verify(agg_key, G["bdn_msg"].encode(), agg_sig)
    # TestBDN_HashPointToR_BN256: P, 2P, 3P with all bits set
    P, _ = bn256.g2_commit(b"".join((i + 1).to_bytes(32, "big") for i in range(3)))
    assert sch.aggregate_public_keys([bytes(p) for p in P], [True] * 3).hex() == G["bdn_agg_key"]


def _groups():
    from kyber_amd.group import edwards25519 as ed
    from kyber_amd.pairing import bls12381 as bls, bn256 as bn

    return {"Ed25519": ed.NewSuite(), "bls12381.G1": bls.NewSuite().G1(), "bn256.G2": bn.NewSuite().G2()}


@pytest.mark.parametrize("name", ["Ed25519", "bls12381.G1", "bn256.G2"])
def test_share_poly_commit_eval_recover(name):
    """poly_test.go style: commitments of a degree t-1 polynomial; public shares evaluate consistently with
    private shares; any t public shares recover the secret commitment."""
    from kyber_amd.share import poly

    g = _groups()[name]
    rng = random.Random(7)
    rand = lambda n: bytes(rng.randrange(256) for _ in range(n))
    t, n = 5, 9
    pri = poly.PriPoly.new(g, t, rand=rand)
    pub = pri.Commit(None)
    assert pub.Threshold() == t
    # commitments are coeffs[i] * B (poly.go:146), one by one through Point.Mul
    for c, A in zip(pri.coeffs, pub.commits):
        assert g.Point().Mul(c, None).Equal(A)
    # sequential Horner of the reference (poly.go:340-348) == the MSM result
    for i in (0, 3, n - 1):
        x = g.Scalar().SetInt64(1 + i)
        v = g.Point().Null()
        for A in reversed(pub.commits):
            v = g.Point().Add(g.Point().Mul(x, v), A)
        assert pub.Eval(i).V.Equal(v)
        assert pub.Check(pri.Eval(i))
    shares = [pub.Eval(i) for i in range(n)]
    shares[1] = None
    rng.shuffle(shares)
    assert poly.recover_commit(g, shares, t, n).Equal(pub.Commit())
    with pytest.raises(ValueError):
        poly.recover_commit(g, shares[:2], t, n)
    # a custom base point
    b = g.Point().Pick(rand)
    pub_b = pri.Commit(b)
    assert pub_b.commits[2].Equal(g.Point().Mul(pri.coeffs[2], b))
    assert pub_b.Check(pri.Eval(4))


def test_tbls_sign_recover_verify_bn256():
    """internal/test/threshold.go shape: t-of-n shares sign, any t partials recover the signature the
    secret itself would produce, a corrupted partial is skipped."""
    from kyber_amd.pairing import bn256
    from kyber_amd.share import poly
    from kyber_amd.sign import bls, tbls

    rng = random.Random(11)
    rand = lambda n: bytes(rng.randrange(256) for _ in range(n))
    suite = bn256.NewSuite()
    t, n = 3, 5
    pri = poly.PriPoly.new(suite.G2(), t, rand=rand)
    pub = pri.Commit(None)  # public sharing polynomial on the key group G2
    sch = tbls.NewThresholdSchemeOnG1_bn256()
    msg = b"Hello threshold Boneh-Lynn-Shacham"
    partials = [sch.sign(pri.Eval(i), msg) for i in range(n)]
    assert all(sch.verify_partial(pub, msg, s) for s in partials)
    partials[1] = partials[1][:10] + bytes([partials[1][10] ^ 1]) + partials[1][11:]  # corrupt one
    sig = sch.recover(pub, msg, partials, t, n)
    plain = bls.NewSchemeOnG1_bn256()
    assert sig == plain.sign(pri.coeffs[0].MarshalBinary(), msg)
    assert plain.verify(pub.Commit().MarshalBinary(), msg, sig)
    with pytest.raises(ValueError):
        sch.recover(pub, msg, partials[:2], t, n)


def test_eddsa_wycheproof_vectors_through_the_engine(golden_dir):
    """sign/eddsa/eddsa_test.go:355 TestWycheProof: all 150 cases, result must equal the vector's verdict."""
    from kyber_amd.sign import eddsa

    M = json.load(open(os.path.join(golden_dir, "ed25519_misc.json")))
    W = M["wycheproof"]
    ok = eddsa.batch_verify_with_checks([bytes.fromhex(c["pk"]) for c in W], [bytes.fromhex(c["msg"]) for c in W],
                                        [bytes.fromhex(c["sig"]) for c in W])
    bad = [c["id"] for c, o in zip(W, ok) if bool(o) != c["valid"]]
    assert not bad, bad
    # RFC 8032 vectors (eddsa_test.go:24-52) verify too
    R8 = M["rfc8032"]
    ok = eddsa.batch_verify_with_checks([bytes.fromhex(c["pub"]) for c in R8], [bytes.fromhex(c["msg"]) for c in R8],
                                        [bytes.fromhex(c["sig"]) for c in R8])
    assert ok.all()


def test_batch_add_all_groups():
    from kyber_amd.group import edwards25519 as ed
    from kyber_amd.pairing import bls12381 as bls, bn256 as bn
    from oracle import bls12381 as OB, bn256 as ON, ed25519 as O

    rng = random.Random(21)
    # Ed25519: incl. identity, doubling (a == b) and inverse (a + (-a))
    ks = [rng.randrange(1, O.L) for _ in range(6)]
    pts = [O.mul_int(k, O.B) for k in ks]
    a = [pts[0], pts[1], pts[2], pts[3], O.IDENTITY, pts[4]]
    b = [pts[1], pts[1], O.neg(pts[2]), O.IDENTITY, pts[5], pts[5]]
    out, st = ed.batch_add(b"".join(O.encode(p) for p in a), b"".join(O.encode(p) for p in b))
    assert not st.any()
    for i in range(6):
        assert bytes(out[i]) == O.encode(O.add(a[i], b[i])), i
    out, st = ed.batch_add(O.encode(pts[0]), bytes([2]) + bytes(31))
    assert st[0] == 1 and not out.any()
    for m, OR, e1, e2, order in ((bls, OB, OB.g1_compress, OB.g2_compress, OB.R), (bn, ON, ON.g1_marshal, ON.g2_marshal, ON.ORDER)):
        ks = [rng.randrange(1, order) for _ in range(4)]
        p1 = [OR.g1_mul(k, OR.G1_GEN) for k in ks]
        p2 = [OR.g2_mul(k, OR.G2_GEN) for k in ks]
        a1, b1 = [p1[0], p1[1], p1[2], None], [p1[1], p1[1], OR.g1_neg(p1[2]), p1[3]]
        a2, b2 = [p2[0], p2[1], p2[2], None], [p2[1], p2[1], OR.g2_neg(p2[2]), p2[3]]
        out, st = m.g1_batch_add(b"".join(e1(p) for p in a1), b"".join(e1(p) for p in b1))
        assert not st.any()
        for i in range(4):
            assert bytes(out[i]) == e1(OR.g1_add(a1[i], b1[i])), (m.__name__, i)
        out, st = m.g2_batch_add(b"".join(e2(p) for p in a2), b"".join(e2(p) for p in b2))
        assert not st.any()
        for i in range(4):
            assert bytes(out[i]) == e2(OR.g2_add(a2[i], b2[i])), (m.__name__, i)


@pytest.mark.parametrize("name", ["Ed25519", "bls12381.G1", "bls12381.G2", "bn256.G1", "bn256.G2"])
def test_pubpoly_batched_eval_matches_per_index_eval(name):
    """PubPoly.EvalMany / Shares (one kyb_*_poly_eval launch: a lane per index runs the reference's Horner loop) ==
    PubPoly.Eval per index (one MSM each) == the private shares times the base; indices at the edges of the 32-bit
    range; t = 1 and an undecodable commitment."""
    from kyber_amd.group import edwards25519 as ed
    from kyber_amd.pairing import bls12381 as bls, bn256 as bn
    from kyber_amd.share import poly

    g = {"Ed25519": ed.NewSuite(), "bls12381.G1": bls.NewSuite().G1(), "bls12381.G2": bls.NewSuite().G2(),
         "bn256.G1": bn.NewSuite().G1(), "bn256.G2": bn.NewSuite().G2()}[name]
    rng = random.Random(17)
    rand = lambda n: bytes(rng.randrange(256) for _ in range(n))
    t = 7
    pri = poly.PriPoly.new(g, t, rand=rand)
    pub = pri.Commit(None)
    idx = [0, 1, 2, 5, 63, 64, 255, 65535, 65536, (1 << 31) - 1, (1 << 32) - 2, (1 << 32) - 1] + [rng.randrange(1 << 32) for _ in range(20)]
    many = pub.EvalMany(idx)
    assert [s.I for s in many] == idx
    for s in many[:6] + many[9:14]:
        assert s.V.Equal(pub.Eval(s.I).V), s.I
    for s in many[6:9]:
        assert s.V.Equal(g.Point().Mul(pri.Eval(s.I).V, None)), s.I
    shares = pub.Shares(9)
    assert all(sh.V.Equal(pub.Eval(i).V) for i, sh in enumerate(shares))
    # constant polynomial: every evaluation is the commitment itself
    one = poly.PubPoly(g, None, pub.commits[:1])
    assert all(s.V.Equal(pub.commits[0]) for s in one.EvalMany([0, 7, (1 << 32) - 1]))
    # an undecodable commitment is reported (the reference fails in UnmarshalBinary before it gets this far)
    bad = list(pub.commits)
    bad[3] = type(g.Point())(b"\x02" + bytes(g.PointLen() - 1) if name == "Ed25519" else b"\x05" * g.PointLen())
    with pytest.raises(ValueError):
        poly.PubPoly(g, None, bad).EvalMany([1, 2])


@pytest.mark.parametrize("name", ["Ed25519", "bls12381.G1", "bn256.G2"])
def test_share_recover_pub_poly_and_pubpoly_add(name):
    """RecoverPubPoly (poly.go:480-508) as t MSMs over the same shares gives back every commitment; PubPoly.Add
    (poly.go:365-380, one batch add) commutes with PriPoly.Add; the recovered polynomial evaluates like the original."""
    from kyber_amd.share import poly

    g = _groups()[name]
    rng = random.Random(23)
    rand = lambda n: bytes(rng.randrange(256) for _ in range(n))
    t, n = 4, 8
    pri, q = poly.PriPoly.new(g, t, rand=rand), poly.PriPoly.new(g, t, rand=rand)
    pub, qpub = pri.Commit(None), q.Commit(None)
    shares = pub.Shares(n)
    shares[0] = None
    shares[5] = None
    rng.shuffle(shares)
    rec = poly.recover_pub_poly(g, shares, t, n)
    assert rec.Threshold() == t and rec.Equal(pub)
    assert rec.Eval(6).V.Equal(pub.Eval(6).V)
    with pytest.raises(ValueError):
        poly.recover_pub_poly(g, [s for s in shares if s is not None][:t - 1], t, n)
    s = pub.Add(qpub)
    assert s.Equal(pri.Add(q).Commit(None)) and not s.Equal(pub)
    for a, b, c in zip(pub.commits, qpub.commits, s.commits):
        assert g.Point().Add(a, b).Equal(c)
    # scalar side against the same polynomial
    assert poly.recover_secret(g, pri.Shares(n)[2:], t, n).Equal(pri.coeffs[0])
    assert poly.recover_pri_poly(g, pri.Shares(n)[3:3 + t], t, n).Equal(pri)


def test_bn256_wrong_length_signature_fails_alone():
    """Generic SchemeOnG1.batch_verify (bn256): one malformed-length signature is one false entry, not a shifted batch
    or a ValueError for everybody."""
    from kyber_amd.pairing import bn256 as bn
    from kyber_amd.sign import bls as sbls

    sch = sbls.NewSchemeOnG1_bn256()
    n = 6
    xs = [((i + 2) * 0x9E3779B97F4A7C15F39CC0605CEDC835 % bn.ORDER).to_bytes(32, "big") for i in range(n)]
    pubs = [bytes(bn.g2_commit(x)[0][0]) for x in xs]
    msgs = [b"bn-msg-%d" % i for i in range(n)]
    sigs = [sch.sign(xs[i], msgs[i]) for i in range(n)]
    assert sch.batch_verify(pubs, msgs, sigs).all()
    bad = list(sigs)
    bad[1] = sigs[1][:-1]
    bad[2] = sigs[2] + b"\x07"
    assert list(sch.batch_verify(pubs, msgs, bad)) == [True, False, False, True, True, True]
    with pytest.raises(ValueError):
        sch.batch_verify(pubs, msgs, sigs[:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("suite", ["bls12381", "bn256"])
def test_g1_poly_eval_at_2p16_against_the_oracle(suite):
    """kyb_<suite>_g1_poly_eval at n = 2^16 indices held DIRECTLY to the oracle (until now GPU vs GPU
    plus host mirror vs oracle): commitments C_j = c_j G with known c_j, so lane i must hold (sum_j c_j (i + 1)^j mod r) G
    -- big-integer Horner + the oracle's own scalar multiplication and encoding on first / last / strided lanes."""
    import importlib

    m = importlib.import_module("kyber_amd.pairing." + suite)
    OR = importlib.import_module("oracle." + suite)
    order = OR.R if suite == "bls12381" else OR.ORDER
    enc = OR.g1_compress if suite == "bls12381" else OR.g1_marshal
    rng = random.Random(91)
    t, n = 5, 1 << 16
    cs = [rng.randrange(order) for _ in range(t)]
    commits = b"".join(enc(OR.g1_mul(c, OR.G1_GEN)) for c in cs)
    idx = np.arange(n, dtype=np.uint32)
    idx[-1] = (1 << 32) - 1  # the top of the index range on the last lane
    out, st = m.ENGINE.poly_eval(1, commits, idx)
    assert not st.any() and out.shape == (n, m.G1_LEN)
    lanes = [0, 1, 2, n - 2, n - 1] + list(range(977, n - 2, n // 29))
    assert len(lanes) >= 32
    for lane in lanes:
        x = int(idx[lane]) + 1
        v = 0
        for c in reversed(cs):
            v = (v * x + c) % order
        assert bytes(out[lane]) == enc(OR.g1_mul(v, OR.G1_GEN)), (suite, lane)


@pytest.mark.gpu
@pytest.mark.parametrize("suite", ["bls12381", "bn256"])
def test_g2_poly_eval_at_2p14_against_the_oracle(suite):
    """the same direct comparison on the KEY group of the signature schemes (tbls.Recover evaluates its public polynomial
    on G2: sign/tbls/tbls.go:124, share/poly.go:340-348): kyb_<suite>_g2_poly_eval at 2^14 indices, oracle lanes"""
    import importlib

    m = importlib.import_module("kyber_amd.pairing." + suite)
    OR = importlib.import_module("oracle." + suite)
    order = OR.R if suite == "bls12381" else OR.ORDER
    enc = OR.g2_compress if suite == "bls12381" else OR.g2_marshal
    rng = random.Random(92)
    t, n = 4, 1 << 14
    cs = [rng.randrange(order) for _ in range(t)]
    commits = b"".join(enc(OR.g2_mul(c, OR.G2_GEN)) for c in cs)
    idx = np.arange(n, dtype=np.uint32)
    idx[-1] = (1 << 32) - 1
    out, st = m.ENGINE.poly_eval(2, commits, idx)
    assert not st.any() and out.shape == (n, m.G2_LEN)
    lanes = [0, 1, n - 2, n - 1] + list(range(311, n - 2, n // 29))
    assert len(lanes) >= 32
    for lane in lanes:
        x = int(idx[lane]) + 1
        v = 0
        for c in reversed(cs):
            v = (v * x + c) % order
        assert bytes(out[lane]) == enc(OR.g2_mul(v, OR.G2_GEN)), (suite, lane)
