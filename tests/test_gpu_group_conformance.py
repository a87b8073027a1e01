"""The reference's generic group conformance test (util/test/test.go:325-400 testGroup, with
testSanityCheck :150-195, testHomomorphicIdentities :197-254, testRandomlyPickedPoint :256-292,
testEncodingDecoding :294-322) replayed against the engine-backed kyber.Group mirrors: Ed25519 and
G1 / G2 of both pairing suites.  Every point operation below is a call into the HIP engine."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu


def _groups():
    from kyber_amd.group import edwards25519 as ed
    from kyber_amd.pairing import bls12381 as bls, bn256 as bn

    return {
        "Ed25519": ed.NewSuite(),
        "bls12381.G1": bls.NewSuite().G1(), "bls12381.G2": bls.NewSuite().G2(),
        "bn256.G1": bn.NewSuite().G1(), "bn256.G2": bn.NewSuite().G2(),
    }


class _Stream:
    """Deterministic stand-in for the cipher.Stream the reference threads through testGroup."""

    def __init__(self, seed: bytes):
        self.x = hashlib.shake_256(seed)
        self.off = 0

    def __call__(self, n: int) -> bytes:
        out = self.x.digest(self.off + n)[self.off:]
        self.off += n
        return out


@pytest.mark.parametrize("name", ["Ed25519", "bls12381.G1", "bls12381.G2", "bn256.G1", "bn256.G2"])
def test_group(name):
    g = _groups()[name]
    rand = _Stream(name.encode())
    ptmp, stmp = g.Point(), g.Scalar()
    pzero, szero, sone = g.Point().Null(), g.Scalar().Zero(), g.Scalar().One()
    s1, s2 = g.Scalar().Pick(rand), g.Scalar().Pick(rand)
    assert not s1.Equal(szero) and not s2.Equal(szero) and not s1.Equal(s2)
    gen = g.Point().Base()
    # testSanityCheck: 2G = G + G, identity, DH commutativity
    stmp.SetInt64(2)
    pt2 = g.Point().Mul(stmp, gen)
    assert g.Point().Add(gen, gen).Equal(pt2)
    assert g.Point().Add(gen, pzero).Equal(gen)
    assert g.Point().Mul(sone, None).Equal(gen)  # nil point means the base (group.go:128-130)
    p1, p2 = g.Point().Mul(s1, gen), g.Point().Mul(s2, gen)
    assert not p1.Equal(p2)
    dh1, dh2 = g.Point().Mul(s2, p1), g.Point().Mul(s1, p2)
    assert dh1.Equal(dh2)
    # scalar inverse
    assert ptmp.Mul(g.Scalar().Inv(s2), dh1).Equal(p1)
    # zero and one
    assert ptmp.Mul(szero, dh1).Equal(pzero)
    assert ptmp.Mul(sone, dh1).Equal(dh1)
    # testHomomorphicIdentities
    ptmp.Add(p1, p2)
    assert g.Point().Mul(g.Scalar().Add(s1, s2), gen).Equal(ptmp)
    ptmp.Sub(p1, p2)
    assert g.Point().Mul(g.Scalar().Sub(s1, s2), gen).Equal(ptmp)
    assert g.Point().Add(g.Point().Neg(p2), p1).Equal(ptmp)
    assert ptmp.Mul(g.Scalar().Mul(s1, s2), gen).Equal(dh1)
    assert g.Scalar().Div(g.Scalar().Mul(s1, s2), s2).Equal(s1)
    # testRandomlyPickedPoint
    last = gen
    for _ in range(3):
        rgen = g.Point().Pick(rand)
        assert not rgen.Equal(last)
        last = rgen
        assert ptmp.Mul(stmp.SetInt64(-1), rgen).Add(ptmp, rgen).Equal(pzero)
        ptmp.Mul(stmp.SetInt64(2), rgen)
        assert ptmp.Mul(g.Scalar().Inv(stmp), ptmp).Equal(rgen)
        assert g.Point().Neg(g.Point().Neg(rgen)).Equal(rgen)
    # testEncodingDecoding + null point round trip
    for _ in range(3):
        s = g.Scalar().Pick(rand)
        assert g.Scalar().UnmarshalBinary(s.MarshalBinary()).Equal(s)
        p = g.Point().Pick(rand)
        assert g.Point().UnmarshalBinary(p.MarshalBinary()).Equal(p)
    assert g.Point().UnmarshalBinary(pzero.MarshalBinary()).Equal(pzero)
    # testPointSet / testPointClone
    q = g.Point().Set(p1)
    assert q.Equal(p1)
    q.Add(q, gen)
    assert not q.Equal(p1)
    c = p1.Clone()
    assert c.Equal(p1)
    c.Null()
    assert not c.Equal(p1)
    # a wrong concrete type panics in the reference (ErrTypeCast) -> TypeError here
    with pytest.raises(TypeError):
        p1.Equal(object())
