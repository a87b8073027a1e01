"""Host-side mirror of the hot-path callers in ``share/poly.go`` over the batch engine:

  PriPoly.Commit   share/poly.go:143-149   t x Mul(coeffs[i], b)        -> ONE same-base batch call
  PubPoly.Eval     share/poly.go:340-348   Horner: t x (Mul + Add)      -> ONE MSM with scalars x^j
  PubPoly.Shares   share/poly.go:350-357   n x Eval                     -> ONE batched evaluation (poly_eval kernel)
  PubPoly.Check    share/poly.go:405-409   Eval + one Mul
  RecoverCommit    share/poly.go:449-476   Lagrange: t x (Mul + Add)    -> ONE MSM with the Lagrange coefficients

``group`` is any engine-backed kyber.Group mirror (``edwards25519.NewSuite()``, ``bls12381.NewSuite().G1()``,
``bn256.NewSuite().G2()`` ...).  Scalar arithmetic stays on the host (as group/mod does in the reference); every
point operation is an engine call.
"""
from __future__ import annotations


def _ops(group):
    """(batch same-base mul, msm, scalar -> wire bytes) for the concrete group behind a kyber.Group mirror."""
    pt = group.Point()
    mod = type(pt).__module__
    if mod.endswith("edwards25519"):
        from ..group import edwards25519 as ed

        return (lambda scalars, base: ed.commit(scalars, base)), ed.msm, 32
    import importlib

    from ..pairing import bls12381, bn256

    for m in (bls12381, bn256):
        if isinstance(pt, m.G1Elt):
            return (lambda s, b: m.ENGINE.mul(1, s, m.G1_BASE if b is None else b, True)[0]), m.g1_msm, m.G1_LEN
        if isinstance(pt, m.G2Elt):
            return (lambda s, b: m.ENGINE.mul(2, s, m.G2_BASE if b is None else b, True)[0]), m.g2_msm, m.G2_LEN
    raise TypeError("not an engine-backed group")


class PubShare:
    def __init__(self, i: int, v):
        self.I, self.V = i, v


class PriShare:
    def __init__(self, i: int, v):
        self.I, self.V = i, v


class PriPoly:
    """share.PriPoly (poly.go:37-44): coefficients a_0 .. a_{t-1} (kyber.Scalar mirrors)."""

    def __init__(self, group, coeffs):
        self.g, self.coeffs = group, list(coeffs)

    @classmethod
    def new(cls, group, t: int, secret=None, rand=None):
        coeffs = [group.Scalar().Pick(rand) for _ in range(t)]
        if secret is not None:
            coeffs[0] = secret
        return cls(group, coeffs)

    def Threshold(self) -> int:
        return len(self.coeffs)

    def Eval(self, i: int) -> PriShare:  # poly.go:85-93 (scalar Horner, host)
        xi = self.g.Scalar().SetInt64(1 + i)
        v = self.g.Scalar().Zero()
        for c in reversed(self.coeffs):
            v.Mul(v, xi)
            v.Add(v, c)
        return PriShare(i, v)

    def Commit(self, b=None) -> "PubPoly":
        """poly.go:143-149 -- all t commitments in one same-base batch."""
        mul_same_base, _, _ = _ops(self.g)
        scal = b"".join(c.MarshalBinary() if not hasattr(c, "v") or not isinstance(c.v, bytes) else c.v for c in self.coeffs)
        out = mul_same_base(scal, None if b is None else b.MarshalBinary())
        commits = [type(self.g.Point())(bytes(row)) for row in out]
        return PubPoly(self.g, b, commits)


class PubPoly:
    """share.PubPoly (poly.go:289-294): commitments A_j = a_j * b."""

    def __init__(self, group, b, commits):
        self.g, self.b, self.commits = group, b, list(commits)

    def Threshold(self) -> int:
        return len(self.commits)

    def Commit(self):
        return self.commits[0].Clone()

    def Eval(self, i: int) -> PubShare:
        """poly.go:340-348: v = sum_j x^j A_j with x = i + 1, as one MSM."""
        _, msm, _ = _ops(self.g)
        x = self.g.Scalar().SetInt64(1 + i)
        pw = self.g.Scalar().One()
        scal = []
        for _ in self.commits:
            scal.append(pw.MarshalBinary())
            pw = self.g.Scalar().Mul(pw, x)
        out, st = msm(b"".join(scal), b"".join(c.MarshalBinary() for c in self.commits))
        if st.any():
            raise ValueError("share: invalid commitment")
        return PubShare(i, type(self.g.Point())(bytes(out)))

    def EvalMany(self, indices) -> list:
        """[Eval(i) for i in indices] in ONE launch: a lane per index runs the reference's Horner loop
        (kyb_*_poly_eval).  PubPoly.Shares(n) (poly.go:350-357) and the per-participant checks of DKG / VSS have
        this shape."""
        pt = self.g.Point()
        cb = b"".join(c.MarshalBinary() for c in self.commits)
        if type(pt).__module__.endswith("edwards25519"):
            from ..group import edwards25519 as ed

            out, st = ed.poly_eval(cb, list(indices))
        else:
            from ..pairing import bls12381, bn256

            for m in (bls12381, bn256):
                if isinstance(pt, (m.G1Elt, m.G2Elt)):
                    out, st = m.ENGINE.poly_eval(1 if isinstance(pt, m.G1Elt) else 2, cb, list(indices))
                    break
            else:
                raise TypeError("not an engine-backed group")
        if st.any():
            raise ValueError("share: invalid commitment")
        return [PubShare(i, type(pt)(bytes(row))) for i, row in zip(indices, out)]

    def Shares(self, n: int) -> list:  # poly.go:350-357
        return self.EvalMany(range(n))

    def Check(self, s: PriShare) -> bool:  # poly.go:405-409
        pv = self.Eval(s.I)
        ps = self.g.Point().Mul(s.V, self.b)
        return pv.V.Equal(ps)


def recover_commit(group, shares, t: int, n: int):
    """share.RecoverCommit (poly.go:449-476): p(0) = sum_i (prod_{j != i} x_j / (x_j - x_i)) * y_i over the
    first t shares by index (xyCommit poly.go:417-445), as one MSM with the Lagrange coefficients."""
    good = sorted((s for s in shares if s is not None and s.V is not None), key=lambda s: s.I)
    xs, ys = {}, {}
    for s in good:
        xs[s.I] = group.Scalar().SetInt64(s.I + 1)
        ys[s.I] = s.V
        if len(xs) == t:
            break
    if len(xs) < t:
        raise ValueError("share: not enough good public shares to reconstruct secret commitment")
    _, msm, _ = _ops(group)
    scal, pts = [], []
    for i, xi in xs.items():
        num, den = group.Scalar().One(), group.Scalar().One()
        for j, xj in xs.items():
            if i == j:
                continue
            num.Mul(num, xj)
            den.Mul(den, group.Scalar().Sub(xj, xi))
        scal.append(group.Scalar().Div(num, den).MarshalBinary())
        pts.append(ys[i].MarshalBinary())
    out, st = msm(b"".join(scal), b"".join(pts))
    if st.any():
        raise ValueError("share: invalid public share")
    return type(group.Point())(bytes(out))
