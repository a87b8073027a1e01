"""Host-side mirror of the hot-path callers in ``share/poly.go`` over the batch engine:

  PriPoly.Commit   share/poly.go:143-149   t x Mul(coeffs[i], b)        -> ONE same-base batch call
  PubPoly.Eval     share/poly.go:340-348   Horner: t x (Mul + Add)      -> ONE MSM with scalars x^j
  PubPoly.Shares   share/poly.go:350-357   n x Eval                     -> ONE batched evaluation (poly_eval kernel)
  PubPoly.Check    share/poly.go:405-409   Eval + one Mul
  RecoverCommit    share/poly.go:449-476   Lagrange: t x (Mul + Add)    -> ONE MSM with the Lagrange coefficients
  RecoverPubPoly   share/poly.go:480-508   t x basis.Commit(y_j) + Adds: t^2 x (Mul + Add)
                                           -> t MSMs over the same t shares (coefficient k = sum_j L_j[k] * y_j)
  PubPoly.Add      share/poly.go:365-380   t x Add                      -> ONE batch add
  PriPoly.Shares   share/poly.go:96-102    n x Eval = n x t scalar (Mul + Add) -> ONE scalar-field Horner launch
  PriPoly.Add / Mul / Equal, RecoverSecret, RecoverPriPoly (poly.go:106-283): scalar-field only, host.

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
    from ..pairing import bls12381, bn256

    for m in (bls12381, bn256):
        if isinstance(pt, m.G1Elt):
            return (lambda s, b: m.ENGINE.mul(1, s, m.G1_BASE if b is None else b, True)[0]), m.g1_msm, m.G1_LEN
        if isinstance(pt, m.G2Elt):
            return (lambda s, b: m.ENGINE.mul(2, s, m.G2_BASE if b is None else b, True)[0]), m.g2_msm, m.G2_LEN
    raise TypeError("not an engine-backed group")


def _engine_backed(group) -> bool:
    """the group's scalars are one of the engine's mirrors (and not a host-only stand-in of the CPU tests)"""
    mod = type(group.Scalar()).__module__
    return mod.startswith("kyber_amd.group.edwards25519") or mod.startswith("kyber_amd.pairing")


def _add_op(group):
    """batch a[i] + b[i] over encoded points of the concrete group behind a kyber.Group mirror."""
    pt = group.Point()
    if type(pt).__module__.endswith("edwards25519"):
        from ..group import edwards25519 as ed

        return ed.batch_add
    from ..pairing import bls12381, bn256

    for m in (bls12381, bn256):
        if isinstance(pt, m.G1Elt):
            return m.g1_batch_add
        if isinstance(pt, m.G2Elt):
            return m.g2_batch_add
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

    # from this many indices one engine launch replaces the host loop (the single-operation policy of INTEGRATION.md 1a:
    # a device call costs one launch's latency whatever the batch)
    DEVICE_MIN = 64

    def EvalMany(self, indices) -> list:
        """[Eval(i) for i in indices] in ONE launch (kyb_<suite>_scalar_poly_eval: a lane per index runs the Horner
        loop of poly.go:85-93 over Montgomery residues modulo the group order)."""
        indices = list(indices)
        sc = self.g.Scalar()
        cb = b"".join(c.MarshalBinary() for c in self.coeffs)
        if type(sc).__module__.endswith("edwards25519"):
            from ..group import edwards25519 as ed

            out = ed.scalar_poly_eval(cb, indices)
        else:
            from ..pairing import bls12381, bn254, bn256

            for m in (bls12381, bn256, bn254):
                if isinstance(sc, m.Scalar):
                    out = m.ENGINE.scalar_poly_eval(cb, indices)
                    break
            else:
                raise TypeError("not an engine-backed group")
        return [PriShare(i, self.g.Scalar().UnmarshalBinary(bytes(row))) for i, row in zip(indices, out)]

    def Shares(self, n: int) -> list:  # poly.go:96-102
        # the batch kernel is an accelerator of a host-only scalar computation: WITHOUT a device (library missing, no gfx950:
        # KYB_E_NODEV = -3) the reference's own loop runs, and an empty polynomial never reaches the device.  Any other
        # failure -- a kernel fault, a poisoned context -- is the caller's to see: it is raised, not hidden behind an
        # O(n t) host loop.
        if n >= self.DEVICE_MIN and self.coeffs and _engine_backed(self.g):
            from .._lib import KyberHipError

            try:
                return self.EvalMany(range(n))
            except OSError:
                pass  # the shared library is not there
            except KyberHipError as e:
                no_device = ("rc=-3", "no ROCm-capable device", "no usable", "not found")  # KYB_E_NODEV, HIP's own wording, a missing library
                if not any(w in str(e) for w in no_device):
                    raise
        return [self.Eval(i) for i in range(n)]

    def Coefficients(self) -> list:  # poly.go:176-178
        return [c.Clone() for c in self.coeffs]

    def Add(self, q: "PriPoly") -> "PriPoly":  # poly.go:106-118
        if self.g.String() != q.g.String():
            raise ValueError("share: non-matching groups")
        if self.Threshold() != q.Threshold():
            raise ValueError("share: different number of coefficients")
        return PriPoly(self.g, [self.g.Scalar().Add(a, b) for a, b in zip(self.coeffs, q.coeffs)])

    def Mul(self, q: "PriPoly") -> "PriPoly":  # poly.go:156-172: plain convolution of the coefficients
        out = [self.g.Scalar().Zero() for _ in range(len(self.coeffs) + len(q.coeffs) - 1)]
        for i, a in enumerate(self.coeffs):
            for j, b in enumerate(q.coeffs):
                out[i + j].Add(out[i + j], self.g.Scalar().Mul(a, b))
        return PriPoly(self.g, out)

    def Equal(self, q: "PriPoly") -> bool:  # poly.go:124-139
        if self.g.String() != q.g.String() or len(self.coeffs) != len(q.coeffs):
            return False
        return all(a.MarshalBinary() == b.MarshalBinary() for a, b in zip(self.coeffs, q.coeffs))

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

    def Info(self):  # poly.go:325-327
        return self.b, self.commits

    def Add(self, q: "PubPoly") -> "PubPoly":
        """poly.go:365-380: component-wise sum, all t additions in one batch call; keeps self.b as the reference does."""
        if self.g.String() != q.g.String():
            raise ValueError("share: non-matching groups")
        if self.Threshold() != q.Threshold():
            raise ValueError("share: different number of coefficients")
        out, st = _add_op(self.g)(b"".join(c.MarshalBinary() for c in self.commits),
                                  b"".join(c.MarshalBinary() for c in q.commits))
        if st.any():
            raise ValueError("share: invalid commitment")
        return PubPoly(self.g, self.b, [type(self.g.Point())(bytes(row)) for row in out])

    def Equal(self, q: "PubPoly") -> bool:  # poly.go:386-402
        if self.g.String() != q.g.String() or self.Threshold() != q.Threshold():
            return False
        return all(a.MarshalBinary() == b.MarshalBinary() for a, b in zip(self.commits, q.commits))

    def Check(self, s: PriShare) -> bool:  # poly.go:405-409
        pv = self.Eval(s.I)
        ps = self.g.Point().Mul(s.V, self.b)
        return pv.V.Equal(ps)


def recover_commit(group, shares, t: int, n: int):
    """share.RecoverCommit (poly.go:449-476): p(0) = sum_i (prod_{j != i} x_j / (x_j - x_i)) * y_i over the
    first t shares by index (xyCommit poly.go:417-445), as one MSM with the Lagrange coefficients."""
    xs, ys = _xy_commit(group, shares, t)
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


def _xy_commit(group, shares, t: int):
    """xyCommit (poly.go:418-445): the first t usable public shares by index -> ({i: x_i}, {i: y_i})."""
    good = sorted((s for s in shares if s is not None and s.V is not None), key=lambda s: s.I)
    xs, ys = {}, {}
    for s in good:
        xs[s.I] = group.Scalar().SetInt64(s.I + 1)
        ys[s.I] = s.V
        if len(xs) == t:
            break
    return xs, ys


def _xy_scalar(group, shares, t: int):
    """xyScalar (poly.go:219-243): same selection for private shares."""
    return _xy_commit(group, shares, t)


def minus_const(group, c) -> PriPoly:
    """minusConst (poly.go:247-253): the polynomial x - c."""
    return PriPoly(group, [group.Scalar().Neg(c), group.Scalar().One()])


def lagrange_basis(group, i: int, xs: dict) -> PriPoly:
    """lagrangeBasis (poly.go:513-536): L_i(x) = prod_{m != i} (x - x_m) / (x_i - x_m) as t scalar coefficients."""
    basis = PriPoly(group, [group.Scalar().One()])
    acc = group.Scalar().One()
    for m, xm in xs.items():
        if m == i:
            continue
        basis = basis.Mul(minus_const(group, xm))
        acc.Mul(acc, group.Scalar().Inv(group.Scalar().Sub(xs[i], xm)))
    return PriPoly(group, [group.Scalar().Mul(c, acc) for c in basis.coeffs])


def recover_secret(group, shares, t: int, n: int):
    """share.RecoverSecret (poly.go:182-208): p(0) by Lagrange interpolation over the scalar field (host)."""
    xs, ys = _xy_scalar(group, shares, t)
    if len(xs) < t:
        raise ValueError("share: not enough shares to recover secret")
    acc = group.Scalar().Zero()
    for i, xi in xs.items():
        num, den = ys[i].Clone(), group.Scalar().One()
        for j, xj in xs.items():
            if i == j:
                continue
            num.Mul(num, xj)
            den.Mul(den, group.Scalar().Sub(xj, xi))
        acc.Add(acc, group.Scalar().Div(num, den))
    return acc


def recover_pri_poly(group, shares, t: int, n: int) -> PriPoly:
    """share.RecoverPriPoly (poly.go:260-283): sum_j y_j * L_j(x) over the scalar field (host)."""
    xs, ys = _xy_scalar(group, shares, t)
    if len(xs) != t:
        raise ValueError("share: not enough shares to recover private polynomial")
    acc = None
    for j in xs:
        term = PriPoly(group, [group.Scalar().Mul(c, ys[j]) for c in lagrange_basis(group, j, xs).coeffs])
        acc = term if acc is None else acc.Add(term)
    return acc


def recover_pub_poly(group, shares, t: int, n: int) -> PubPoly:
    """share.RecoverPubPoly (poly.go:480-508).  The reference commits every Lagrange basis polynomial to its share
    (t x t Mul) and adds the t polynomials (t x t Add); coefficient k of the result is sum_j L_j[k] * y_j, i.e. t
    MSMs over the SAME t points with the k-th coefficients of the basis polynomials as scalars.  The base point of
    the result is None, as in the reference (basis.Commit(y_j) leaves accPoly.b = y_0's role undefined there too)."""
    xs, ys = _xy_commit(group, shares, t)
    if len(xs) < t:
        raise ValueError("share: not enough good public shares to reconstruct secret commitment")
    _, msm, _ = _ops(group)
    order = list(xs)
    basis = [lagrange_basis(group, j, xs).coeffs for j in order]  # basis[j][k]
    pts = b"".join(ys[j].MarshalBinary() for j in order)
    commits = []
    for k in range(t):
        out, st = msm(b"".join(basis[j][k].MarshalBinary() for j in range(t)), pts)
        if st.any():
            raise ValueError("share: invalid public share")
        commits.append(type(group.Point())(bytes(out)))
    return PubPoly(group, None, commits)
