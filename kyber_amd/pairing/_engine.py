"""Shared host-side plumbing for the pairing suites (BLS12-381, bn256): batch entry points over the
C ABI and kyber.Scalar / kyber.Point / pairing.Suite mirrors holding canonical wire encodings.
No curve or field arithmetic happens in Python (scalars mod the group order are host plumbing,
like the reference's group/mod.Int)."""
from __future__ import annotations

import os

import numpy as np

from .._lib import check, load


# flags of the pairing-suite entry points (include/kyber_hip.h)
F_UNCOMPRESSED = 2
F_UNCOMPRESSED_OUT = 4
F_TRUSTED_ALL = 0xF00


# Batch sizes from which a device call beats one CPU core of the reference (tools/latency_probe.py on MI355X,
# profiles/r03_single_call_latency.json; the Go suite's MinDeviceBatch / MinDevicePairings): below them a caller with a
# CPU implementation at hand should use it.  tests/test_callers_host.py holds these equal to the Go suite's.
MIN_DEVICE_BATCH, MIN_DEVICE_PAIRINGS = 64, 8


def F_SCALAR_BITS(b: int) -> int:
    """*_msm only: every scalar is below 2^b (KYB_F_SCALAR_BITS); higher bits are ignored."""
    if not 1 <= b <= 256:
        raise ValueError("scalar bits must be in 1..256")
    return b << 16


def F_TRUSTED(i: int) -> int:
    """Point argument i was validated before (an output of this library / of an earlier UnmarshalBinary)."""
    return 0x100 << i


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _host(buf, width: int) -> np.ndarray:
    a = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray, memoryview)) else np.asarray(buf, dtype=np.uint8)
    return np.ascontiguousarray(a).reshape(-1, width)


def pack_fixed(items, width: int):
    """Pack a list of per-element encodings into an (n, width) array.  Elements of the wrong length are replaced by
    `width` zero bytes (which no suite decodes) and their indices returned, so that one malformed element neither
    shifts the elements after it nor makes the native call read past the buffer: the caller reports those lanes as
    ok = False / status = BAD_POINT, the way the reference's UnmarshalBinary rejects only the offending element."""
    n = len(items)
    out = np.zeros((n, width), dtype=np.uint8)
    bad = []
    for i, it in enumerate(items):
        b = bytes(it)
        if len(b) != width:
            bad.append(i)
        else:
            out[i] = np.frombuffer(b, dtype=np.uint8)
    return out, bad


def _stream():
    import torch

    return torch.cuda.current_stream().cuda_stream


class Engine:
    """Batch API of one pairing suite; `prefix` selects the kyb_<prefix>_* entry points."""

    def __init__(self, prefix, name, order, g1_len, g2_len, gt_len, g1_base, g2_base, g1_null, g2_null, neg):
        self.prefix, self.name, self.ORDER = prefix, name, order
        self.neg = neg  # neg(group, encoding) -> encoding of the inverse point (wire-level, no curve arithmetic)
        self.G1_LEN, self.G2_LEN, self.GT_LEN, self.SCALAR_LEN = g1_len, g2_len, gt_len, 32
        self.G1_BASE, self.G2_BASE, self.G1_NULL, self.G2_NULL = g1_base, g2_base, g1_null, g2_null
        # BLS12-381 inputs may be ZCash uncompressed (KYB_F_UNCOMPRESSED); bn256's wire format already is
        self.unc_factor = 2 if prefix == "bls12381" else 1

    def _in_len(self, group: int, flags: int) -> int:
        w = self.G1_LEN if group == 1 else self.G2_LEN
        return w * self.unc_factor if flags & F_UNCOMPRESSED else w

    def _fn(self, suffix):
        return getattr(load(), f"kyb_{self.prefix}_{suffix}"), f"kyb_{self.prefix}_{suffix}"

    def mul(self, group: int, scalars, points, same_base: bool, flags: int = 0):
        w = self.G1_LEN if group == 1 else self.G2_LEN
        if flags & F_UNCOMPRESSED_OUT:
            w *= self.unc_factor
        wi = self._in_len(group, flags)
        if _is_torch(scalars):
            import torch

            s = scalars.contiguous().view(-1, 32)
            if not _is_torch(points):  # (a base point given as bytes next to device scalars: Commit's default base)
                points = torch.from_numpy(_host(points, wi).copy()).to(s.device)
            p = points.contiguous().view(-1, wi)
            n = s.shape[0]
            if not same_base and p.shape[0] != n:
                raise ValueError("scalars/points length mismatch")
            out = torch.empty((n, w), dtype=torch.uint8, device=s.device)
            st = torch.empty(n, dtype=torch.uint8, device=s.device)
            fn, nm = self._fn(f"g{group}_mul_dev")
            check(fn(n, s.data_ptr(), p.data_ptr(), 0 if same_base else wi, out.data_ptr(), st.data_ptr(), flags,
                     _stream()), nm)
            return out, st
        s = _host(scalars, 32)
        p = _host(points, wi)
        n = s.shape[0]
        out = np.empty((n, w), dtype=np.uint8)
        st = np.empty(n, dtype=np.uint8)
        if same_base:
            fn, nm = self._fn(f"g{group}_mul_same_base")
        else:
            if p.shape[0] != n:
                raise ValueError("scalars/points length mismatch")
            fn, nm = self._fn(f"g{group}_mul")
        check(fn(n, s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data, flags), nm)
        return out, st

    def g1_batch_mul(self, scalars, points, flags: int = 0):
        """(out, status): out[i] = scalars[i] * points[i] on G1."""
        return self.mul(1, scalars, points, False, flags)

    def g2_batch_mul(self, scalars, points, flags: int = 0):
        return self.mul(2, scalars, points, False, flags)

    def g1_commit(self, scalars, base=None, flags: int = 0):
        """share.PriPoly.Commit (share/poly.go:143-149): commits[i] = coeffs[i] * base."""
        if base is None:  # the suite's own generator: valid by construction, nothing to re-check in every lane
            return self.mul(1, scalars, self.G1_BASE, True, flags | F_TRUSTED(0))
        return self.mul(1, scalars, base, True, flags)

    def g2_commit(self, scalars, base=None, flags: int = 0):
        if base is None:
            return self.mul(2, scalars, self.G2_BASE, True, flags | F_TRUSTED(0))
        return self.mul(2, scalars, base, True, flags)

    def add(self, group: int, a, b):
        """(out, status): out[i] = a[i] + b[i]  (N x Point.Add).  CUDA tensors stay on the device (enqueue only)."""
        w = self.G1_LEN if group == 1 else self.G2_LEN
        if _is_torch(a) and a.is_cuda:
            import torch

            if not _is_torch(b):
                b = torch.from_numpy(_host(b, w).copy())
            x, y = a.contiguous().view(-1, w), b.to(a.device).contiguous().view(-1, w)
            if x.shape != y.shape:
                raise ValueError("length mismatch")
            out = torch.empty_like(x)
            st = torch.empty(x.shape[0], dtype=torch.uint8, device=x.device)
            fn, nm = self._fn(f"g{group}_add_dev")
            check(fn(x.shape[0], x.data_ptr(), y.data_ptr(), out.data_ptr(), st.data_ptr(), _stream()), nm)
            return out, st
        x, y = _host(a, w), _host(b, w)
        if x.shape != y.shape:
            raise ValueError("length mismatch")
        n = x.shape[0]
        out = np.empty((n, w), dtype=np.uint8)
        st = np.empty(n, dtype=np.uint8)
        fn, nm = self._fn(f"g{group}_add")
        check(fn(n, x.ctypes.data, y.ctypes.data, out.ctypes.data, st.ctypes.data), nm)
        return out, st

    def _out_len(self, group: int, flags: int) -> int:
        w = self.G1_LEN if group == 1 else self.G2_LEN
        return w * self.unc_factor if flags & F_UNCOMPRESSED_OUT else w

    def batch_unmarshal(self, group: int, points, flags: int = 0):
        """(out, status): N x Point.UnmarshalBinary (kilic/g1.go:127-131; pairing/bn256/point.go:206-238, 466-499):
        status[i] != 0 where the reference returns an error, out[i] = the accepted point re-encoded (uncompressed
        affine with F_UNCOMPRESSED_OUT on BLS12-381).  Host buffers or torch device tensors.  Points that pass may
        be handed to later calls with F_TRUSTED(i) (| F_UNCOMPRESSED)."""
        wi, wo = self._in_len(group, flags), self._out_len(group, flags)
        if _is_torch(points):
            import torch

            p = points.contiguous().view(-1, wi)
            n = p.shape[0]
            out = torch.empty((n, wo), dtype=torch.uint8, device=p.device)
            st = torch.empty(max(n, 1), dtype=torch.uint8, device=p.device)
            fn, nm = self._fn(f"g{group}_unmarshal_dev")
            check(fn(n, p.data_ptr(), out.data_ptr(), st.data_ptr(), flags, _stream()), nm)
            return out, st[:n]
        p = _host(points, wi)
        n = p.shape[0]
        out = np.empty((n, wo), dtype=np.uint8)
        st = np.zeros(max(n, 1), dtype=np.uint8)
        fn, nm = self._fn(f"g{group}_unmarshal")
        check(fn(n, p.ctypes.data, out.ctypes.data, st.ctypes.data, flags), nm)
        return out, st[:n]

    def msm(self, group: int, scalars, points, flags: int = 0):
        """(out, status): out = sum_i scalars[i] * points[i] as ONE encoded point -- the MSM-shaped call
        sites of the reference (share/poly.go:340-348, 449-476; sign/bdn/bdn.go:126-181).  If any
        status is non-zero the output is all-zero bytes.  flags: F_TRUSTED(0) for points validated before (as every
        kyber.Point of the reference's call sites is), F_UNCOMPRESSED for BLS12-381 uncompressed-affine input."""
        w = self.G1_LEN if group == 1 else self.G2_LEN
        wi = self._in_len(group, flags)
        if _is_torch(scalars):
            import torch

            s = scalars.contiguous().view(-1, 32)
            p = points.contiguous().view(-1, wi)
            n = s.shape[0]
            if p.shape[0] != n:
                raise ValueError("scalars/points length mismatch")
            out = torch.empty(w, dtype=torch.uint8, device=s.device)
            st = torch.empty(max(n, 1), dtype=torch.uint8, device=s.device)
            fn, nm = self._fn(f"g{group}_msm_dev")
            check(fn(n, s.data_ptr(), p.data_ptr(), out.data_ptr(), st.data_ptr(), flags, _stream()), nm)
            return out, st[:n]
        s = _host(scalars, 32)
        p = _host(points, wi)
        n = s.shape[0]
        if p.shape[0] != n:
            raise ValueError("scalars/points length mismatch")
        out = np.empty(w, dtype=np.uint8)
        st = np.zeros(max(n, 1), dtype=np.uint8)
        fn, nm = self._fn(f"g{group}_msm")
        check(fn(n, s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data, flags), nm)
        return out, st[:n]

    def poly_eval(self, group: int, commits, indices, flags: int = 0):
        """(out, status): out[i] = sum_j commits[j] * (indices[i] + 1)^j -- share.PubPoly.Eval (share/poly.go:340-348)
        for many indices in one launch (host buffers); status has one entry per commitment."""
        w = self.G1_LEN if group == 1 else self.G2_LEN
        c = _host(commits, self._in_len(group, flags))
        idx = np.ascontiguousarray(np.asarray(indices, dtype=np.uint32))
        n, t = idx.shape[0], c.shape[0]
        out = np.empty((n, w), dtype=np.uint8)
        st = np.zeros(max(t, 1), dtype=np.uint8)
        fn, nm = self._fn(f"g{group}_poly_eval")
        check(fn(n, idx.ctypes.data, t, c.ctypes.data, out.ctypes.data, st.ctypes.data, flags), nm)
        return out, st[:t]

    def scalar_poly_eval(self, coeffs, indices):
        """out[i] = sum_j coeffs[j] * (indices[i] + 1)^j mod the group order, 32-byte big-endian scalars (mod.Int) --
        share.PriPoly.Eval (share/poly.go:85-93) for many indices in one launch: PriPoly.Shares (poly.go:96-102)."""
        c = _host(coeffs, 32)
        idx = np.ascontiguousarray(np.asarray(indices, dtype=np.uint32))
        n, t = idx.shape[0], c.shape[0]
        out = np.empty((n, 32), dtype=np.uint8)
        fn, nm = self._fn("scalar_poly_eval")
        check(fn(n, idx.ctypes.data, t, c.ctypes.data, out.ctypes.data), nm)
        return out

    def g1_msm(self, scalars, points, flags: int = 0):
        return self.msm(1, scalars, points, flags)

    def g2_msm(self, scalars, points, flags: int = 0):
        return self.msm(2, scalars, points, flags)

    def batch_pair(self, g1, g2, flags: int = 0):
        """(gt, status): gt[i] = e(g1[i], g2[i])  (N x Suite.Pair)."""
        w1, w2 = self._in_len(1, flags), self._in_len(2, flags)
        if _is_torch(g1):
            import torch

            a = g1.contiguous().view(-1, w1)
            b = g2.contiguous().view(-1, w2)
            n = a.shape[0]
            if b.shape[0] != n:
                raise ValueError("g1/g2 length mismatch")
            gt = torch.empty((n, self.GT_LEN), dtype=torch.uint8, device=a.device)
            st = torch.empty(n, dtype=torch.uint8, device=a.device)
            fn, nm = self._fn("pair_dev")
            check(fn(n, a.data_ptr(), b.data_ptr(), gt.data_ptr(), st.data_ptr(), flags, _stream()), nm)
            return gt, st
        a, b = _host(g1, w1), _host(g2, w2)
        n = a.shape[0]
        if b.shape[0] != n:
            raise ValueError("g1/g2 length mismatch")
        gt = np.empty((n, self.GT_LEN), dtype=np.uint8)
        st = np.empty(n, dtype=np.uint8)
        fn, nm = self._fn("pair")
        check(fn(n, a.ctypes.data, b.ctypes.data, gt.ctypes.data, st.ctypes.data, flags), nm)
        return gt, st

    def gt_batch_mul(self, scalars, gts):
        """(out, status): out[i] = gts[i] ^ scalars[i]  (N x GT Point.Mul: pairing/bn256/point.go:613,
        kilic/gt.go:79-84)."""
        if _is_torch(scalars):
            import torch

            s = scalars.contiguous().view(-1, 32)
            g = gts.contiguous().view(-1, self.GT_LEN)
            n = s.shape[0]
            if g.shape[0] != n:
                raise ValueError("length mismatch")
            out = torch.empty_like(g)
            st = torch.empty(n, dtype=torch.uint8, device=s.device)
            fn, nm = self._fn("gt_mul_dev")
            check(fn(n, s.data_ptr(), g.data_ptr(), out.data_ptr(), st.data_ptr(), _stream()), nm)
            return out, st
        s, g = _host(scalars, 32), _host(gts, self.GT_LEN)
        n = s.shape[0]
        if g.shape[0] != n:
            raise ValueError("length mismatch")
        out = np.empty_like(g)
        st = np.empty(n, dtype=np.uint8)
        fn, nm = self._fn("gt_mul")
        check(fn(n, s.ctypes.data, g.ctypes.data, out.ctypes.data, st.ctypes.data), nm)
        return out, st

    def batch_validate_pairing(self, p1, p2, inv1, inv2, flags: int = 0):
        """(ok, status): ok[i] = e(p1[i], p2[i]) == e(inv1[i], inv2[i])  (N x Suite.ValidatePairing,
        pairing/pairing.go:13-15).  p1/inv1 are G1, p2/inv2 are G2; F_TRUSTED(0..3) refer to p1, p2, inv1, inv2."""
        w1, w2 = self._in_len(1, flags), self._in_len(2, flags)
        if _is_torch(p1):
            import torch

            a, c = p1.contiguous().view(-1, w1), inv1.contiguous().view(-1, w1)
            b, d = p2.contiguous().view(-1, w2), inv2.contiguous().view(-1, w2)
            n = a.shape[0]
            if not (b.shape[0] == c.shape[0] == d.shape[0] == n):
                raise ValueError("length mismatch")
            ok = torch.empty(n, dtype=torch.uint8, device=a.device)
            st = torch.empty(n, dtype=torch.uint8, device=a.device)
            fn, nm = self._fn("pair_check_dev")
            check(fn(n, a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), ok.data_ptr(), st.data_ptr(), flags,
                     _stream()), nm)
            return ok, st
        a, c = _host(p1, w1), _host(inv1, w1)
        b, d = _host(p2, w2), _host(inv2, w2)
        n = a.shape[0]
        if not (b.shape[0] == c.shape[0] == d.shape[0] == n):
            raise ValueError("length mismatch")
        ok = np.empty(n, dtype=np.uint8)
        st = np.empty(n, dtype=np.uint8)
        fn, nm = self._fn("pair_check")
        check(fn(n, a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, ok.ctypes.data, st.ctypes.data, flags), nm)
        return ok, st

    # ------------------------------------------------------------ kyber interface mirrors
    def make_types(self):
        eng = self
        ORDER = self.ORDER

        class Scalar:
            """mod.Int modulo the group order, 32-byte big-endian wire format (group/mod/int.go:334-350)."""

            __slots__ = ("v",)

            def __init__(self, v: int = 0):
                self.v = v % ORDER

            def MarshalBinary(self) -> bytes:
                return self.v.to_bytes(32, "big")

            def UnmarshalBinary(self, buf: bytes):
                if len(buf) != 32:
                    raise ValueError("UnmarshalBinary: wrong size buffer")
                x = int.from_bytes(buf, "big")
                if x >= ORDER:  # group/mod/int.go:362-364
                    raise ValueError("UnmarshalBinary: value out of range")
                self.v = x
                return self

            def MarshalSize(self) -> int: return 32
            def SetInt64(self, x: int): self.v = x % ORDER; return self
            def SetBytes(self, b: bytes): self.v = int.from_bytes(b, "big") % ORDER; return self
            def Zero(self): return self.SetInt64(0)
            def One(self): return self.SetInt64(1)
            def Set(self, a): self.v = _sc(a).v; return self
            def Clone(self): return Scalar(self.v)
            def Equal(self, a) -> bool: return self.v == _sc(a).v
            def Add(self, a, b): self.v = (_sc(a).v + _sc(b).v) % ORDER; return self
            def Sub(self, a, b): self.v = (_sc(a).v - _sc(b).v) % ORDER; return self
            def Neg(self, a): self.v = -_sc(a).v % ORDER; return self
            def Mul(self, a, b): self.v = _sc(a).v * _sc(b).v % ORDER; return self
            def Inv(self, a): self.v = pow(_sc(a).v, ORDER - 2, ORDER); return self
            def Div(self, a, b): self.v = _sc(a).v * pow(_sc(b).v, ORDER - 2, ORDER) % ORDER; return self

            def Pick(self, rand=None):
                raw = rand(64) if rand is not None else os.urandom(64)
                self.v = int.from_bytes(raw, "big") % ORDER
                return self

            def String(self) -> str: return self.MarshalBinary().hex()
            __repr__ = String

        def _sc(s) -> Scalar:
            if not isinstance(s, Scalar):
                raise TypeError(f"ErrTypeCast: not a {eng.name} scalar")
            return s

        class _Elt:
            """G1/G2 element held as its canonical wire encoding; arithmetic = engine calls."""

            __slots__ = ("enc",)
            GROUP, LEN, BASE, NULL = 0, 0, b"", b""

            def __init__(self, enc=None):
                self.enc = self.NULL if enc is None else bytes(enc)

            def MarshalBinary(self) -> bytes: return self.enc
            def MarshalSize(self) -> int: return self.LEN

            def UnmarshalBinary(self, buf: bytes):
                if len(buf) != self.LEN:
                    raise ValueError(f"{eng.name}: wrong size buffer")
                out, st = eng.mul(self.GROUP, (1).to_bytes(32, "big"), buf, False)
                if st[0]:
                    raise ValueError(f"{eng.name}: malformed point" if st[0] == 1 else f"{eng.name}: point not in subgroup")
                self.enc = bytes(out[0])
                return self

            def Null(self): self.enc = self.NULL; return self
            def Base(self): self.enc = self.BASE; return self
            def Set(self, p): self.enc = self._cast(p).enc; return self
            def Clone(self): return type(self)(self.enc)
            def Equal(self, p) -> bool: return self.enc == self._cast(p).enc

            def Add(self, a, b):
                out, st = eng.add(self.GROUP, self._cast(a).enc, self._cast(b).enc)
                if st.any():
                    raise ValueError(f"{eng.name}: invalid point")
                self.enc = bytes(out[0])
                return self

            def Neg(self, a):
                self.enc = eng.neg(self.GROUP, self._cast(a).enc)
                return self

            def Sub(self, a, b):
                nb = type(self)().Neg(b)
                return self.Add(a, nb)

            def Pick(self, rand=None):
                """A uniformly random group element as k * Base (what pairing/bn256/point.go:48-57 does)."""
                return self.Mul(Scalar().Pick(rand), None)

            def Mul(self, s, A=None):
                """Point.Mul, a batch of ONE: ~1-9 ms on the device whatever the batch size (profiles/
                r03_single_call_latency.json) against 0.1-0.3 ms on a CPU core of the reference.  The Go suite keeps
                single elements on the embedded reference (go/kyberhip/suite/point.go, MinDeviceBatch); this mirror
                has no CPU implementation to delegate to -- the product path has no CPU fallback -- so loops belong
                in g1_batch_mul / g1_commit / g1_msm."""
                base = self.BASE if A is None else self._cast(A).enc
                out, st = eng.mul(self.GROUP, _sc(s).MarshalBinary(), base, False)
                if st[0]:
                    raise ValueError(f"{eng.name}: invalid point")
                self.enc = bytes(out[0])
                return self

            def _cast(self, p):
                if type(p) is not type(self):
                    raise TypeError(f"ErrTypeCast: wrong {eng.name} group element")
                return p

            def String(self) -> str: return self.enc.hex()
            __repr__ = String

        class G1Elt(_Elt):
            __slots__ = ()
            GROUP, LEN, BASE, NULL = 1, eng.G1_LEN, eng.G1_BASE, eng.G1_NULL

        class G2Elt(_Elt):
            __slots__ = ()
            GROUP, LEN, BASE, NULL = 2, eng.G2_LEN, eng.G2_BASE, eng.G2_NULL

        class GTElt:
            __slots__ = ("enc",)

            def __init__(self, enc: bytes = b""):
                self.enc = bytes(enc)

            def MarshalBinary(self) -> bytes: return self.enc
            def MarshalSize(self) -> int: return eng.GT_LEN
            def Equal(self, o) -> bool: return self.enc == o.enc

            def Mul(self, s, q):
                out, st = eng.gt_batch_mul(_sc(s).MarshalBinary(), q.enc)
                if st[0]:
                    raise ValueError(f"{eng.name}: invalid GT element")
                self.enc = bytes(out[0])
                return self

            def Pair(self, p1, p2):
                gt, st = eng.batch_pair(p1.enc, p2.enc)
                if st[0]:
                    raise ValueError(f"{eng.name}: invalid pairing input")
                self.enc = bytes(gt[0])
                return self

        class _Group:
            def __init__(self, elt, gname):
                self._elt, self._name = elt, gname

            def String(self): return self._name
            def ScalarLen(self): return 32
            def Scalar(self): return Scalar()
            def PointLen(self): return self._elt.LEN
            def Point(self): return self._elt()

        class Suite:
            """pairing.Suite (pairing/pairing.go:8-20): G1(), G2(), GT(), Pair, ValidatePairing."""

            def G1(self): return _Group(G1Elt, eng.name + ".G1")
            def G2(self): return _Group(G2Elt, eng.name + ".G2")
            def GT(self): return GTElt

            def Pair(self, p1, p2):
                if not isinstance(p1, G1Elt) or not isinstance(p2, G2Elt):
                    raise TypeError("ErrTypeCast")
                return GTElt().Pair(p1, p2)

            def ValidatePairing(self, p1, p2, inv1, inv2) -> bool:
                ok, st = eng.batch_validate_pairing(p1.enc, p2.enc, inv1.enc, inv2.enc)
                if st[0]:
                    raise ValueError(f"{eng.name}: invalid pairing input")
                return bool(ok[0])

        return Scalar, G1Elt, G2Elt, GTElt, Suite
