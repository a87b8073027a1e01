"""Host-side mirror of the reference's ``pairing/bls12381`` suites for the hot path
(pairing.Suite, pairing/pairing.go:8-20; kilic adapter kilic/{g1,g2,gt,suite}.go), backed by the
HIP engine through the C ABI.  No curve or field arithmetic happens in Python.

Wire formats are the adapters' MarshalBinary encodings: scalars 32-byte big-endian (mod.Int),
G1 48-byte / G2 96-byte ZCash compressed, GT 576 bytes.

Batch functions accept host data (bytes / numpy uint8) or device-resident ``torch.uint8`` CUDA
tensors; device inputs are processed on the current stream and results stay on the device.
"""
from __future__ import annotations

import os

import numpy as np

from .._lib import check, load

ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # kilic/scalar.go:11-12
G1_LEN, G2_LEN, GT_LEN, SCALAR_LEN = 48, 96, 576, 32
G1_BASE = bytes.fromhex(
    "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
G2_BASE = bytes.fromhex(
    "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
    "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")
G1_NULL = bytes([0xC0]) + bytes(47)
G2_NULL = bytes([0xC0]) + bytes(95)


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _host(buf, width: int) -> np.ndarray:
    a = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray, memoryview)) else np.asarray(buf, dtype=np.uint8)
    return np.ascontiguousarray(a).reshape(-1, width)


def _stream():
    import torch

    return torch.cuda.current_stream().cuda_stream


def _mul(group: int, scalars, points, same_base: bool):
    lib = load()
    w = G1_LEN if group == 1 else G2_LEN
    name = f"kyb_bls12381_g{group}_mul"
    if _is_torch(scalars):
        import torch

        s = scalars.contiguous().view(-1, 32)
        p = points.contiguous().view(-1, w)
        n = s.shape[0]
        if not same_base and p.shape[0] != n:
            raise ValueError("scalars/points length mismatch")
        out = torch.empty((n, w), dtype=torch.uint8, device=s.device)
        st = torch.empty(n, dtype=torch.uint8, device=s.device)
        check(getattr(lib, name + "_dev")(n, s.data_ptr(), p.data_ptr(), 0 if same_base else w, out.data_ptr(),
                                          st.data_ptr(), _stream()), name + "_dev")
        return out, st
    s = _host(scalars, 32)
    p = _host(points, w)
    n = s.shape[0]
    out = np.empty((n, w), dtype=np.uint8)
    st = np.empty(n, dtype=np.uint8)
    if same_base:
        check(getattr(lib, name + "_same_base")(n, s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data),
              name + "_same_base")
    else:
        if p.shape[0] != n:
            raise ValueError("scalars/points length mismatch")
        check(getattr(lib, name)(n, s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data), name)
    return out, st


def g1_batch_mul(scalars, points):
    """(out, status): out[i] = scalars[i] * points[i] on G1 (N x G1Elt.Mul, kilic/g1.go:110-116)."""
    return _mul(1, scalars, points, False)


def g2_batch_mul(scalars, points):
    return _mul(2, scalars, points, False)


def g1_commit(scalars, base=G1_BASE):
    """share.PriPoly.Commit on G1 (share/poly.go:143-149): commits[i] = coeffs[i] * base."""
    return _mul(1, scalars, base, True)


def g2_commit(scalars, base=G2_BASE):
    return _mul(2, scalars, base, True)


def batch_pair(g1, g2):
    """(gt, status): gt[i] = e(g1[i], g2[i])  (N x Suite.Pair, kilic/suite.go:70-75)."""
    lib = load()
    if _is_torch(g1):
        import torch

        a = g1.contiguous().view(-1, G1_LEN)
        b = g2.contiguous().view(-1, G2_LEN)
        n = a.shape[0]
        if b.shape[0] != n:
            raise ValueError("g1/g2 length mismatch")
        gt = torch.empty((n, GT_LEN), dtype=torch.uint8, device=a.device)
        st = torch.empty(n, dtype=torch.uint8, device=a.device)
        check(lib.kyb_bls12381_pair_dev(n, a.data_ptr(), b.data_ptr(), gt.data_ptr(), st.data_ptr(), _stream()),
              "kyb_bls12381_pair_dev")
        return gt, st
    a, b = _host(g1, G1_LEN), _host(g2, G2_LEN)
    n = a.shape[0]
    if b.shape[0] != n:
        raise ValueError("g1/g2 length mismatch")
    gt = np.empty((n, GT_LEN), dtype=np.uint8)
    st = np.empty(n, dtype=np.uint8)
    check(lib.kyb_bls12381_pair(n, a.ctypes.data, b.ctypes.data, gt.ctypes.data, st.ctypes.data), "kyb_bls12381_pair")
    return gt, st


def batch_validate_pairing(p1, p2, inv1, inv2):
    """(ok, status): ok[i] = e(p1[i], p2[i]) == e(inv1[i], inv2[i])  (N x Suite.ValidatePairing,
    pairing/pairing.go:13-15).  p1/inv1 are G1, p2/inv2 are G2."""
    lib = load()
    if _is_torch(p1):
        import torch

        a, c = p1.contiguous().view(-1, G1_LEN), inv1.contiguous().view(-1, G1_LEN)
        b, d = p2.contiguous().view(-1, G2_LEN), inv2.contiguous().view(-1, G2_LEN)
        n = a.shape[0]
        if not (b.shape[0] == c.shape[0] == d.shape[0] == n):
            raise ValueError("length mismatch")
        ok = torch.empty(n, dtype=torch.uint8, device=a.device)
        st = torch.empty(n, dtype=torch.uint8, device=a.device)
        check(lib.kyb_bls12381_pair_check_dev(n, a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(),
                                              ok.data_ptr(), st.data_ptr(), _stream()), "kyb_bls12381_pair_check_dev")
        return ok, st
    a, c = _host(p1, G1_LEN), _host(inv1, G1_LEN)
    b, d = _host(p2, G2_LEN), _host(inv2, G2_LEN)
    n = a.shape[0]
    if not (b.shape[0] == c.shape[0] == d.shape[0] == n):
        raise ValueError("length mismatch")
    ok = np.empty(n, dtype=np.uint8)
    st = np.empty(n, dtype=np.uint8)
    check(lib.kyb_bls12381_pair_check(n, a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, ok.ctypes.data,
                                      st.ctypes.data), "kyb_bls12381_pair_check")
    return ok, st


# ------------------------------------------------------------ kyber.Scalar mirror (mod.Int mod r)
class Scalar:
    """kilic scalar = mod.Int modulo r, 32-byte big-endian wire format (group/mod/int.go:334-350)."""

    __slots__ = ("v",)

    def __init__(self, v: int = 0):
        self.v = v % ORDER

    def MarshalBinary(self) -> bytes:
        return self.v.to_bytes(32, "big")

    def UnmarshalBinary(self, buf: bytes) -> "Scalar":
        if len(buf) != 32:
            raise ValueError("UnmarshalBinary: wrong size buffer")
        x = int.from_bytes(buf, "big")
        if x >= ORDER:  # group/mod/int.go:362-364
            raise ValueError("UnmarshalBinary: value out of range")
        self.v = x
        return self

    def MarshalSize(self) -> int:
        return 32

    def SetInt64(self, x: int) -> "Scalar":
        self.v = x % ORDER
        return self

    def SetBytes(self, b: bytes) -> "Scalar":
        self.v = int.from_bytes(b, "big") % ORDER
        return self

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

    def Pick(self, rand=None) -> "Scalar":
        raw = rand(64) if rand is not None else os.urandom(64)
        self.v = int.from_bytes(raw, "big") % ORDER
        return self

    def String(self) -> str:
        return self.MarshalBinary().hex()

    __repr__ = String


def _sc(s) -> Scalar:
    if not isinstance(s, Scalar):
        raise TypeError("ErrTypeCast: not a bls12381 scalar")
    return s


class _Elt:
    """Shared G1/G2 behaviour: the element is held as its canonical compressed encoding."""

    __slots__ = ("enc",)
    GROUP = 0
    LEN = 0
    BASE = b""
    NULL = b""

    def __init__(self, enc: bytes | None = None):
        self.enc = self.NULL if enc is None else bytes(enc)

    def MarshalBinary(self) -> bytes:
        return self.enc

    def MarshalSize(self) -> int:
        return self.LEN

    def UnmarshalBinary(self, buf: bytes):
        if len(buf) != self.LEN:
            raise ValueError("bls12381: wrong size buffer")
        out, st = _mul(self.GROUP, (1).to_bytes(32, "big"), buf, False)
        if st[0]:
            raise ValueError("bls12381: invalid point encoding" if st[0] == 1 else "bls12381: point not in subgroup")
        self.enc = bytes(out[0])
        return self

    def Null(self): self.enc = self.NULL; return self
    def Base(self): self.enc = self.BASE; return self
    def Set(self, p): self.enc = self._cast(p).enc; return self
    def Clone(self): return type(self)(self.enc)
    def Equal(self, p) -> bool: return self.enc == self._cast(p).enc

    def Mul(self, s: Scalar, A=None):
        base = self.BASE if A is None else self._cast(A).enc
        out, st = _mul(self.GROUP, _sc(s).MarshalBinary(), base, False)
        if st[0]:
            raise ValueError("bls12381: invalid point")
        self.enc = bytes(out[0])
        return self

    def _cast(self, p):
        if type(p) is not type(self):
            raise TypeError("ErrTypeCast: wrong bls12381 group element")
        return p

    def String(self) -> str:
        return self.enc.hex()

    __repr__ = String


class G1Elt(_Elt):
    __slots__ = ()
    GROUP, LEN, BASE, NULL = 1, G1_LEN, G1_BASE, G1_NULL


class G2Elt(_Elt):
    __slots__ = ()
    GROUP, LEN, BASE, NULL = 2, G2_LEN, G2_BASE, G2_NULL


class GTElt:
    __slots__ = ("enc",)

    def __init__(self, enc: bytes = b""):
        self.enc = bytes(enc)

    def MarshalBinary(self) -> bytes:
        return self.enc

    def MarshalSize(self) -> int:
        return GT_LEN

    def Equal(self, o) -> bool:
        return self.enc == o.enc

    def Pair(self, p1: G1Elt, p2: G2Elt) -> "GTElt":
        gt, st = batch_pair(p1.enc, p2.enc)
        if st[0]:
            raise ValueError("bls12381: invalid pairing input")
        self.enc = bytes(gt[0])
        return self


class _Group:
    def __init__(self, elt, name):
        self._elt, self._name = elt, name

    def String(self): return self._name
    def ScalarLen(self): return SCALAR_LEN
    def Scalar(self): return Scalar()
    def PointLen(self): return self._elt.LEN
    def Point(self): return self._elt()


class Suite:
    """pairing.Suite (pairing/pairing.go:8-20) for BLS12-381, kilic adapter conventions
    (kilic/suite.go:20-75): G1(), G2(), GT(), Pair, ValidatePairing."""

    def G1(self): return _Group(G1Elt, "bls12-381.G1")
    def G2(self): return _Group(G2Elt, "bls12-381.G2")
    def GT(self): return GTElt

    def Pair(self, p1: G1Elt, p2: G2Elt) -> GTElt:
        return GTElt().Pair(p1, p2)

    def ValidatePairing(self, p1: G1Elt, p2: G2Elt, inv1: G1Elt, inv2: G2Elt) -> bool:
        ok, st = batch_validate_pairing(p1.enc, p2.enc, inv1.enc, inv2.enc)
        if st[0]:
            raise ValueError("bls12381: invalid pairing input")
        return bool(ok[0])


def NewSuite() -> Suite:
    return Suite()
