"""Host-side mirror of the reference's ``group/edwards25519`` suite for the hot
path (kyber.Group / kyber.Point / kyber.Scalar, group.go:23-183), backed by
the HIP engine through the C ABI.  No group arithmetic happens in Python.

Conventions kept from the reference:
  * receiver-mutating methods that return the receiver (``P.Mul(s, A)`` sets P);
  * ``A is None`` means the base point (group.go:128-130);
  * a wrong concrete type raises ``TypeError`` (the reference panics with
    ErrTypeCast, point.go:237-249);
  * ``UnmarshalBinary`` raises ``ValueError`` for an invalid encoding
    (the reference returns an error, point.go:65-70);
  * scalars are 32-byte little-endian; ``UnmarshalBinary`` copies them
    unreduced (scalar.go:226-233), ``SetBytes`` reduces mod l (scalar.go:187).

The batch API (``batch_mul``, ``batch_mul_base``, ``commit``) is what the
engine adds: the reference has no batch entry points.
Batch functions accept either host data (bytes / numpy uint8 arrays) or
device-resident ``torch.uint8`` CUDA tensors; device inputs are processed in
place on the current stream and the result is a CUDA tensor.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np

from .. import _lib
from .._lib import KYB_F_UNIFORM, KYB_F_VARTIME, check, load

# group/edwards25519/const.go:15
ORDER = 2**252 + 27742317777372353535851937790883648493
_P = 2**255 - 19
POINT_LEN = 32
SCALAR_LEN = 32
_BASE_ENC = bytes([0x58]) + bytes([0x66]) * 31
_NULL_ENC = bytes([1]) + bytes(31)


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _as_host(buf, n_elem_bytes: int) -> np.ndarray:
    a = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray, memoryview)) else np.asarray(buf, dtype=np.uint8)
    a = np.ascontiguousarray(a).reshape(-1, n_elem_bytes)
    return a


def _stream_ptr():
    import torch

    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------ batch API
def _flags(vartime: bool, uniform: bool) -> int:
    if vartime and uniform:
        raise ValueError("vartime and uniform are exclusive")
    return KYB_F_VARTIME if vartime else (KYB_F_UNIFORM if uniform else 0)


def batch_mul_base(scalars, vartime: bool = False, uniform: bool = False):
    """out[i] = scalars[i] * B   (replaces N x Point.Mul(s, nil), ge.go:373).  uniform: KYB_F_UNIFORM -- the table is
    scanned, not indexed (the access pattern of the reference's constant-time path), for secret scalars."""
    lib = load()
    flags = _flags(vartime, uniform)
    if _is_torch(scalars):
        import torch

        s = scalars.contiguous().view(-1, 32)
        out = torch.empty_like(s)
        check(lib.kyb_ed25519_mul_base_dev(s.shape[0], s.data_ptr(), out.data_ptr(), flags, _stream_ptr()),
              "kyb_ed25519_mul_base_dev")
        return out
    s = _as_host(scalars, 32)
    out = np.empty_like(s)
    check(lib.kyb_ed25519_mul_base(s.shape[0], s.ctypes.data, out.ctypes.data, flags), "kyb_ed25519_mul_base")
    return out


def batch_mul(scalars, points, vartime: bool = False, uniform: bool = False):
    """(out, status): out[i] = scalars[i] * points[i]; status[i] != 0 where
    points[i] is not a valid encoding (then out[i] is zero bytes).
    Replaces N x (UnmarshalBinary + Point.Mul(s, A) + MarshalBinary)."""
    lib = load()
    flags = _flags(vartime, uniform)
    if _is_torch(scalars):
        import torch

        s = scalars.contiguous().view(-1, 32)
        p = points.contiguous().view(-1, 32)
        if s.shape != p.shape:
            raise ValueError("scalars/points length mismatch")
        out = torch.empty_like(s)
        st = torch.empty(s.shape[0], dtype=torch.uint8, device=s.device)
        check(lib.kyb_ed25519_mul_dev(s.shape[0], s.data_ptr(), p.data_ptr(), out.data_ptr(), st.data_ptr(),
                                      flags, _stream_ptr()), "kyb_ed25519_mul_dev")
        return out, st
    s = _as_host(scalars, 32)
    p = _as_host(points, 32)
    if s.shape != p.shape:
        raise ValueError("scalars/points length mismatch")
    out = np.empty_like(s)
    st = np.empty(s.shape[0], dtype=np.uint8)
    check(lib.kyb_ed25519_mul(s.shape[0], s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data, flags),
          "kyb_ed25519_mul")
    return out, st


def commit(scalars, base=None, vartime: bool = False, uniform: bool = False):
    """commits[i] = coeffs[i] * b  -- share.PriPoly.Commit (share/poly.go:143-149).
    ``base`` None means the standard base point (poly.go:144 passes nil through).  uniform: the coefficients of a
    PriPoly are secrets -- KYB_F_UNIFORM keeps them out of the memory addresses."""
    if base is None:
        return batch_mul_base(scalars, vartime, uniform)
    lib = load()
    flags = _flags(vartime, uniform)
    s = _as_host(scalars, 32)
    b = _as_host(base, 32)
    out = np.empty_like(s)
    st = np.empty(s.shape[0], dtype=np.uint8)
    check(lib.kyb_ed25519_mul_same_base(s.shape[0], s.ctypes.data, b.ctypes.data, out.ctypes.data, st.ctypes.data,
                                        flags), "kyb_ed25519_mul_same_base")
    if s.shape[0] and st[0]:
        raise ValueError("invalid Ed25519 curve point")
    return out


def msm(scalars, points, scalar_bits: int = 256):
    """(out, status): out = sum_i scalars[i] * points[i] as one 32-byte point -- what PubPoly.Eval /
    RecoverCommit (share/poly.go:340-348, 449-476) compute with N x (Mul + Add).  Scalars are 256-bit
    little-endian integers, never reduced mod l; one whose top radix-16 digit the reference's recoding drops (>= ~2^255)
    counts as the integer the reference's Mul multiplies by, as in batch_mul.  If any status is non-zero the output is
    all-zero bytes.  scalar_bits < 256 (host buffers): every scalar is below 2^scalar_bits, higher bits are ignored
    (KYB_F_SCALAR_BITS: proportionally fewer windows)."""
    lib = load()
    if _is_torch(scalars):
        import torch

        s = scalars.contiguous().view(-1, 32)
        p = points.contiguous().view(-1, 32)
        if s.shape != p.shape:
            raise ValueError("scalars/points length mismatch")
        n = s.shape[0]
        out = torch.empty(32, dtype=torch.uint8, device=s.device)
        st = torch.empty(max(n, 1), dtype=torch.uint8, device=s.device)
        check(lib.kyb_ed25519_msm_dev(n, s.data_ptr(), p.data_ptr(), out.data_ptr(), st.data_ptr(), _stream_ptr()),
              "kyb_ed25519_msm_dev")
        return out, st[:n]
    s = _as_host(scalars, 32)
    p = _as_host(points, 32)
    if s.shape != p.shape:
        raise ValueError("scalars/points length mismatch")
    n = s.shape[0]
    out = np.empty(32, dtype=np.uint8)
    st = np.zeros(max(n, 1), dtype=np.uint8)
    if scalar_bits != 256:
        check(lib.kyb_ed25519_msm_flags(n, s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data,
                                        scalar_bits << 16), "kyb_ed25519_msm_flags")
        return out, st[:n]
    check(lib.kyb_ed25519_msm(n, s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data), "kyb_ed25519_msm")
    return out, st[:n]


def poly_eval(commits, indices):
    """(out, status): out[i] = sum_j commits[j] * (indices[i] + 1)^j -- share.PubPoly.Eval (share/poly.go:340-348) for
    many indices in one launch (host buffers).  status has one entry per commitment."""
    lib = load()
    c = _as_host(commits, 32)
    idx = np.ascontiguousarray(np.asarray(indices, dtype=np.uint32))
    n, t = idx.shape[0], c.shape[0]
    out = np.empty((n, 32), dtype=np.uint8)
    st = np.zeros(max(t, 1), dtype=np.uint8)
    check(lib.kyb_ed25519_poly_eval(n, idx.ctypes.data, t, c.ctypes.data, out.ctypes.data, st.ctypes.data),
          "kyb_ed25519_poly_eval")
    return out, st[:t]


def scalar_poly_eval(coeffs, indices):
    """out[i] = sum_j coeffs[j] * (indices[i] + 1)^j mod l, 32-byte little-endian scalars -- share.PriPoly.Eval
    (share/poly.go:85-93) for many indices in one launch: PriPoly.Shares (poly.go:96-102)."""
    lib = load()
    c = _as_host(coeffs, 32)
    idx = np.ascontiguousarray(np.asarray(indices, dtype=np.uint32))
    n, t = idx.shape[0], c.shape[0]
    out = np.empty((n, 32), dtype=np.uint8)
    check(lib.kyb_ed25519_scalar_poly_eval(n, idx.ctypes.data, t, c.ctypes.data, out.ctypes.data),
          "kyb_ed25519_scalar_poly_eval")
    return out


def batch_unmarshal(points):
    """(out, status): N x (*point).UnmarshalBinary (point.go:65-70 -> ge.go:110-150): status[i] != 0 where the reference
    returns an error; out[i] = MarshalBinary of the accepted point (canonical y, point.go:54-58)."""
    lib = load()
    if _is_torch(points):
        import torch

        p = points.contiguous().view(-1, 32)
        out = torch.empty_like(p)
        st = torch.empty(max(p.shape[0], 1), dtype=torch.uint8, device=p.device)
        check(lib.kyb_ed25519_unmarshal_dev(p.shape[0], p.data_ptr(), out.data_ptr(), st.data_ptr(), _stream_ptr()),
              "kyb_ed25519_unmarshal_dev")
        return out, st[:p.shape[0]]
    p = _as_host(points, 32)
    n = p.shape[0]
    out = np.empty((n, 32), dtype=np.uint8)
    st = np.zeros(max(n, 1), dtype=np.uint8)
    check(lib.kyb_ed25519_unmarshal(n, p.ctypes.data, out.ctypes.data, st.ctypes.data), "kyb_ed25519_unmarshal")
    return out, st[:n]


def batch_add(a, b):
    """(out, status): out[i] = a[i] + b[i]  (N x Point.Add, point.go:216-223)."""
    lib = load()
    if _is_torch(a):
        import torch

        x, y = a.contiguous().view(-1, 32), b.contiguous().view(-1, 32)
        if x.shape != y.shape:
            raise ValueError("length mismatch")
        out = torch.empty_like(x)
        st = torch.empty(x.shape[0], dtype=torch.uint8, device=x.device)
        check(lib.kyb_ed25519_add_dev(x.shape[0], x.data_ptr(), y.data_ptr(), out.data_ptr(), st.data_ptr(), _stream_ptr()),
              "kyb_ed25519_add_dev")
        return out, st
    x, y = _as_host(a, 32), _as_host(b, 32)
    if x.shape != y.shape:
        raise ValueError("length mismatch")
    out = np.empty_like(x)
    st = np.empty(x.shape[0], dtype=np.uint8)
    check(lib.kyb_ed25519_add(x.shape[0], x.ctypes.data, y.ctypes.data, out.ctypes.data, st.ctypes.data), "kyb_ed25519_add")
    return out, st


def batch_hash(msgs, dst: bytes):
    """out[i] = Hash(msgs[i], dst): (*point).Hash (point.go:325-334, RFC 9380 edwards25519_XMD:SHA-512_ELL2_RO_)
    for n equal-length messages (list of bytes, (n, len) uint8 array or CUDA tensor)."""
    import ctypes

    lib = load()
    dbuf = ctypes.create_string_buffer(bytes(dst), len(dst))
    dptr = ctypes.cast(dbuf, ctypes.c_void_p)
    if _is_torch(msgs):
        import torch

        m = msgs.contiguous()
        n, ln = m.shape[0], m.shape[1]
        out = torch.empty((n, 32), dtype=torch.uint8, device=m.device)
        check(lib.kyb_ed25519_hash_dev(n, m.data_ptr(), ln, dptr, len(dst), out.data_ptr(), _stream_ptr()),
              "kyb_ed25519_hash_dev")
        return out
    if isinstance(msgs, (list, tuple)):
        ln = len(msgs[0]) if msgs else 0
        if any(len(x) != ln for x in msgs):
            raise ValueError("batch_hash: messages must have equal length")
        n = len(msgs)
        buf = np.frombuffer(b"".join(msgs), dtype=np.uint8)
    else:
        a = np.ascontiguousarray(msgs, dtype=np.uint8)
        n, ln = a.shape[0], a.shape[1]
        buf = a.reshape(-1)
    buf = np.ascontiguousarray(buf) if buf.size else np.zeros(1, dtype=np.uint8)
    out = np.empty((n, 32), dtype=np.uint8)
    check(lib.kyb_ed25519_hash(n, buf.ctypes.data, ln, dptr, len(dst), out.ctypes.data), "kyb_ed25519_hash")
    return out


# ------------------------------------------------------- kyber.Scalar mirror
class Scalar:
    """kyber.Scalar for Ed25519 (group/edwards25519/scalar.go:32-34): 32 bytes LE."""

    __slots__ = ("v",)

    def __init__(self, v: bytes = bytes(32)):
        self.v = bytes(v)

    # -- encoding
    def MarshalBinary(self) -> bytes:
        return (int.from_bytes(self.v, "little") % ORDER).to_bytes(32, "little")

    def UnmarshalBinary(self, buf: bytes) -> "Scalar":
        if len(buf) != 32:
            raise ValueError("wrong size buffer")
        self.v = bytes(buf)  # unreduced, scalar.go:226-233
        return self

    def MarshalSize(self) -> int:
        return 32

    def SetBytes(self, b: bytes) -> "Scalar":
        self.v = (int.from_bytes(b, "little") % ORDER).to_bytes(32, "little")
        return self

    def SetInt64(self, v: int) -> "Scalar":
        self.v = (v % ORDER).to_bytes(32, "little")
        return self

    def Zero(self) -> "Scalar":
        self.v = bytes(32)
        return self

    def One(self) -> "Scalar":
        return self.SetInt64(1)

    def Set(self, a: "Scalar") -> "Scalar":
        self.v = _sc(a).v
        return self

    def Clone(self) -> "Scalar":
        return Scalar(self.v)

    def Equal(self, a: "Scalar") -> bool:
        return self._int() % ORDER == _sc(a)._int() % ORDER

    def _int(self) -> int:
        return int.from_bytes(self.v, "little")

    def _set(self, x: int) -> "Scalar":
        self.v = (x % ORDER).to_bytes(32, "little")
        return self

    # -- arithmetic mod l (host plumbing, like group/mod.Int)
    def Add(self, a, b):
        return self._set(_sc(a)._int() + _sc(b)._int())

    def Sub(self, a, b):
        return self._set(_sc(a)._int() - _sc(b)._int())

    def Neg(self, a):
        return self._set(-_sc(a)._int())

    def Mul(self, a, b):
        return self._set(_sc(a)._int() * _sc(b)._int())

    def Inv(self, a):
        return self._set(pow(_sc(a)._int() % ORDER, ORDER - 2, ORDER))

    def Div(self, a, b):
        return self._set(_sc(a)._int() * pow(_sc(b)._int() % ORDER, ORDER - 2, ORDER))

    def Pick(self, rand=None) -> "Scalar":
        raw = rand(64) if rand is not None else os.urandom(64)
        return self._set(int.from_bytes(raw, "little"))

    def String(self) -> str:
        return self.MarshalBinary().hex()

    __repr__ = String


def _sc(s) -> Scalar:
    if not isinstance(s, Scalar):
        raise TypeError("ErrTypeCast: not an edwards25519 scalar")
    return s


# -------------------------------------------------------- kyber.Point mirror
class Point:
    """kyber.Point for Ed25519 holding the canonical 32-byte encoding; every
    operation that needs curve arithmetic is one (batch-of-one) engine call."""

    __slots__ = ("enc", "var_time")

    def __init__(self, enc: bytes = _NULL_ENC):
        self.enc = bytes(enc)
        self.var_time = False

    def AllowVarTime(self, on: bool) -> None:  # point_vartime.go:9
        self.var_time = bool(on)

    def MarshalBinary(self) -> bytes:
        return self.enc

    def MarshalSize(self) -> int:
        return 32

    def UnmarshalBinary(self, b: bytes) -> "Point":
        if len(b) != 32:
            raise ValueError("invalid Ed25519 curve point")
        one = (1).to_bytes(32, "little")
        out, st = batch_mul(one, b)
        if st[0]:
            raise ValueError("invalid Ed25519 curve point")
        # the reference keeps the decoded point; its re-encoding is canonical
        self.enc = bytes(out[0])
        return self

    def Null(self) -> "Point":
        self.enc = _NULL_ENC
        return self

    def Base(self) -> "Point":
        self.enc = _BASE_ENC
        return self

    def Set(self, p: "Point") -> "Point":
        self.enc = _pt(p).enc
        return self

    def Clone(self) -> "Point":
        return Point(self.enc)

    def Equal(self, p: "Point") -> bool:
        return self.enc == _pt(p).enc

    def Add(self, a: "Point", b: "Point") -> "Point":
        """a + b through the engine's batch addition (the complete unified addition, ge.go:183)."""
        out, st = batch_add(_pt(a).enc, _pt(b).enc)
        if st.any():
            raise ValueError("invalid Ed25519 curve point")
        self.enc = bytes(out[0])
        return self

    def Neg(self, a: "Point") -> "Point":
        """-(x, y) = (-x, y): flip the sign bit of the encoding unless x = 0 (ge.go Neg: X and T negated)."""
        e = _pt(a).enc
        y = int.from_bytes(e, "little") & ((1 << 255) - 1)
        self.enc = e if y in (1, _P - 1) else e[:31] + bytes([e[31] ^ 0x80])
        return self

    def Sub(self, a: "Point", b: "Point") -> "Point":
        return self.Add(a, Point().Neg(b))

    def Pick(self, rand=None) -> "Point":
        """A random element of the prime-order subgroup, k * B.  (The reference's Pick embeds random
        data, point.go:177-233; its outputs are random too and only reproducible with Go's XOF stream.)"""
        return self.Mul(Scalar().Pick(rand), None)

    def Hash(self, m: bytes, dst: str | bytes) -> "Point":
        """kyber.HashablePoint (hash.go:13; point.go:325-334)."""
        d = dst.encode() if isinstance(dst, str) else bytes(dst)
        self.enc = bytes(batch_hash([bytes(m)], d)[0])
        return self

    def Mul(self, s: Scalar, A: "Point | None") -> "Point":
        a = _sc(s).v
        if A is None:
            self.enc = bytes(batch_mul_base(a)[0])
        else:
            out, st = batch_mul(a, _pt(A).enc, vartime=self.var_time)
            if st[0]:
                raise ValueError("invalid Ed25519 curve point")
            self.enc = bytes(out[0])
        return self

    def String(self) -> str:
        return self.enc.hex()

    __repr__ = String


def _pt(p) -> Point:
    if not isinstance(p, Point):
        raise TypeError("ErrTypeCast: not an edwards25519 point")
    return p


class Curve:
    """kyber.Group (group.go:175-183) for Ed25519."""

    def String(self) -> str:
        return "Ed25519"

    def ScalarLen(self) -> int:
        return SCALAR_LEN

    def Scalar(self) -> Scalar:
        return Scalar()

    def PointLen(self) -> int:
        return POINT_LEN

    def Point(self) -> Point:
        return Point()

    def NewKeyAndSeedWithInput(self, buffer: bytes):
        """curve.go:51-60: clamped, unreduced secret."""
        digest = bytearray(hashlib.sha512(buffer).digest())
        digest[0] &= 0xF8
        digest[31] &= 0x7F
        digest[31] |= 0x40
        return Scalar(bytes(digest[:32])), buffer, bytes(digest[32:])


def NewSuite() -> Curve:
    return Curve()
