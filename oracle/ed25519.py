"""CPU ORACLE (test infrastructure, not product code) -- Ed25519 group arithmetic.

Big-integer restatement of the observable behaviour of the reference's
``group/edwards25519`` package.  Only canonical 32-byte encodings are ever
observable through kyber.Point, so this oracle works in
affine coordinates with Python integers and the complete twisted-Edwards
addition law; every function cites the reference code whose *behaviour* it
restates.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline
leg may import this module.  The product path (``kyber_amd``) never does.

Pinned against: RFC 8032 vectors, the 1024 ``sign.input`` KATs, RFC 9380
edwards25519 hash-to-curve points and the small-order list held by the
reference's own tests (see tests/golden/README.md and
tests/test_oracle_ed25519.py).
"""
from __future__ import annotations

import hashlib

# group/edwards25519/const.go:12,15
P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
# curve -x^2 + y^2 = 1 + d x^2 y^2, d = -121665/121666 (ge.go:7-8, const.go:34)
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)  # const.go:42

IDENTITY = (0, 1)


def _inv(x: int) -> int:
    return pow(x, P - 2, P)


# base point (x, 4/5) with x "positive" (even)  -- const.go:51-56
_BY = (4 * _inv(5)) % P


def _recover_x(y: int, sign: int):
    """ge.go:110-150 FromBytes: x = u v^3 (u v^7)^((p-5)/8), then fix-ups."""
    u = (y * y - 1) % P
    v = (D * y * y + 1) % P
    x = (u * pow(v, 3, P) * pow(u * pow(v, 7, P) % P, (P - 5) // 8, P)) % P
    vxx = (v * x * x) % P
    if (vxx - u) % P != 0:
        if (vxx + u) % P != 0:
            return None
        x = (x * SQRT_M1) % P
    if (x & 1) != sign:
        x = (-x) % P  # note: x == 0 with sign == 1 stays 0 and is accepted
    return x


B = (_recover_x(_BY, 0), _BY)


def decode(s: bytes):
    """point.UnmarshalBinary (point.go:65-70) -> FromBytes (ge.go:110).

    Bit 255 is the sign; y is taken mod 2^255 and then mod p (non-canonical
    y >= p accepted, fe.go:81-110).  Returns affine (x, y) or None.
    """
    if len(s) != 32:
        return None
    n = int.from_bytes(s, "little")
    sign = n >> 255
    y = (n & ((1 << 255) - 1)) % P
    x = _recover_x(y, sign)
    if x is None:
        return None
    return (x, y)


def encode(pt) -> bytes:
    """point.MarshalBinary (point.go:54-58) -> ToBytes (ge.go:99-107)."""
    x, y = pt
    return (y | ((x & 1) << 255)).to_bytes(32, "little")


def add(p1, p2):
    """Complete addition law on -x^2+y^2=1+dx^2y^2 (what ge.go:183 computes)."""
    x1, y1 = p1
    x2, y2 = p2
    t = D * x1 * x2 % P * y1 % P * y2 % P
    x3 = (x1 * y2 + x2 * y1) * _inv(1 + t) % P
    y3 = (y1 * y2 + x1 * x2) * _inv(1 - t) % P
    return (x3, y3)


def neg(p):
    return ((-p[0]) % P, p[1])


# --- fast projective internals (extended coordinates) -----------------------
def _ext(p):
    return (p[0], p[1], 1, p[0] * p[1] % P)


def _ext_add(a, b):
    x1, y1, z1, t1 = a
    x2, y2, z2, t2 = b
    A = (y1 - x1) * (y2 - x2) % P
    Bq = (y1 + x1) * (y2 + x2) % P
    C = 2 * D * t1 % P * t2 % P
    Dq = 2 * z1 * z2 % P
    E, F, G, H = Bq - A, Dq - C, Dq + C, Bq + A
    return (E * F % P, G * H % P, F * G % P, E * H % P)


def _ext_dbl(a):
    x1, y1, z1, _ = a
    A = x1 * x1 % P
    Bq = y1 * y1 % P
    C = 2 * z1 * z1 % P
    H = A + Bq
    E = H - (x1 + y1) * (x1 + y1) % P
    G = A - Bq
    F = C + G
    return (E * F % P, G * H % P, F * G % P, E * H % P)


def _aff(a):
    zi = _inv(a[2])
    return (a[0] * zi % P, a[1] * zi % P)


def mul_int(k: int, pt):
    """k*pt for an arbitrary (possibly negative) integer k, pt affine."""
    if k < 0:
        return mul_int(-k, neg(pt))
    acc = (0, 1, 1, 0)
    q = _ext(pt)
    while k:
        if k & 1:
            acc = _ext_add(acc, q)
        q = _ext_dbl(q)
        k >>= 1
    return _aff(acc)


def recode_radix16(a: bytes):
    """Signed radix-16 recoding exactly as ge.go:374-390 / ge.go:453-467.

    Returns the 64 digits; e[0..62] in [-8,8], e[63] in [0,16].
    """
    e = []
    for v in a:
        e.append(v & 15)
        e.append((v >> 4) & 15)
    carry = 0
    for i in range(63):
        e[i] += carry
        carry = (e[i] + 8) >> 4
        e[i] -= carry << 4
    e[63] += carry
    return e


def effective_scalar_consttime(a: bytes) -> int:
    """Integer that geScalarMult / geScalarMultBase actually multiply by.

    For a < 2^255 this is just a.  For a[31] > 127 the top digit e[63] can
    exceed 8; selectCached/selectPreComputed (ge.go:352-365, 419-435) then
    match no table entry and contribute the identity, i.e. the digit is
    dropped (SURVEY.md section 8a edge case 1).
    """
    e = recode_radix16(a)
    if e[63] > 8:
        e[63] = 0
    return sum(d << (4 * i) for i, d in enumerate(e))


def mul_base(a: bytes) -> bytes:
    """point.Mul(s, nil) -> geScalarMultBase (point.go:243, ge.go:373)."""
    return encode(mul_int(effective_scalar_consttime(a), B))


def mul(a: bytes, pt_bytes: bytes, vartime: bool = False):
    """point.Mul(s, A) on wire formats; None when A does not decode.

    vartime=False: geScalarMult (ge.go:443) incl. its >=2^255 quirk.
    vartime=True : geScalarMultVartime (ge_mult_vartime.go:11) -- slide()
    handles all 256 bits, so the multiplier is the plain 256-bit integer.
    """
    pt = decode(pt_bytes)
    if pt is None:
        return None
    k = int.from_bytes(a, "little") if vartime else effective_scalar_consttime(a)
    return encode(mul_int(k, pt))


def msm(scalars, points):
    """Sum_i s_i * P_i with plain 256-bit integer scalars (what N x Mul + N x Add
    compute in share/poly.go:340-348, 449-476).  Returns encoding or None."""
    acc = (0, 1, 1, 0)
    for s, pb in zip(scalars, points):
        pt = decode(pb)
        if pt is None:
            return None
        k = effective_scalar_consttime(s)
        q = mul_int(k, pt)
        acc = _ext_add(acc, _ext(q))
    return encode(_aff(acc))


def clamp(h32: bytes) -> bytes:
    """curve.go:51-58 NewKeyAndSeedWithInput clamping (unreduced)."""
    b = bytearray(h32)
    b[0] &= 0xF8
    b[31] &= 0x7F
    b[31] |= 0x40
    return bytes(b)


def secret_scalar(seed: bytes) -> bytes:
    return clamp(hashlib.sha512(seed).digest()[:32])


def is_small_order(pt) -> bool:
    return mul_int(8, pt) == IDENTITY


def on_curve(pt) -> bool:
    x, y = pt
    return (-x * x + y * y - 1 - D * x * x * y * y) % P == 0


# --- RFC 9380 elligator2 (only used to replay the reference's hash-to-curve
# golden points, point_test.go:405-445, which pin Add + Mul-by-cofactor) ------
def _sqrt(a: int):
    """Square root mod p (p = 5 mod 8) or None."""
    a %= P
    r = pow(a, (P + 3) // 8, P)
    if r * r % P == a:
        return r
    r = r * SQRT_M1 % P
    if r * r % P == a:
        return r
    return None


def map_to_curve_elligator2(u: int):
    """RFC 9380 section 6.8.2 (curve25519 elligator2 + birational map), the
    behaviour of point.go:520-640 mapToCurveElligator2Ed25519."""
    J, Z = 486662, 2
    c1 = _sqrt(-486664)
    if c1 & 1:
        c1 = P - c1  # sgn0(c1) == 0
    den = (1 + Z * u * u) % P
    x1 = (-J * _inv(den)) % P if den else (-J) % P
    gx1 = (x1 * x1 % P * x1 + J * x1 * x1 + x1) % P
    y = _sqrt(gx1)
    if y is not None:
        xm = x1
        if (y & 1) != 1:
            y = P - y
    else:
        xm = (-x1 - J) % P
        gx2 = (xm * xm % P * xm + J * xm * xm + xm) % P
        y = _sqrt(gx2)
        if (y & 1) != 0:
            y = P - y
    ym = y
    # birational map Montgomery -> Edwards
    if ym == 0 or (xm + 1) % P == 0:
        return (0, 1)
    xe = c1 * xm % P * _inv(ym) % P
    ye = (xm - 1) * _inv(xm + 1) % P
    return (xe, ye)


def hash_to_curve_from_u(u0: int, u1: int):
    """point.go:325-334 Hash after hashToField: Q0 + Q1, times cofactor 8."""
    q = add(map_to_curve_elligator2(u0), map_to_curve_elligator2(u1))
    return mul_int(8, q)


def expand_message_xmd_sha512(msg: bytes, dst: bytes, n: int) -> bytes:
    """point.go:362-430 expandMessageXMD with SHA-512 (RFC 9380 section 5.3.1; b_in_bytes 64, s_in_bytes 128)."""
    import struct

    ell = (n + 63) // 64
    if ell > 255 or n > 65535 or not dst or len(dst) > 255:
        raise ValueError("invalid parameters")
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha512(bytes(128) + msg + struct.pack(">H", n) + b"\x00" + dst_prime).digest()
    b = [hashlib.sha512(b0 + b"\x01" + dst_prime).digest()]
    for i in range(2, ell + 1):
        b.append(hashlib.sha512(bytes(x ^ y for x, y in zip(b0, b[-1])) + bytes([i]) + dst_prime).digest())
    return b"".join(b)[:n]


def hash_to_field(msg: bytes, dst: bytes):
    """point.go:336-360 hashToField(m, dst, 2): two 48-byte big-endian integers mod p."""
    ub = expand_message_xmd_sha512(msg, dst, 96)
    return int.from_bytes(ub[:48], "big") % P, int.from_bytes(ub[48:], "big") % P


def hash_to_curve(msg: bytes, dst: bytes) -> bytes:
    """(*point).Hash (point.go:325-334): 32-byte encoding of clear_cofactor(map(u0) + map(u1))."""
    u0, u1 = hash_to_field(msg, dst)
    return encode(hash_to_curve_from_u(u0, u1))
