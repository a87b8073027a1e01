"""CPU ORACLE (test infrastructure, not product code) -- the reference's in-tree bn256 suite.

Big-integer restatement of pairing/bn256 (dclxvi parameters, NOT Ethereum's alt_bn128):
  constants.go:17-25      u, p, Order;  xi = i + 3 (gfp2.go / gfp6.go:11)
  curve.go:12-24,189-203  G1: y^2 = x^3 + 3, generator (1, -2), Mul = double-and-add
  twist.go:12-33,162      G2: y^2 = x^3 + 3/xi over Fp2, generator twistGen
  optate.go:5-115         lineFunctionAdd / lineFunctionDouble / mulLine   (restated formula by formula)
  optate.go:117-213       sixuPlus2NAF, miller (incl. the Q1 / -Q2 Frobenius steps)
  optate.go:215-264       finalExponentiation (same addition chain)
  point.go:170-238        G1 wire format (x || y, 32-byte big-endian each, infinity = 64 zero bytes,
                          coordinates reduced mod p on input, on-curve check only)
  point.go:423-499        G2 wire format (x.x || x.y || y.x || y.y with gfP2{x, y} = x i + y)
  point.go:630-662        GT wire format (12 x 32 bytes, x.x.x ... y.z.y)
  point.go:261-313        Hash: SHA-256 try-and-increment, y = t^((p+1)/4)
  hash.go:10-110, gfp.go:46-68,137-164   HashG1: HKDF-SHA-256 to the base field, then the Shallue-van de Woestijne
                          map in the reference's own arrangement (x1, x2, x3 in this order, legendre == 1, sign0)

Element conventions here: Fp2 = (real, imag); an Fp12 element is the list [a_0..a_5] of Fp2
coefficients of w^k (w = omega, w^2 = tau, w^6 = xi), i.e. gfP12{x, y} with gfP6{x, y, z} maps to
a_0 = y.z, a_2 = y.y, a_4 = y.x, a_1 = x.z, a_3 = x.y, a_5 = x.x.

Pinned against (tests/test_oracle_bn256.py, fixtures in tests/golden/bn256.json): the BDN fixtures
sign/bdn/bdn_vartime_test.go:24-48 (coefficients + aggregated G2 key) and :90-135 (3 private
scalars -> G2 public keys, G1 signatures on a fixed message through Hash), the two Hash outputs
of pairing/bn256/point_test.go:13-45, bilinearity (suite_test.go:231-259) and an independent
textbook pairing (affine Miller loop on E(Fp12), exponent (p^12-1)/Order as one big power).
The reference holds no fixed KAT for pairing output bytes; following optate.go formula by
formula + the checks above is the strongest pin available.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import hashlib

U = 6518589491078791937
P = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
ORDER = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1
assert P == 65000549695646603732796438742359905742825358107623003571877145026864184071783
assert ORDER == 65000549695646603732796438742359905742570406053903786389881062969044166799969

F2_ZERO, F2_ONE = (0, 0), (1, 0)
XI = (3, 1)


def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return (-a[0] % P, -a[1] % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_sqr(a): return f2_mul(a, a)
def f2_conj(a): return (a[0], -a[1] % P)
def f2_muls(a, s): return (a[0] * s % P, a[1] * s % P)


def f2_inv(a):
    # gfP2.Invert (gfp2.go:146-161) over gfP.Invert = f^(p-2) (gfp.go): the inverse of zero is zero, not an error
    n = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
    return (a[0] * n % P, -a[1] * n % P)


def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_mul(a, a)
        e >>= 1
    return r


def fp_sqrt(a):
    a %= P
    s = pow(a, (P + 1) // 4, P)  # p = 3 mod 4
    return s if s * s % P == a else None


def f2_sqrt(a):
    """Any square root of a in Fp2, or None."""
    a0, a1 = a
    if a1 == 0:
        s = fp_sqrt(a0)
        if s is not None:
            return (s, 0)
        return (0, fp_sqrt(-a0))
    n = fp_sqrt(a0 * a0 + a1 * a1)
    if n is None:
        return None
    inv2 = pow(2, -1, P)
    for s in (n, -n):
        x0 = fp_sqrt((a0 + s) * inv2)
        if x0:
            x1 = a1 * pow(2 * x0, -1, P) % P
            if f2_mul((x0, x1), (x0, x1)) == (a0 % P, a1 % P):
                return (x0, x1)
    return None


TWIST_B = f2_mul((3, 0), f2_inv(XI))  # twist.go:16-19
G1_GEN = (1, P - 2)  # curve.go:19-24
G2_GEN = (  # twist.go:21-33, de-Montgomerised; (real, imag) per coordinate
    (0x8F25386F72C9462B81597D65AE2092C4B97792155DCDAAD32B8A6DD41792534C,
     0x2ECCA446FF6F3D4D03C76E9B5C752F28BC37B364CB05AC4A37EB32E1C3245970),
    (0x274E5747E8CAFACC3716CC8699DB79B22F0E4FF3C23E898F694420A3BE3087A5,
     0x2DB10EF5233B0FE3962B9EE6A4BBC2B5BDE01A54F3513D42DF972E128F31BF12),
)

# ------------------------------------------------------------------ Fp12 (w-basis)
F12_ONE = [F2_ONE] + [F2_ZERO] * 5


def f12_mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        if a[i] == F2_ZERO:
            continue
        for j in range(6):
            if b[j] == F2_ZERO:
                continue
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    return [f2_add(t[k], f2_mul(t[k + 6], XI)) if k < 5 else t[k] for k in range(6)]


def f12_sqr(a): return f12_mul(a, a)
def f12_conj(a): return [a[k] if k % 2 == 0 else f2_neg(a[k]) for k in range(6)]


GAMMA = [f2_pow(XI, k * (P - 1) // 6) for k in range(6)]


def f12_frob(a, n=1):
    """gfP12.Frobenius (gfp12.go:124) applied n times."""
    for _ in range(n):
        a = [f2_mul(f2_conj(a[k]), GAMMA[k]) for k in range(6)]
    return a


def f12_pow(a, e):
    """gfP12.Exp (gfp12.go:177-192)."""
    r = list(F12_ONE)
    for bit in bin(e)[2:] if e else "":
        r = f12_sqr(r)
        if bit == "1":
            r = f12_mul(r, a)
    return r


def _f6_mul(a, b):
    t = [F2_ZERO] * 5
    for i in range(3):
        for j in range(3):
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    return [f2_add(t[0], f2_mul(t[3], XI)), f2_add(t[1], f2_mul(t[4], XI)), t[2]]


def _f6_inv(a):
    a0, a1, a2 = a
    t0 = f2_sub(f2_sqr(a0), f2_mul(XI, f2_mul(a1, a2)))
    t1 = f2_sub(f2_mul(XI, f2_sqr(a2)), f2_mul(a0, a1))
    t2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    d = f2_add(f2_mul(a0, t0), f2_mul(XI, f2_add(f2_mul(a2, t1), f2_mul(a1, t2))))
    di = f2_inv(d)
    return [f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di)]


def f12_inv(a):
    c0, c1 = [a[0], a[2], a[4]], [a[1], a[3], a[5]]
    sq1 = _f6_mul(c1, c1)
    t = [f2_sub(x, y) for x, y in zip(_f6_mul(c0, c0), [f2_mul(sq1[2], XI), sq1[0], sq1[1]])]
    ti = _f6_inv(t)
    r0 = _f6_mul(c0, ti)
    r1 = [f2_neg(x) for x in _f6_mul(c1, ti)]
    return [r0[0], r1[0], r0[1], r1[1], r0[2], r1[2]]


# ------------------------------------------------------- curves (affine, None = infinity)
class _F1:
    zero = 0
    add = staticmethod(lambda a, b: (a + b) % P)
    sub = staticmethod(lambda a, b: (a - b) % P)
    mul = staticmethod(lambda a, b: a * b % P)
    neg = staticmethod(lambda a: -a % P)
    inv = staticmethod(lambda a: pow(a, -1, P))
    b = 3


class _F2:
    zero = F2_ZERO
    add, sub, mul, neg, inv = map(staticmethod, (f2_add, f2_sub, f2_mul, f2_neg, f2_inv))
    b = TWIST_B


def _ec_add(F, p, q):
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if F.add(y1, y2) == F.zero:
            return None
        lam = F.mul(F.mul(F.add(F.add(x1, x1), x1), x1), F.inv(F.add(y1, y1)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    return (x3, F.sub(F.mul(lam, F.sub(x1, x3)), y1))


def _ec_mul(F, k, p):
    """curvePoint.Mul / twistPoint.Mul (curve.go:189-203, twist.go:162): k >= 0 plain integer."""
    r = None
    for bit in bin(k)[2:] if k else "":
        r = _ec_add(F, r, r)
        if bit == "1":
            r = _ec_add(F, r, p)
    return r


def _on_curve(F, p):
    if p is None:
        return True
    return F.mul(p[1], p[1]) == F.add(F.mul(F.mul(p[0], p[0]), p[0]), F.b)


def g1_add(p, q): return _ec_add(_F1, p, q)
def g1_neg(p): return None if p is None else (p[0], -p[1] % P)
def g1_mul(k, p): return _ec_mul(_F1, k, p)
def g1_on_curve(p): return _on_curve(_F1, p)
def g2_add(p, q): return _ec_add(_F2, p, q)
def g2_neg(p): return None if p is None else (p[0], f2_neg(p[1]))
def g2_mul(k, p): return _ec_mul(_F2, k, p)
def g2_on_curve(p): return _on_curve(_F2, p)


# -------------------------------------------------------------------- wire formats
class DecodeError(ValueError):
    pass


def _be(x): return x.to_bytes(32, "big")


def g1_marshal(p) -> bytes:
    """pointG1.MarshalBinary (point.go:170-192)."""
    return bytes(64) if p is None else _be(p[0]) + _be(p[1])


def g1_unmarshal(buf: bytes):
    """pointG1.UnmarshalBinary (point.go:206-238): coordinates are reduced mod p by montEncode,
    (0, 0) is infinity, otherwise on-curve check only."""
    if len(buf) < 64:
        raise DecodeError("bn256.G1: not enough data")
    x, y = int.from_bytes(buf[:32], "big") % P, int.from_bytes(buf[32:64], "big") % P
    if x == 0 and y == 0:
        return None
    if not g1_on_curve((x, y)):
        raise DecodeError("bn256.G1: malformed point")
    return (x, y)


def g2_marshal(p) -> bytes:
    """pointG2.MarshalBinary (point.go:423-452): x.x, x.y, y.x, y.y with gfP2{x, y} = x i + y."""
    if p is None:
        return bytes(128)
    (x0, x1), (y0, y1) = p
    return _be(x1) + _be(x0) + _be(y1) + _be(y0)


def g2_unmarshal(buf: bytes):
    """pointG2.UnmarshalBinary (point.go:466-499): on-curve only, never a subgroup check."""
    if len(buf) < 128:
        raise DecodeError("bn256.G2: not enough data")
    v = [int.from_bytes(buf[32 * i:32 * i + 32], "big") % P for i in range(4)]
    x, y = (v[1], v[0]), (v[3], v[2])
    if x == F2_ZERO and y == F2_ZERO:
        return None
    if not g2_on_curve((x, y)):
        raise DecodeError("bn256.G2: malformed point")
    return (x, y)


def gt_marshal(a) -> bytes:
    """pointGT.MarshalBinary (point.go:630-662): x.x.x, x.x.y, x.y.x, ... y.z.y."""
    out = b""
    for half in (1, 0):  # gfP12.x (omega coefficient) first
        for m in (2, 1, 0):  # gfP6 x (tau^2), y (tau), z
            c = a[2 * m + half]
            out += _be(c[1]) + _be(c[0])
    return out


def gt_unmarshal(buf: bytes):
    """pointGT.UnmarshalBinary (point.go:664-716): coefficients reduced mod p, no membership check."""
    if len(buf) < 384:
        raise DecodeError("bn256.GT: not enough data")
    a = [None] * 6
    k = 0
    for half in (1, 0):
        for m in (2, 1, 0):
            im = int.from_bytes(buf[64 * k:64 * k + 32], "big") % P
            re = int.from_bytes(buf[64 * k + 32:64 * k + 64], "big") % P
            a[2 * m + half] = (re, im)
            k += 1
    return a


def gt_mul_bytes(scalar_be: bytes, gt: bytes) -> bytes:
    """pointGT.Mul (point.go:613-628) -> gfP12.Exp (gfp12.go:177-192)."""
    return gt_marshal(f12_pow(gt_unmarshal(gt), int.from_bytes(scalar_be, "big")))


def hash_to_g1(m: bytes):
    """pointG1.Hash -> hashToPoint (point.go:261-313)."""
    x = int.from_bytes(hashlib.sha256(m).digest(), "big") % P
    while True:
        t = (x * x * x + 3) % P
        y = pow(t, (P + 1) // 4, P)  # big.Int.ModSqrt for p = 3 mod 4
        if y * y % P == t:
            return (x, y)
        x = (x + 1) % P


# s = sqrt(-3) as the reference chose it and (s - 1) / 2 (constants.go:104-108, de-Montgomerised; tests/golden/bn256.json
# holds the extraction): which root is data -- the other root gives different points
SVDW_S = 0x196AC037F07E9F9F10EFB671F725F42C373FE702B4D85525D
SVDW_S_MINUS_1_OVER_2 = (SVDW_S - 1) * pow(2, -1, P) % P
assert SVDW_S * SVDW_S % P == P - 3


def hash_to_base(msg: bytes, dst: bytes = b"") -> int:
    """hashToBase (gfp.go:46-68): 48 bytes of HKDF-SHA-256 (secret = msg, salt = dst, info = "H2C" 0x00 0x01) as a
    big-endian integer mod p.  HKDF (RFC 5869): an empty salt is 32 zero bytes."""
    import hmac

    prk = hmac.new(dst if dst else bytes(32), msg, hashlib.sha256).digest()
    info = b"H2C\x00\x01"
    t1 = hmac.new(prk, info + b"\x01", hashlib.sha256).digest()
    t2 = hmac.new(prk, t1 + info + b"\x02", hashlib.sha256).digest()
    return int.from_bytes((t1 + t2)[:48], "big") % P


def _sign0(x: int) -> int:
    """sign0 (gfp.go:137-148): 1 when x >= (p - 1) / 2 ... the comparison falls through to 1 on equality"""
    h = (P - 1) // 2
    return 1 if x > h else (-1 if x < h else 1)


def _legendre(x: int) -> int:
    """legendre (gfp.go:150-164): x^((p - 1) / 2) as 2 (f & 1) - 1 on the residue, 0 for 0"""
    f = pow(x, (P - 1) // 2, P)
    return 0 if f == 0 else 2 * (f & 1) - 1


def map_to_curve(t: int):
    """mapToCurve (hash.go:14-110), statement by statement"""
    a = (3 + t * t + 1) % P
    st = SVDW_S * t % P
    w0 = pow(st * a % P, P - 2, P)  # gfP.Invert = f^(p - 2): 0 stays 0
    w = st * st % P * w0 % P
    e = _sign0(t)

    def finish(x):
        y = pow((x * x * x + 3) % P, (P + 1) // 4, P)
        if e != _sign0(y):
            y = -y % P
        return (x, y)

    x1 = (SVDW_S_MINUS_1_OVER_2 - t * w) % P
    if _legendre((x1 * x1 * x1 + 3) % P) == 1:
        return finish(x1)
    x2 = (-1 - x1) % P
    if _legendre((x2 * x2 * x2 + 3) % P) == 1:
        return finish(x2)
    x3 = (pow(a, 4, P) * w0 % P * w0 + 1) % P
    return finish(x3)


def hash_g1_svdw(msg: bytes, dst: bytes = b""):
    """HashG1 (hash.go:10-12)"""
    return map_to_curve(hash_to_base(msg, dst))


def g1_mul_bytes(scalar_be: bytes, pt: bytes) -> bytes:
    return g1_marshal(g1_mul(int.from_bytes(scalar_be, "big"), g1_unmarshal(pt)))


def g2_mul_bytes(scalar_be: bytes, pt: bytes) -> bytes:
    return g2_marshal(g2_mul(int.from_bytes(scalar_be, "big"), g2_unmarshal(pt)))


# ------------------------------------------------------------- pairing (optate.go restated)
SIXU_PLUS_2_NAF = [  # optate.go:117-122
    0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, -1, 0, 1, 0,
    1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, -1,
    0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, -1, 0, 0, 0,
    0, 1, 0, 0, 0, 1,
]
assert sum(d << i for i, d in enumerate(SIXU_PLUS_2_NAF)) == 6 * U + 2


def _line_add(r, p, q, r2):
    """lineFunctionAdd (optate.go:5-52).  r = (x, y, z, t) Jacobian twist point with t = z^2,
    p = affine twist point (x, y), q = affine curve point (x, y), r2 = p.y^2."""
    rx, ry, rz, rt = r
    B = f2_mul(p[0], rt)
    D = f2_add(p[1], rz)
    D = f2_mul(f2_sub(f2_sub(f2_sqr(D), r2), rt), rt)
    H = f2_sub(B, rx)
    I = f2_sqr(H)
    E = f2_add(I, I)
    E = f2_add(E, E)
    J = f2_mul(H, E)
    L1 = f2_sub(f2_sub(D, ry), ry)
    V = f2_mul(rx, E)
    ox = f2_sub(f2_sub(f2_sub(f2_sqr(L1), J), V), V)
    oz = f2_sub(f2_sub(f2_sqr(f2_add(rz, H)), rt), I)
    t = f2_mul(f2_sub(V, ox), L1)
    t2 = f2_mul(ry, J)
    t2 = f2_add(t2, t2)
    oy = f2_sub(t, t2)
    ot = f2_sqr(oz)
    t = f2_sub(f2_sub(f2_sqr(f2_add(p[1], oz)), r2), ot)
    t2 = f2_mul(L1, p[0])
    t2 = f2_add(t2, t2)
    a = f2_sub(t2, t)
    c = f2_muls(oz, q[1])
    c = f2_add(c, c)
    b = f2_muls(f2_neg(L1), q[0])
    b = f2_add(b, b)
    return a, b, c, (ox, oy, oz, ot)


def _line_double(r, q):
    """lineFunctionDouble (optate.go:54-94)."""
    rx, ry, rz, rt = r
    A = f2_sqr(rx)
    B = f2_sqr(ry)
    C = f2_sqr(B)
    D = f2_sub(f2_sub(f2_sqr(f2_add(rx, B)), A), C)
    D = f2_add(D, D)
    E = f2_add(f2_add(A, A), A)
    G = f2_sqr(E)
    ox = f2_sub(f2_sub(G, D), D)
    oz = f2_sub(f2_sub(f2_sqr(f2_add(ry, rz)), B), rt)
    oy = f2_mul(f2_sub(D, ox), E)
    t = f2_add(C, C)
    t = f2_add(t, t)
    t = f2_add(t, t)
    oy = f2_sub(oy, t)
    ot = f2_sqr(oz)
    t = f2_mul(E, rt)
    t = f2_add(t, t)
    b = f2_muls(f2_neg(t), q[0])
    a = f2_sub(f2_sub(f2_sqr(f2_add(rx, E)), A), G)
    t = f2_add(B, B)
    t = f2_add(t, t)
    a = f2_sub(a, t)
    c = f2_mul(oz, rt)
    c = f2_muls(f2_add(c, c), q[1])
    return a, b, c, (ox, oy, oz, ot)


def _mul_line(ret, a, b, c):
    """mulLine (optate.go:96-115): ret *= (a tau + b) omega + c  ==  c + b w + a w^3."""
    line = [c, b, F2_ZERO, a, F2_ZERO, F2_ZERO]
    return f12_mul(ret, line)


def miller(q, p):
    """miller (optate.go:126-213); q affine twist point, p affine curve point (both finite)."""
    ret = list(F12_ONE)
    minus_a = (q[0], f2_neg(q[1]))
    r = (q[0], q[1], F2_ONE, F2_ONE)
    r2 = f2_sqr(q[1])
    n = len(SIXU_PLUS_2_NAF)
    for i in range(n - 1, 0, -1):
        a, b, c, new_r = _line_double(r, p)
        if i != n - 1:
            ret = f12_sqr(ret)
        ret = _mul_line(ret, a, b, c)
        r = new_r
        d = SIXU_PLUS_2_NAF[i - 1]
        if d == 1:
            a, b, c, new_r = _line_add(r, q, p, r2)
        elif d == -1:
            a, b, c, new_r = _line_add(r, minus_a, p, r2)
        else:
            continue
        ret = _mul_line(ret, a, b, c)
        r = new_r
    q1 = (f2_mul(f2_conj(q[0]), f2_pow(XI, (P - 1) // 3)), f2_mul(f2_conj(q[1]), f2_pow(XI, (P - 1) // 2)))
    minus_q2 = (f2_muls(q[0], f2_pow(XI, (P * P - 1) // 3)[0]), q[1])
    assert f2_pow(XI, (P * P - 1) // 3)[1] == 0
    r2 = f2_sqr(q1[1])
    a, b, c, new_r = _line_add(r, q1, p, r2)
    ret = _mul_line(ret, a, b, c)
    r = new_r
    r2 = f2_sqr(minus_q2[1])
    a, b, c, _ = _line_add(r, minus_q2, p, r2)
    return _mul_line(ret, a, b, c)


def final_exponentiation(inp):
    """finalExponentiation (optate.go:215-264), same chain."""
    t1 = f12_mul(f12_conj(inp), f12_inv(inp))
    t2 = f12_frob(t1, 2)
    t1 = f12_mul(t1, t2)
    fp = f12_frob(t1)
    fp2 = f12_frob(t1, 2)
    fp3 = f12_frob(fp2)
    fu = f12_pow(t1, U)
    fu2 = f12_pow(fu, U)
    fu3 = f12_pow(fu2, U)
    y3 = f12_frob(fu)
    fu2p = f12_frob(fu2)
    fu3p = f12_frob(fu3)
    y2 = f12_frob(fu2, 2)
    y0 = f12_mul(f12_mul(fp, fp2), fp3)
    y1 = f12_conj(t1)
    y5 = f12_conj(fu2)
    y3 = f12_conj(y3)
    y4 = f12_conj(f12_mul(fu, fu2p))
    y6 = f12_conj(f12_mul(fu3, fu3p))
    t0 = f12_mul(f12_mul(f12_sqr(y6), y4), y5)
    t1 = f12_mul(f12_mul(y3, y5), t0)
    t0 = f12_mul(t0, y2)
    t1 = f12_sqr(f12_mul(f12_sqr(t1), t0))
    t0 = f12_mul(t1, y1)
    t1 = f12_mul(t1, y0)
    return f12_mul(f12_sqr(t0), t1)


def pair(p, q):
    """Suite.Pair -> optimalAte (suite.go:97, point.go:755, optate.go:266-274): p in G1, q in G2."""
    if p is None or q is None:
        return list(F12_ONE)
    return final_exponentiation(miller(q, p))


def pair_bytes(g1: bytes, g2: bytes) -> bytes:
    return gt_marshal(pair(g1_unmarshal(g1), g2_unmarshal(g2)))


def validate_pairing(p1, p2, inv1, inv2) -> bool:
    """Suite.ValidatePairing (suite.go:105-107): two full pairings + Equal."""
    return pair(p1, p2) == pair(inv1, inv2)


# ------------------------------------------------- independent textbook cross-check
def pair_textbook(p, q):
    """Optimal ate pairing the slow way: untwist Q to E(Fp12), affine Miller loop for 6u+2 plus the
    two Frobenius lines, then ONE big power (p^12-1)/Order.  Used only to check the restatement."""
    if p is None or q is None:
        return list(F12_ONE)

    def emb(a): return [(a % P, 0)] + [F2_ZERO] * 5
    def add(a, b): return [f2_add(x, y) for x, y in zip(a, b)]
    def sub(a, b): return [f2_sub(x, y) for x, y in zip(a, b)]

    def untwist(t):  # (x', y') -> (x' w^2, y' w^3)
        x, y = [F2_ZERO] * 6, [F2_ZERO] * 6
        x[2], y[3] = t[0], t[1]
        return x, y

    px, py = emb(p[0]), emb(p[1])
    qx, qy = untwist(q)
    tx, ty = qx, qy
    f = list(F12_ONE)

    def step(f, tx, ty, ax, ay):
        """f *= line through (tx,ty),(ax,ay) at P; returns updated f and the sum point."""
        if tx == ax and ty == ay:
            lam = f12_mul(f12_mul(emb(3), f12_sqr(tx)), f12_inv(f12_mul(emb(2), ty)))
        else:
            lam = f12_mul(sub(ay, ty), f12_inv(sub(ax, tx)))
        f = f12_mul(f, sub(sub(py, ty), f12_mul(lam, sub(px, tx))))
        nx = sub(sub(f12_sqr(lam), tx), ax)
        ny = sub(f12_mul(lam, sub(tx, nx)), ty)
        return f, nx, ny

    s = 6 * U + 2
    for bit in bin(s)[3:]:
        f = f12_sqr(f)
        f, tx, ty = step(f, tx, ty, tx, ty)
        if bit == "1":
            f, tx, ty = step(f, tx, ty, qx, qy)
    q1 = (f2_mul(f2_conj(q[0]), f2_pow(XI, (P - 1) // 3)), f2_mul(f2_conj(q[1]), f2_pow(XI, (P - 1) // 2)))
    mq2 = (f2_muls(q[0], f2_pow(XI, (P * P - 1) // 3)[0]), q[1])
    ax, ay = untwist(q1)
    f, tx, ty = step(f, tx, ty, ax, ay)
    ax, ay = untwist(mq2)
    f, tx, ty = step(f, tx, ty, ax, ay)
    return f12_pow(f, (P**12 - 1) // ORDER)
