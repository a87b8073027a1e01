"""CPU ORACLE (test infrastructure, not product code) -- the reference's in-tree bn254 suite.

Big-integer restatement of pairing/bn254 (Ethereum's alt_bn128 parameters); the package is the reference's bn256 code
with other constants and these differences, each restated here:
  constants.go:17-27      u, p, Order;  xi = i + 9 (gfp2.go:104-121)
  curve.go:13-23          G1: y^2 = x^3 + 3, generator (1, 2); Mul = GLV lattice walk (curve.go:196-222, lattice.go):
                          a different addition chain for the same multiple
  twist.go:16-33          G2: y^2 = x^3 + 3/xi over Fp2, generator twistGen
  twist.go:47-66          twistPoint.IsOnCurve ALSO requires [Order]Q = infinity: G2 UnmarshalBinary rejects points outside
                          the subgroup (bn256 does not)
  gfp.go:101-118          gfP.Unmarshal rejects coordinates >= p (bn256 reduces them)
  optate.go:117-120       sixuPlus2NAF (a signed-digit form of 6u + 2, not a NAF: adjacent non-zero digits)
  optate.go:124-269       miller / finalExponentiation / optimalAte: bn256's, formula by formula
  suite.go:134-140        ValidatePairing: two full pairings + Equal
  point.go:127-213        G1 wire format (x || y, 32-byte big-endian each, infinity = 64 zero bytes)
  point.go:215-341        Hash: RFC 9380 hash_to_curve with expand_message_xmd over legacy Keccak-256 and the
                          Shallue-van de Woestijne map (constants.go:72-84); no cofactor (G1 has prime order)
  point.go:431-520        G2 wire format (x.x || x.y || y.x || y.y with gfP2{x, y} = x i + y)
  point.go:617-735        GT wire format (12 x 32 bytes, x.x.x ... y.z.y); coefficients >= p rejected

Element conventions as in oracle/bn256.py: Fp2 = (real, imag); an Fp12 element is the list [a_0..a_5] of Fp2
coefficients of w^k (w^6 = xi).

Pinned against (tests/test_oracle_bn254.py, fixtures in tests/golden/bn254.json written by
tests/golden/make_golden_bn254.py from the reference tree): the hash_to_field / map_to_point / hash_to_point /
expand_message vectors of test_vectors_test.go and point_test.go:14-98, the generators (curve.go:19-23, twist.go:21-33,
de-Montgomerised), bilinearity (suite_test.go:240-251) and an independent textbook pairing (affine Miller loop on
E(Fp12), exponent (p^12-1)/Order as one big power).  The reference holds no fixed KAT for pairing output bytes.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

U = 4965661367192848881
P = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
ORDER = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1
assert P == 21888242871839275222246405745257275088696311157297823662689037894645226208583  # constants.go:27
assert ORDER == 21888242871839275222246405745257275088548364400416034343698204186575808495617  # constants.go:23

F2_ZERO, F2_ONE = (0, 0), (1, 0)
XI = (9, 1)  # gfp2.go:104-121


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
G1_GEN = (1, 2)  # curve.go:19-23
G2_GEN = (  # twist.go:21-33, de-Montgomerised; (real, imag) per coordinate
    (0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
     0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2),
    (0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
     0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B),
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
    """gfP12.Frobenius (gfp12.go:62-72) applied n times."""
    for _ in range(n):
        a = [f2_mul(f2_conj(a[k]), GAMMA[k]) for k in range(6)]
    return a


def f12_pow(a, e):
    """gfP12.Exp (gfp12.go:115-130)."""
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
    """curvePoint.Mul / twistPoint.Mul (curve.go:196-222, twist.go:170-183): the multiple [k]P, k >= 0 plain integer (the
    reference walks a GLV lattice decomposition on G1; the group element is the same)."""
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


def _coord(buf: bytes) -> int:
    """gfP.Unmarshal (gfp.go:101-118): a coordinate >= p is an error."""
    v = int.from_bytes(buf, "big")
    if v >= P:
        raise DecodeError("bn254: coordinate exceeds modulus" if v > P else "bn254: coordinate equals modulus")
    return v


def g1_marshal(p) -> bytes:
    """pointG1.MarshalBinary (point.go:127-147)."""
    return bytes(64) if p is None else _be(p[0]) + _be(p[1])


def g1_unmarshal(buf: bytes):
    """pointG1.UnmarshalBinary (point.go:161-200): coordinates < p, (0, 0) is infinity, otherwise on-curve
    (G1 is the whole curve: no subgroup check needed)."""
    if len(buf) < 64:
        raise DecodeError("bn254.G1: not enough data")
    x, y = _coord(buf[:32]), _coord(buf[32:64])
    if x == 0 and y == 0:
        return None
    if not g1_on_curve((x, y)):
        raise DecodeError("bn254.G1: malformed point")
    return (x, y)


def g2_marshal(p) -> bytes:
    """pointG2.MarshalBinary (point.go:431-460): x.x, x.y, y.x, y.y with gfP2{x, y} = x i + y."""
    if p is None:
        return bytes(128)
    (x0, x1), (y0, y1) = p
    return _be(x1) + _be(x0) + _be(y1) + _be(y0)


def g2_in_subgroup(p) -> bool:
    """twistPoint.IsOnCurve's second half (twist.go:62-65): [Order]Q = infinity."""
    return g2_mul(ORDER, p) is None


def g2_psi(q):
    """psi = twist^-1 o Frobenius o twist on E'(Fp2) (the Q1 of miller, optate.go:180-186): acts as [p] on G2."""
    if q is None:
        return None
    return (f2_mul(f2_conj(q[0]), f2_pow(XI, (P - 1) // 3)), f2_mul(f2_conj(q[1]), f2_pow(XI, (P - 1) // 2)))


G2_COFACTOR_PRIMES = (10069, 5864401, 1875725156269, 197620364512881247228717050342013327560683201906968909)


def g2_in_subgroup_fast(q) -> bool:
    """The criterion the device library uses instead of the 254-bit ladder: [u+1]Q + psi([u]Q) + psi^2([u]Q) ==
    psi^3([2u]Q).  Equivalent to g2_in_subgroup on every point of the twist: #E'(Fp2) = n (2p - n) with 2p - n the
    product of the four distinct primes above (none dividing n), so the group is cyclic, psi acts on each prime
    component as a root of X^2 - t X + p, and (u+1) + u X + u X^2 - 2u X^3 vanishes at that root only on the
    n-component (tests/test_oracle_bn254.py, tests/test_constants.py)."""
    xq = g2_mul(U, q)
    b = g2_psi(xq)
    lhs = g2_add(g2_add(g2_add(xq, q), b), g2_psi(b))
    rhs = g2_psi(g2_psi(b))
    return lhs == g2_add(rhs, rhs)


def g2_unmarshal(buf: bytes):
    """pointG2.UnmarshalBinary (point.go:474-520): coordinates < p, all-zero = infinity, otherwise on the twist AND in
    the order-n subgroup (twist.go:47-66)."""
    if len(buf) < 128:
        raise DecodeError("bn254.G2: not enough data")
    v = [_coord(buf[32 * i:32 * i + 32]) for i in range(4)]
    x, y = (v[1], v[0]), (v[3], v[2])
    if x == F2_ZERO and y == F2_ZERO:
        return None
    if not g2_on_curve((x, y)) or not g2_in_subgroup((x, y)):
        raise DecodeError("bn254.G2: malformed point")
    return (x, y)


def gt_marshal(a) -> bytes:
    """pointGT.MarshalBinary (point.go:617-651): x.x.x, x.x.y, x.y.x, ... y.z.y."""
    out = b""
    for half in (1, 0):  # gfP12.x (omega coefficient) first
        for m in (2, 1, 0):  # gfP6 x (tau^2), y (tau), z
            c = a[2 * m + half]
            out += _be(c[1]) + _be(c[0])
    return out


def gt_unmarshal(buf: bytes):
    """pointGT.UnmarshalBinary (point.go:662-735): coefficients < p (gfP.Unmarshal), no membership check."""
    if len(buf) < 384:
        raise DecodeError("bn254.GT: not enough data")
    a = [None] * 6
    k = 0
    for half in (1, 0):
        for m in (2, 1, 0):
            im = _coord(buf[64 * k:64 * k + 32])
            re = _coord(buf[64 * k + 32:64 * k + 64])
            a[2 * m + half] = (re, im)
            k += 1
    return a


def gt_mul_bytes(scalar_be: bytes, gt: bytes) -> bytes:
    """pointGT.Mul (point.go:606-615) -> gfP12.Exp (gfp12.go:115-130)."""
    return gt_marshal(f12_pow(gt_unmarshal(gt), int.from_bytes(scalar_be, "big")))


# ---- legacy Keccak-256 (sha3.NewLegacyKeccak256: padding 0x01, not SHA-3's 0x06)
_KRC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
        0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
        0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
        0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
        0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_KROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(v, n): return ((v << n) | (v >> (64 - n))) & _M64 if n else v


def _keccak_f(a):
    for rc in _KRC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _KROT[x][y])
        a = [[b[x][y] ^ (~b[(x + 1) % 5][y] & _M64 & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data) + b"\x01"
    msg += bytes(-len(msg) % rate)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        a = _keccak_f(a)
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


def expand_message_xmd(dst: bytes, msg: bytes, out_len: int) -> bytes:
    """expandMsgXmdKeccak256 (point.go:289-340): RFC 9380 5.3.1 with H = legacy Keccak-256 (block 136, digest 32)."""
    assert len(dst) <= 255
    dst_prime = dst + bytes([len(dst)])
    b0 = keccak256(bytes(136) + msg + out_len.to_bytes(2, "big") + b"\x00" + dst_prime)
    bi = keccak256(b0 + b"\x01" + dst_prime)
    ell = (out_len + 31) // 32
    out = b""
    for i in range(1, ell):
        out += bi
        bi = keccak256(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([1 + i]) + dst_prime)
    return (out + bi)[:out_len]


def hash_to_field(dst: bytes, msg: bytes):
    """hashToField (point.go:223-235): two elements, 48 bytes each, reduced mod p."""
    t = expand_message_xmd(dst, msg, 96)
    return int.from_bytes(t[:48], "big") % P, int.from_bytes(t[48:], "big") % P


# constants.go:72-84 de-Montgomerised; RFC 9380 6.6.1 with Z = 1 on y^2 = x^3 + 3
SVDW_C1 = 4                                   # g(Z)
SVDW_C2 = (-pow(2, -1, P)) % P                # -Z / 2
SVDW_C3 = 8815841940592487685674414971303048083897117035520822607866   # sqrt(-g(Z) (3 Z^2 + 4 A)), the root with sgn0 = 0
SVDW_C4 = (-16 * pow(3, -1, P)) % P           # 4 (-g(Z)) / (3 Z^2 + 4 A)
assert SVDW_C3 * SVDW_C3 % P == -12 % P and SVDW_C3 % 2 == 0


def _g(x): return (x * x * x + 3) % P


def map_to_point(u: int):
    """mapToPoint (point.go:239-285): inv0 by Fermat (0 -> 0); x1, x2, x3 tried in this order with legendre == 1
    (strictly: a zero g(x) is not accepted, point.go:270-279); sign of y made equal to sgn0(u)."""
    u %= P
    tv1 = u * u % P * SVDW_C1 % P
    tv2 = (1 + tv1) % P
    tv1 = (1 - tv1) % P
    tv3 = pow(tv1 * tv2 % P, P - 2, P)
    tv5 = u * tv1 % P * tv3 % P * SVDW_C3 % P
    x1 = (SVDW_C2 - tv5) % P
    x2 = (SVDW_C2 + tv5) % P
    tv8 = tv2 * tv2 % P * tv3 % P
    x3 = (1 + SVDW_C4 * (tv8 * tv8 % P)) % P
    leg = lambda a: pow(a, (P - 1) // 2, P) == 1
    x = x1 if leg(_g(x1)) else (x2 if leg(_g(x2)) else x3)
    y = pow(_g(x), (P + 1) // 4, P)  # gfP.Sqrt (gfp.go:87-91): no check that a root exists
    if (u & 1) != (y & 1):
        y = -y % P
    return (x, y)


DEFAULT_DST_G1 = b"BN254G1_XMD:KECCAK-256_SVDW_RO_"  # suite.go:42-44


def hash_to_g1(m: bytes, dst: bytes = DEFAULT_DST_G1):
    """pointG1.Hash -> hashToPoint (point.go:211-221): map both field elements, add; no cofactor clearing."""
    e0, e1 = hash_to_field(dst, m)
    return g1_add(map_to_point(e0), map_to_point(e1))


def g1_mul_bytes(scalar_be: bytes, pt: bytes) -> bytes:
    return g1_marshal(g1_mul(int.from_bytes(scalar_be, "big"), g1_unmarshal(pt)))


def g2_mul_bytes(scalar_be: bytes, pt: bytes) -> bytes:
    return g2_marshal(g2_mul(int.from_bytes(scalar_be, "big"), g2_unmarshal(pt)))


# ------------------------------------------------------------- pairing (optate.go restated)
SIXU_PLUS_2_NAF = [  # optate.go:117-120 (signed digits, least significant first)
    0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0,
    0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0, 1, 1,
    1, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1,
    1, 0, 0, -1, 0, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, 1, 1,
]
assert sum(d << i for i, d in enumerate(SIXU_PLUS_2_NAF)) == 6 * U + 2


def _line_add(r, p, q, r2):
    """lineFunctionAdd (optate.go:5-52, identical to bn256's).  r = (x, y, z, t) Jacobian twist point with t = z^2,
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
    """miller (optate.go:124-211); q affine twist point, p affine curve point (both finite)."""
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
    """finalExponentiation (optate.go:213-262), same chain."""
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
    """Suite.Pair -> optimalAte (suite.go:128-130, point.go:781-786, optate.go:264-272): p in G1, q in G2."""
    if p is None or q is None:
        return list(F12_ONE)
    return final_exponentiation(miller(q, p))


def pair_bytes(g1: bytes, g2: bytes) -> bytes:
    return gt_marshal(pair(g1_unmarshal(g1), g2_unmarshal(g2)))


def validate_pairing(p1, p2, inv1, inv2) -> bool:
    """Suite.ValidatePairing (suite.go:134-140): two full pairings + Equal."""
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
