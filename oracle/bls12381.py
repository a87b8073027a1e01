"""CPU ORACLE (test infrastructure, not product code) -- BLS12-381 G1/G2/GT.

The reference's BLS12-381 arithmetic is NOT under /root/reference: the three
suites in pairing/bls12381/{kilic,circl,gnark} are adapters over un-vendored
modules pinned in go.mod:6-8 (github.com/kilic/bls12-381 v0.1.0,
github.com/cloudflare/circl v1.6.3, github.com/consensys/gnark-crypto v0.19.2).
This oracle therefore restates the *published* algorithms with Python integers:
the curve (E: y^2 = x^3 + 4 over Fp, E': y^2 = x^3 + 4(1+i) over Fp2, SURVEY.md
Appendix A), the ZCash compressed encoding that all three adapters emit from
MarshalBinary (kilic/g1.go:119-131, g2.go), and the optimal ate pairing
e(P, Q) = conj(f_{|x|,Q}(P))^((p^12-1)/r) computed the slow textbook way
(affine Miller loop on E(Fp12), final exponent applied as one big power).

Pinned against (tests/test_oracle_bls12381.py):
  * the 16 + 18 ZCash deserialisation fixtures the reference's own test replays
    (pairing/bls12381/deserialization_tests, bls12381_test.go:74-186), committed
    as tests/golden/bls12381_zcash.json;
  * the algebraic identities the reference tests (bilinearity
    bls12381_test.go:448-474, e(aP,bQ) relations :580-630).
PARITY UNPINNED for GT bytes: no reference test fixes the 576-byte GT encoding
or the exact final exponent; the layout here is the one
kilic is believed to use (c1 before c0 at every tower level, big-endian) and the
exponent is the canonical (p^12-1)/r.  Scalar-mul result bytes are pinned only
through the canonical encoding + group law (no fixed KAT in the reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

# ---------------------------------------------------------------- parameters
X_ABS = 0xD201000000010000  # the BLS parameter is x = -X_ABS
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # kilic/scalar.go:11-12
H1 = 0x396C8C005555E1568C00AAAB0000AAAB
assert R == X_ABS**4 - X_ABS**2 + 1
assert P == (X_ABS + 1) ** 2 * R // 3 - X_ABS  # (x-1)^2 r / 3 + x with x = -X_ABS
assert H1 == (X_ABS + 1) ** 2 // 3

G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
G2_GEN = (
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
)

# ------------------------------------------------------------------- Fp2
F2_ZERO, F2_ONE = (0, 0), (1, 0)
XI = (1, 1)  # Fp6 = Fp2[v]/(v^3 - XI), Fp12 = Fp6[w]/(w^2 - v)  => w^6 = XI


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return (-a[0] % P, -a[1] % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return ((a[0] + a[1]) * (a[0] - a[1]) % P, 2 * a[0] * a[1] % P)


def f2_conj(a):
    return (a[0], -a[1] % P)


def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * n % P, -a[1] * n % P)


def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


def fp_sqrt(a):
    """p = 3 mod 4."""
    a %= P
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a else None


def f2_sqrt(a):
    """Any square root of a in Fp2, or None."""
    a0, a1 = a
    if a1 == 0:
        s = fp_sqrt(a0)
        if s is not None:
            return (s, 0)
        s = fp_sqrt(-a0)  # sqrt(a0) = i * sqrt(-a0)
        return (0, s)
    n = fp_sqrt(a0 * a0 + a1 * a1)
    if n is None:
        return None
    inv2 = pow(2, -1, P)
    for s in (n, -n):
        x0 = fp_sqrt((a0 + s) * inv2)
        if x0 is not None and x0 != 0:
            x1 = a1 * pow(2 * x0, -1, P) % P
            if f2_sqr((x0, x1)) == (a0 % P, a1 % P):
                return (x0, x1)
    return None


# ------------------------------------------------------------------ Fp12
# An Fp12 element is held in the w-basis: [a_0..a_5], a_k in Fp2, value sum a_k w^k, w^6 = XI.
# Tower view (c0 + c1 w, c_j = b_0 + b_1 v + b_2 v^2, v = w^2): a_{2m} = c0.b_m, a_{2m+1} = c1.b_m.
F12_ONE = [F2_ONE] + [F2_ZERO] * 5


def f12_mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        if a[i] == F2_ZERO:
            continue
        for j in range(6):
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    return [f2_add(t[k], f2_mul(t[k + 6], XI)) if k < 5 else t[k] for k in range(6)]


def f12_sqr(a):
    return f12_mul(a, a)


def f12_conj(a):
    """a^(p^6): w -> -w."""
    return [a[k] if k % 2 == 0 else f2_neg(a[k]) for k in range(6)]


# Frobenius: (sum a_k w^k)^p = sum conj(a_k) * GAMMA[k] * w^k, GAMMA[k] = XI^(k (p-1)/6)
GAMMA = [f2_pow(XI, k * (P - 1) // 6) for k in range(6)]


def f12_frob(a, n=1):
    for _ in range(n):
        a = [f2_mul(f2_conj(a[k]), GAMMA[k]) for k in range(6)]
    return a


def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_sqr(r)
        if bit == "1":
            r = f12_mul(r, a)
    return r


def f12_inv(a):
    """Via the norm to Fp6 then Fp2: a^-1 = conj-product / norm.  Uses a^(p^6) etc. generically:
    N = a * a^(p^2) * a^(p^4) * ... would be long; instead solve by the tower."""
    # tower split: a = c0 + c1 w, c_j in Fp6 (coefficients of v)
    c0 = [a[0], a[2], a[4]]
    c1 = [a[1], a[3], a[5]]
    # (c0 + c1 w)^-1 = (c0 - c1 w) / (c0^2 - c1^2 v)
    t = f6_sub(f6_mul(c0, c0), f6_mul_v(f6_mul(c1, c1)))
    ti = f6_inv(t)
    r0 = f6_mul(c0, ti)
    r1 = f6_neg(f6_mul(c1, ti))
    return [r0[0], r1[0], r0[1], r1[1], r0[2], r1[2]]


def f6_mul(a, b):
    t = [F2_ZERO] * 5
    for i in range(3):
        for j in range(3):
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    return [f2_add(t[0], f2_mul(t[3], XI)), f2_add(t[1], f2_mul(t[4], XI)), t[2]]


def f6_sub(a, b):
    return [f2_sub(x, y) for x, y in zip(a, b)]


def f6_neg(a):
    return [f2_neg(x) for x in a]


def f6_mul_v(a):
    return [f2_mul(a[2], XI), a[0], a[1]]


def f6_inv(a):
    a0, a1, a2 = a
    t0 = f2_sub(f2_sqr(a0), f2_mul(XI, f2_mul(a1, a2)))
    t1 = f2_sub(f2_mul(XI, f2_sqr(a2)), f2_mul(a0, a1))
    t2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    d = f2_add(f2_mul(a0, t0), f2_mul(XI, f2_add(f2_mul(a2, t1), f2_mul(a1, t2))))
    di = f2_inv(d)
    return [f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di)]


# ------------------------------------------------- curves (affine, None = infinity)
class _Fp:
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % P)
    sub = staticmethod(lambda a, b: (a - b) % P)
    mul = staticmethod(lambda a, b: a * b % P)
    neg = staticmethod(lambda a: -a % P)
    inv = staticmethod(lambda a: pow(a, -1, P))
    b = 4


class _Fp2:
    zero, one = F2_ZERO, F2_ONE
    add, sub, mul, neg, inv = map(staticmethod, (f2_add, f2_sub, f2_mul, f2_neg, f2_inv))
    b = (4, 4)  # 4 * XI


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


def _ec_neg(F, p):
    return None if p is None else (p[0], F.neg(p[1]))


def _ec_mul(F, k, p):
    if k < 0:
        return _ec_mul(F, -k, _ec_neg(F, p))
    r = None
    for bit in bin(k)[2:] if k else "":
        r = _ec_add(F, r, r)
        if bit == "1":
            r = _ec_add(F, r, p)
    return r


def _on_curve(F, p):
    if p is None:
        return True
    x, y = p
    return F.mul(y, y) == F.add(F.mul(F.mul(x, x), x), F.b)


def g1_add(p, q): return _ec_add(_Fp, p, q)
def g1_neg(p): return _ec_neg(_Fp, p)
def g1_mul(k, p): return _ec_mul(_Fp, k, p)
def g1_on_curve(p): return _on_curve(_Fp, p)
def g1_in_subgroup(p): return g1_mul(R, p) is None
def g2_add(p, q): return _ec_add(_Fp2, p, q)
def g2_neg(p): return _ec_neg(_Fp2, p)
def g2_mul(k, p): return _ec_mul(_Fp2, k, p)
def g2_on_curve(p): return _on_curve(_Fp2, p)
def g2_in_subgroup(p): return g2_mul(R, p) is None


# ------------------------------------------- ZCash compressed encoding (Appendix A)
_HALF = (P - 1) // 2


def _fp_larger(y):
    return y > _HALF


def _fp2_larger(y):
    return y[1] > _HALF if y[1] != 0 else y[0] > _HALF


def g1_compress(p) -> bytes:
    if p is None:
        return bytes([0xC0]) + bytes(47)
    b = bytearray(p[0].to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if _fp_larger(p[1]) else 0)
    return bytes(b)


def g2_compress(p) -> bytes:
    if p is None:
        return bytes([0xC0]) + bytes(95)
    (x0, x1), y = p
    b = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if _fp2_larger(y) else 0)
    return bytes(b)


class DecodeError(ValueError):
    pass


def _flags(buf, n):
    if len(buf) != n:
        raise DecodeError("length")
    c, i, s = buf[0] >> 7, (buf[0] >> 6) & 1, (buf[0] >> 5) & 1
    if not c:
        raise DecodeError("compression flag not set")
    body = bytes([buf[0] & 0x1F]) + bytes(buf[1:])
    if i:
        if s or any(body):
            raise DecodeError("bad infinity encoding")
    return i, s, body


def g1_decompress(buf: bytes, subgroup_check: bool = True):
    """kilic FromCompressed + subgroup check (kilic/g1.go:127-131): 48 bytes -> affine / None."""
    inf, s, body = _flags(buf, 48)
    if inf:
        return None
    x = int.from_bytes(body, "big")
    if x >= P:
        raise DecodeError("x >= p")
    y = fp_sqrt(x * x * x + 4)
    if y is None:
        raise DecodeError("not on curve")
    if _fp_larger(y) != bool(s):
        y = -y % P
    pt = (x, y)
    if subgroup_check and not g1_in_subgroup(pt):
        raise DecodeError("not in G1")
    return pt


def g2_decompress(buf: bytes, subgroup_check: bool = True):
    inf, s, body = _flags(buf, 96)
    if inf:
        return None
    x1, x0 = int.from_bytes(body[:48], "big"), int.from_bytes(body[48:], "big")
    if x0 >= P or x1 >= P:
        raise DecodeError("x >= p")
    x = (x0, x1)
    y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), _Fp2.b))
    if y is None:
        raise DecodeError("not on curve")
    if _fp2_larger(y) != bool(s):
        y = f2_neg(y)
    pt = (x, y)
    if subgroup_check and not g2_in_subgroup(pt):
        raise DecodeError("not in G2")
    return pt


def g1_serialize_unc(p) -> bytes:
    """ZCash uncompressed G1: x || y big-endian, 96 bytes; infinity = 0x40 then zeros."""
    if p is None:
        return bytes([0x40]) + bytes(95)
    return p[0].to_bytes(48, "big") + p[1].to_bytes(48, "big")


def g2_serialize_unc(p) -> bytes:
    """ZCash uncompressed G2: x.c1 || x.c0 || y.c1 || y.c0, 192 bytes."""
    if p is None:
        return bytes([0x40]) + bytes(191)
    (x0, x1), (y0, y1) = p
    return b"".join(v.to_bytes(48, "big") for v in (x1, x0, y1, y0))


def _unc_flags(buf, n):
    if len(buf) != n:
        raise DecodeError("length")
    c, i, s = buf[0] >> 7, (buf[0] >> 6) & 1, (buf[0] >> 5) & 1
    if c or s:
        raise DecodeError("compression / sort flag set on an uncompressed encoding")
    body = bytes([buf[0] & 0x1F]) + bytes(buf[1:])
    if i and any(body):
        raise DecodeError("bad infinity encoding")
    return i, body


def g1_deserialize_unc(buf: bytes, validate: bool = True):
    inf, body = _unc_flags(buf, 96)
    if inf:
        return None
    x, y = int.from_bytes(body[:48], "big"), int.from_bytes(body[48:], "big")
    if x >= P or y >= P:
        raise DecodeError("coordinate >= p")
    pt = (x, y)
    if validate:
        if not g1_on_curve(pt):
            raise DecodeError("not on curve")
        if not g1_in_subgroup(pt):
            raise DecodeError("not in G1")
    return pt


def g2_deserialize_unc(buf: bytes, validate: bool = True):
    inf, body = _unc_flags(buf, 192)
    if inf:
        return None
    x1, x0, y1, y0 = (int.from_bytes(body[48 * j:48 * j + 48], "big") for j in range(4))
    if max(x0, x1, y0, y1) >= P:
        raise DecodeError("coordinate >= p")
    pt = ((x0, x1), (y0, y1))
    if validate:
        if not g2_on_curve(pt):
            raise DecodeError("not on curve")
        if not g2_in_subgroup(pt):
            raise DecodeError("not in G2")
    return pt


def scalar_from_be(b: bytes) -> int:
    """mod.Int wire format: 32 bytes big-endian (group/mod/int.go:75,334-350)."""
    return int.from_bytes(b, "big")


def g1_mul_bytes(scalar_be: bytes, pt: bytes) -> bytes:
    """G1Elt.UnmarshalBinary + Mul + MarshalBinary (kilic/g1.go:110-131)."""
    return g1_compress(g1_mul(scalar_from_be(scalar_be), g1_decompress(pt)))


def g2_mul_bytes(scalar_be: bytes, pt: bytes) -> bytes:
    return g2_compress(g2_mul(scalar_from_be(scalar_be), g2_decompress(pt)))


def g1_msm_bytes(scalars, pts) -> bytes:
    acc = None
    for s, p in zip(scalars, pts):
        acc = g1_add(acc, g1_mul(scalar_from_be(s), g1_decompress(p)))
    return g1_compress(acc)


def g2_msm_bytes(scalars, pts) -> bytes:
    acc = None
    for s, p in zip(scalars, pts):
        acc = g2_add(acc, g2_mul(scalar_from_be(s), g2_decompress(p)))
    return g2_compress(acc)


# ----------------------------------------------------------------- pairing
def _embed_fp(a):
    return [(a % P, 0)] + [F2_ZERO] * 5


def _untwist(q):
    """E'(Fp2) -> E(Fp12): (x, y) -> (x / w^2, y / w^3) = (x w^4 / XI, y w^3 / XI)."""
    xi_inv = f2_inv(XI)
    x = [F2_ZERO] * 6
    y = [F2_ZERO] * 6
    x[4] = f2_mul(q[0], xi_inv)
    y[3] = f2_mul(q[1], xi_inv)
    return x, y


def _f12_add(a, b): return [f2_add(x, y) for x, y in zip(a, b)]
def _f12_sub(a, b): return [f2_sub(x, y) for x, y in zip(a, b)]


def miller_loop(p, q):
    """f_{|x|,Q}(P) on E(Fp12), affine, textbook (lines through T,T and T,Q evaluated at P;
    vertical lines omitted -- they lie in a proper subfield and die in the final exponentiation)."""
    if p is None or q is None:
        return list(F12_ONE)
    px, py = _embed_fp(p[0]), _embed_fp(p[1])
    qx, qy = _untwist(q)
    tx, ty = qx, qy
    f = list(F12_ONE)
    three = _embed_fp(3)
    two = _embed_fp(2)

    def line(lam, x0, y0):  # l(P) = (py - y0) - lam (px - x0)
        return _f12_sub(_f12_sub(py, y0), f12_mul(lam, _f12_sub(px, x0)))

    for bit in bin(X_ABS)[3:]:
        lam = f12_mul(f12_mul(three, f12_sqr(tx)), f12_inv(f12_mul(two, ty)))
        f = f12_mul(f12_sqr(f), line(lam, tx, ty))
        nx = _f12_sub(_f12_sub(f12_sqr(lam), tx), tx)
        ty = _f12_sub(f12_mul(lam, _f12_sub(tx, nx)), ty)
        tx = nx
        if bit == "1":
            lam = f12_mul(_f12_sub(qy, ty), f12_inv(_f12_sub(qx, tx)))
            f = f12_mul(f, line(lam, tx, ty))
            nx = _f12_sub(_f12_sub(f12_sqr(lam), tx), qx)
            ty = _f12_sub(f12_mul(lam, _f12_sub(tx, nx)), ty)
            tx = nx
    return f12_conj(f)  # x < 0


HARD_EXP = (P**4 - P**2 + 1) // R


def final_exp(f):
    """f^(3 (p^12-1)/r): easy part by conjugate / inverse / Frobenius, hard part as one big power.

    The factor 3: the reference's default backend (github.com/kilic/bls12-381 v0.1.0, go.mod:8, not vendored) runs the
    five-exponentiation chain of final_exp_kilic_chain() below, which is NOT the canonical exponent (p^4-p^2+1)/r but
    three times it -- test_oracle_bls12381.py checks that the chain, restated from the published code, equals exactly
    this cube (gnark-crypto documents the same cofactor 3 for its FinalExponentiation).  ValidatePairing is unaffected
    (gcd(3, r) = 1); Suite.Pair's GT value is the cube of the canonical reduced pairing.  Pinned by the reference's IBE
    vector (encrypt/ibe/ibe_test.go:202-245): the cube decrypts it, the canonical exponent does not
    (tests/test_oracle_bls12381.py::test_ibe_vector_pins_gt_bytes); round 1 of this repo used the canonical exponent."""
    f = f12_mul(f12_conj(f), f12_inv(f))  # ^(p^6 - 1)
    f = f12_mul(f12_frob(f, 2), f)  # ^(p^2 + 1)
    return f12_pow(f, 3 * HARD_EXP)


def final_exp_kilic_chain(f):
    """kilic/bls12-381 pairing.go finalExp as published (exp(a) = conj(cyclotomicExp(a, |x|)); frobeniusMap(., 6) is the
    conjugation), restated step by step.  Equals final_exp(f); kept separate as the evidence for its exponent."""
    def exp(a):
        return f12_conj(f12_pow(a, X_ABS))

    t = [None] * 7
    t[0] = f12_conj(f)
    t[1] = f12_inv(f)
    t[2] = f12_mul(t[0], t[1])
    t[1] = t[2]
    t[2] = f12_mul(f12_frob(t[2], 2), t[1])
    t[1] = f12_conj(f12_sqr(t[2]))
    t[3] = exp(t[2])
    t[4] = f12_sqr(t[3])
    t[5] = f12_mul(t[1], t[3])
    t[1] = exp(t[5])
    t[0] = exp(t[1])
    t[6] = exp(t[0])
    t[6] = f12_mul(t[6], t[4])
    t[4] = exp(t[6])
    t[5] = f12_conj(t[5])
    t[4] = f12_mul(f12_mul(t[4], t[5]), t[2])
    t[5] = f12_conj(t[2])
    t[1] = f12_frob(f12_mul(t[1], t[2]), 3)
    t[6] = f12_frob(f12_mul(t[6], t[5]), 1)
    t[3] = f12_frob(f12_mul(t[3], t[0]), 2)
    return f12_mul(f12_mul(f12_mul(t[3], t[1]), t[6]), t[4])


def pair(p, q):
    """Suite.Pair (kilic/suite.go:70-75): e(P, Q) in GT (w-basis list of 6 Fp2)."""
    return final_exp(miller_loop(p, q))


def pair_check(p1, q1, p2, q2) -> bool:
    """ValidatePairing(p1, p2, inv1, inv2) == (e(p1, p2) == e(inv1, inv2)) (pairing/pairing.go:13-15,
    kilic/suite.go:57-68).  Arguments here: p* in G1, q* in G2."""
    f = f12_mul(miller_loop(p1, q1), miller_loop(g1_neg(p2), q2))
    return final_exp(f) == F12_ONE


def gt_to_bytes(a) -> bytes:
    """576 bytes, big-endian Fp, reverse tower order (Fp12.c1 then c0; within Fp6 c2,c1,c0; within
    Fp2 c1,c0) -- kilic's layout, PINNED together with the exponent of final_exp() by the
    reference's IBE interop vector (encrypt/ibe/ibe_test.go:202-245, tests/golden/bls12381_ibe.json: decryption hashes
    these 576 bytes; tests/test_oracle_bls12381.py::test_ibe_vector_pins_gt_bytes)."""
    out = b""
    for half in (1, 0):  # c1 (odd w-powers) first
        for m in (2, 1, 0):
            c = a[2 * m + half]
            out += c[1].to_bytes(48, "big") + c[0].to_bytes(48, "big")
    return out


def pair_bytes(g1: bytes, g2: bytes) -> bytes:
    return gt_to_bytes(pair(g1_decompress(g1), g2_decompress(g2)))


def gt_from_bytes(buf: bytes):
    """kilic GT.FromBytes (kilic/gt.go:100-104): 576 bytes, coefficients < p, order-r subgroup."""
    if len(buf) != 576:
        raise DecodeError("length")
    a = [None] * 6
    k = 0
    for half in (1, 0):
        for m in (2, 1, 0):
            c1 = int.from_bytes(buf[96 * k:96 * k + 48], "big")
            c0 = int.from_bytes(buf[96 * k + 48:96 * k + 96], "big")
            if c0 >= P or c1 >= P:
                raise DecodeError("coefficient >= p")
            a[2 * m + half] = (c0, c1)
            k += 1
    if f12_pow(a, R) != F12_ONE:
        raise DecodeError("not in GT")
    return a


def gt_mul_bytes(scalar_be: bytes, gt: bytes) -> bytes:
    """GTElt.Mul (kilic/gt.go:79-84): gt ^ k."""
    return gt_to_bytes(f12_pow(gt_from_bytes(gt), scalar_from_be(scalar_be)))


# ------------------------------------------------------- hash to curve (RFC 9380 section 8.8)
# kilic/g1.go:161-170 (G1Elt.Hash -> HashToCurve with the suite DST) and g2.go likewise.  The isogeny
# constants are derived by tools/derive_bls12381_isogenies.py (the backends are not vendored); the choice among
# the automorphisms of the j = 0 curve is pinned by the drand fixtures (tests/test_oracle_bls12381_h2c.py).
def expand_message_xmd(msg: bytes, dst: bytes, n: int) -> bytes:
    import hashlib
    import struct

    ell = (n + 31) // 32
    if ell > 255 or len(dst) > 255:
        raise ValueError("expand_message_xmd: bad lengths")
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(bytes(64) + msg + struct.pack(">H", n) + b"\x00" + dst_prime).digest()
    b = [hashlib.sha256(b0 + b"\x01" + dst_prime).digest()]
    for i in range(2, ell + 1):
        b.append(hashlib.sha256(bytes(x ^ y for x, y in zip(b0, b[-1])) + bytes([i]) + dst_prime).digest())
    return b"".join(b)[:n]


def _poly(c, x, F):
    acc = F.zero
    for a in reversed(c):
        acc = F.add(F.mul(acc, x), a)
    return acc


def _sgn0_fp(x):
    return x & 1


def _sgn0_fp2(x):
    return (x[0] & 1) | ((x[0] == 0) & (x[1] & 1))


def _sswu(F, A, B, Z, u, sqrt, sgn0):
    """map_to_curve_simple_swu (RFC 9380 section 6.6.2) on y^2 = x^3 + A x + B."""
    u2 = F.mul(u, u)
    tv1 = F.add(F.mul(F.mul(Z, Z), F.mul(u2, u2)), F.mul(Z, u2))
    if tv1 == F.zero:
        x1 = F.mul(B, F.inv(F.mul(Z, A)))
    else:
        x1 = F.mul(F.mul(F.neg(B), F.inv(A)), F.add(F.one, F.inv(tv1)))
    g = lambda x: F.add(F.add(F.mul(F.mul(x, x), x), F.mul(A, x)), B)
    y = sqrt(g(x1))
    if y is None:
        x1 = F.mul(F.mul(Z, u2), x1)
        y = sqrt(g(x1))
    if sgn0(u) != sgn0(y):
        y = F.neg(y)
    return (x1, y)


def _iso(F, consts, pt):
    xn, xd, yn, yd = consts
    x, y = pt
    return (F.mul(_poly(xn, x, F), F.inv(_poly(xd, x, F))), F.mul(y, F.mul(_poly(yn, x, F), F.inv(_poly(yd, x, F)))))


H_EFF_G1 = 0xD201000000010001  # 1 - x
_XX = -X_ABS
H2 = (_XX**8 - 4 * _XX**7 + 5 * _XX**6 - 4 * _XX**4 + 6 * _XX**3 - 4 * _XX**2 - 4 * _XX + 13) // 9
H_EFF_G2 = H2 * (3 * _XX * _XX - 3)


def hash_to_g1(msg: bytes, dst: bytes):
    from . import bls12381_h2c_consts as K

    ub = expand_message_xmd(msg, dst, 128)
    pts = []
    for k in range(2):
        u = int.from_bytes(ub[64 * k:64 * k + 64], "big") % P
        q = _sswu(_Fp, K.G1_A, K.G1_B, K.G1_Z, u, fp_sqrt, _sgn0_fp)
        pts.append(_iso(_Fp, (K.G1_XNUM, K.G1_XDEN, K.G1_YNUM, K.G1_YDEN), q))
    return g1_mul(H_EFF_G1, g1_add(pts[0], pts[1]))


def hash_to_g2(msg: bytes, dst: bytes):
    from . import bls12381_h2c_consts as K

    ub = expand_message_xmd(msg, dst, 256)
    e = [int.from_bytes(ub[64 * k:64 * k + 64], "big") % P for k in range(4)]
    pts = []
    for u in ((e[0], e[1]), (e[2], e[3])):
        q = _sswu(_Fp2, K.G2_A, K.G2_B, K.G2_Z, u, f2_sqrt, _sgn0_fp2)
        pts.append(_iso(_Fp2, (K.G2_XNUM, K.G2_XDEN, K.G2_YNUM, K.G2_YDEN), q))
    return g2_mul(H_EFF_G2, g2_add(pts[0], pts[1]))
