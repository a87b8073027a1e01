"""Device constants (kyber_amd/csrc/*.inc) re-derived independently."""
import os
import re

import pytest

from oracle import ed25519 as O

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber_amd", "csrc")


def _limbs_to_int(ls):
    x, off = 0, 0
    for i, l in enumerate(ls):
        x += l << off
        off += 25 if i & 1 else 26
    return x


def test_ed25519_consts():
    src = open(os.path.join(CSRC, "ed25519_consts.inc")).read()
    vals = {}
    for name, body in re.findall(r"fe fe_(\w+)\(\) \{ return fe\{\{([^}]*)\}\}", src):
        ls = [int(x) for x in body.split(",")]
        assert len(ls) == 10
        # limb bounds the field code relies on
        for i, l in enumerate(ls):
            assert 0 <= l < (1 << (25 if i & 1 else 26))
        vals[name] = _limbs_to_int(ls)
    assert vals["d"] == O.D
    assert vals["d2"] == 2 * O.D % O.P
    assert vals["sqrtm1"] == O.SQRT_M1
    assert (vals["bx"], vals["by"]) == O.B
    assert vals["bt"] == O.B[0] * O.B[1] % O.P
    # the two constants of the straight-line Elligator 2 (RFC 9380 appendix G.2): c1 = sqrt(-486664) with sgn0 = 0,
    # c2 = 2^((p + 3) / 8)
    assert vals["elligator_c1"] ** 2 % O.P == (-486664) % O.P and vals["elligator_c1"] % 2 == 0
    assert vals["elligator_c2"] == pow(2, (O.P + 3) // 8, O.P)


# ------------------------------------------------------------------ pairing-curve parameter headers
def _parse_params(fname):
    src = open(os.path.join(CSRC, fname)).read()
    out = {}
    for name, dims, body in re.findall(r"static constexpr uint32_t (\w+)((?:\[\w+\])+) = (\{.*?\});", src, re.S):
        nums = [int(x.rstrip("u"), 16) for x in re.findall(r"0x[0-9a-f]+u", body)]
        out.setdefault(name, []).append(nums)
    scal = dict(re.findall(r"static constexpr (?:uint32_t|int|uint64_t) (\w+) = (0x[0-9a-f]+|\d+)u?(?:ll)?;", src))
    nw = re.search(r"N = (\d+), W = (\d+), NWORDS = (\d+)", src)
    return out, scal, tuple(int(x) for x in nw.groups())


def _val(limbs, w):
    return sum(l << (w * i) for i, l in enumerate(limbs))


def _check_field(P, fname):
    arr, scal, (n, w, nwords) = _parse_params(fname)
    R = 1 << (n * w)
    assert _val(arr["P"][0], w) == P
    assert all(l < (1 << w) for l in arr["P"][0])
    # elements are stored as NWORDS saturated 32-bit words (the multiplier alone works on the W-bit limbs of P)
    assert _val(arr["PW"][0], 32) == P and len(arr["PW"][0]) == nwords
    assert _val(arr["ONE"][0], 32) == R % P
    assert _val(arr["R2"][0], 32) == R * R % P
    ninv = int(re.search(r"NINV = (0x[0-9a-f]+)u", open(os.path.join(CSRC, fname)).read()).group(1), 16)
    assert (ninv * P + 1) % (1 << w) == 0
    assert _val(arr["PM2"][0], 32) == P - 2
    assert _val(arr["HALF"][0], 32) == (P - 1) // 2
    assert _val(arr["SQRT_EXP"][0], 32) == (P + 1) // 4
    assert _val(arr["PM3D4"][0], 32) == (P - 3) // 4
    assert _val(arr["INV2"][0], 32) * 2 % P == R % P
    return arr, R, 32, nwords


def test_bls12381_params():
    from oracle import bls12381 as B

    arr, R, w, n = _check_field(B.P, "bls12381_params.h")
    ri = pow(R, -1, B.P)
    mont = lambda limbs: _val(limbs, w) * ri % B.P
    assert mont(arr["B1"][0]) == 4
    assert (mont(arr["G1X"][0]), mont(arr["G1Y"][0])) == B.G1_GEN
    # beta: phi(P) = (beta x, y) acts as [-x^2] on G1
    beta = mont(arr["BETA"][0])
    g = B.G1_GEN
    assert (g[0] * beta % B.P, g[1]) == B.g1_mul((-B.X_ABS**2) % B.R, g)


@pytest.mark.parametrize("name", ["bn256", "bn254"])
def test_bn_params(name):
    import importlib

    N = importlib.import_module("oracle." + name)
    arr, R, w, n = _check_field(N.P, name + "_params.h")
    ri = pow(R, -1, N.P)
    mont = lambda limbs: _val(limbs, w) * ri % N.P
    assert mont(arr["B1"][0]) == 3
    assert (mont(arr["G1X"][0]), mont(arr["G1Y"][0])) == N.G1_GEN
    flat = lambda a: [a[i * n:(i + 1) * n] for i in range(len(a) // n)]
    b2 = flat(arr["B2"][0])
    assert (mont(b2[0]), mont(b2[1])) == N.TWIST_B
    gx, gy = flat(arr["G2X"][0]), flat(arr["G2Y"][0])
    assert ((mont(gx[0]), mont(gx[1])), (mont(gy[0]), mont(gy[1]))) == N.G2_GEN
    assert _val(arr["ORDER"][0], 32) == N.ORDER
    # GLV: phi(x, y) = (beta x, y) is [lambda] on G1 and the split reproduces k for edge scalars
    beta = mont(arr["BETA"][0])
    lam = 36 * N.U**3 + 18 * N.U**2 + 6 * N.U + 1
    g = N.G1_GEN
    assert (g[0] * beta % N.P, g[1]) == N.g1_mul(lam, g)
    a1, b1n, a2, b2v = (_val(arr[k][0], 32) for k in ("GLV_A1", "GLV_B1N", "GLV_A2", "GLV_B2"))
    g1, g2 = _val(arr["GLV_G1"][0], 32), _val(arr["GLV_G2"][0], 32)
    for k in (0, 1, N.ORDER - 1, N.ORDER, (1 << 256) - 1, lam, 0xDEADBEEF << 200):
        c1, c2 = (k * g1) >> 256, (k * g2) >> 256
        k1, k2 = k - c1 * a1 - c2 * a2, c1 * b1n - c2 * b2v
        assert (k1 + k2 * lam - k) % N.ORDER == 0 and abs(k1) < 1 << 130 and abs(k2) < 1 << 130
    if name == "bn254":
        assert [mont(arr["SVDW_C%d" % i][0]) for i in (1, 2, 3, 4)] == [N.SVDW_C1, N.SVDW_C2, N.SVDW_C3, N.SVDW_C4]
    if name == "bn256":  # HashG1's root of -3 (constants.go:105): the header, the oracle and the extracted golden value agree
        import json
        import os

        g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bn256.json")))
        assert mont(arr["SVDW_S"][0]) == N.SVDW_S == int(g["svdw_s"], 16)
        assert mont(arr["SVDW_SM1D2"][0]) == N.SVDW_S_MINUS_1_OVER_2 == int(g["svdw_s_minus_1_over_2"], 16)
    assert mont(arr["TWO256"][0]) == (1 << 256) % N.P


def test_bn254_g2_subgroup_criterion_is_exact():
    """kyber_amd/csrc/bn_suite.inc g2_in_subgroup decides [n]Q = infinity by (u+1) + u psi + u psi^2 - 2u psi^3 = 0.
    Exactness on alt_bn128: #E'(Fp2) = n h, h = 2p - n = q1 q2 q3 q4 (distinct primes, none dividing n: the group is
    cyclic); on the q-component psi is a root mu of X^2 - t X + p (mod q), and the polynomial is non-zero at BOTH roots
    for every q -- so the relation holds exactly on G2."""
    from oracle import bn254 as N

    def is_prime(m):
        d, s = m - 1, 0
        while d % 2 == 0:
            d, s = d // 2, s + 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53):
            if m == a:
                return True
            x = pow(a, d, m)
            if x in (1, m - 1):
                continue
            for _ in range(s - 1):
                x = x * x % m
                if x == m - 1:
                    break
            else:
                return False
        return True

    def sqrt_mod(a, q):  # Tonelli-Shanks
        a %= q
        assert pow(a, (q - 1) // 2, q) == 1
        Q, S = q - 1, 0
        while Q % 2 == 0:
            Q, S = Q // 2, S + 1
        z = 2
        while pow(z, (q - 1) // 2, q) != q - 1:
            z += 1
        m, c, t, r = S, pow(z, Q, q), pow(a, Q, q), pow(a, (Q + 1) // 2, q)
        while t != 1:
            i, t2 = 0, t
            while t2 != 1:
                t2, i = t2 * t2 % q, i + 1
            b = pow(c, 1 << (m - i - 1), q)
            m, c = i, b * b % q
            t, r = t * c % q, r * b % q
        return r

    u, p, n = N.U, N.P, N.ORDER
    t = 6 * u * u + 1
    assert p + 1 - t == n
    h = 2 * p - n
    prod = 1
    for q in N.G2_COFACTOR_PRIMES:
        assert is_prime(q) and n % q != 0
        prod *= q
    assert prod == h and len(set(N.G2_COFACTOR_PRIMES)) == 4
    assert ((u + 1) + u * p + u * p * p - 2 * u * p**3) % n == 0          # the relation on G2, where psi = [p]
    for q in N.G2_COFACTOR_PRIMES:
        r = sqrt_mod(t * t - 4 * p, q)
        for mu in ((t + r) * pow(2, -1, q) % q, (t - r) * pow(2, -1, q) % q):
            assert (mu * mu - t * mu + p) % q == 0
            assert ((u + 1) + u * mu + u * mu * mu - 2 * u * mu**3) % q != 0


def test_bn256_twist_cofactor_and_the_criterion_that_would_gate_gls():
    """DESIGN.md section 9 item 4 (not built): bn256's UnmarshalBinary accepts G2 points outside the order-n subgroup, so
    its default G2 ladder is the plain one.  The facts a membership-gated GLS walk would rest on, checked here so that the
    claim in the design notes is not hearsay: the twist's cofactor is 13 * 7369 * (a 239-bit prime), coprime to n (the
    group is cyclic), and bn254's criterion (u+1) + u psi + u psi^2 - 2u psi^3 = 0 is exact on this curve too -- the
    polynomial vanishes at psi = [p] on G2 and is non-zero at both eigenvalues of psi modulo every cofactor prime."""
    from oracle import bn256 as N

    def is_prime(m):
        d, s = m - 1, 0
        while d % 2 == 0:
            d, s = d // 2, s + 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53):
            if m == a:
                return True
            x = pow(a, d, m)
            if x in (1, m - 1):
                continue
            for _ in range(s - 1):
                x = x * x % m
                if x == m - 1:
                    break
            else:
                return False
        return True

    def roots(t, p, q):  # roots of X^2 - t X + p modulo the prime q (the discriminant is a residue for all three)
        disc = (t * t - 4 * p) % q
        assert pow(disc, (q - 1) // 2, q) == 1
        # Tonelli-Shanks
        Q, S = q - 1, 0
        while Q % 2 == 0:
            Q, S = Q // 2, S + 1
        z = 2
        while pow(z, (q - 1) // 2, q) != q - 1:
            z += 1
        m, c, tt, r = S, pow(z, Q, q), pow(disc, Q, q), pow(disc, (Q + 1) // 2, q)
        while tt != 1:
            i, t2 = 0, tt
            while t2 != 1:
                t2, i = t2 * t2 % q, i + 1
            b = pow(c, 1 << (m - i - 1), q)
            m, c = i, b * b % q
            tt, r = tt * c % q, r * b % q
        inv2 = pow(2, -1, q)
        return ((t + r) * inv2 % q, (t - r) * inv2 % q)

    u, p, n = N.U, N.P, N.ORDER
    t = 6 * u * u + 1
    assert p + 1 - t == n
    h = 2 * p - n
    q3 = h // (13 * 7369)
    primes = (13, 7369, q3)
    assert 13 * 7369 * q3 == h and q3.bit_length() == 239 and all(is_prime(q) and n % q != 0 for q in primes)
    assert ((u + 1) + u * p + u * p * p - 2 * u * p**3) % n == 0
    for q in primes:
        for mu in roots(t, p, q):
            assert (mu * mu - t * mu + p) % q == 0
            assert ((u + 1) + u * mu + u * mu * mu - 2 * u * mu**3) % q != 0


def test_bls12381_hash_to_curve_square_root_constants():
    """the constants that let one power serve both branches of an SSWU map (round 5): sqrt(-Z) on G1 (RFC 9380 F.2.1.2),
    sqrt(-N(Z)^3) in Fp on G2 (the norm root of g(x2) from the norm root of g(x1)); Z must be what makes them exist"""
    from oracle import bls12381 as B
    from oracle import bls12381_h2c_consts as HC

    arr, _, _ = _parse_params("bls12381_params.h")
    R = 1 << 390
    src = open(os.path.join(CSRC, "bls12381_h2c_params.h")).read()
    get = lambda name: [int(x.rstrip("u"), 16) for x in re.findall(r"0x[0-9a-f]+u", re.search(name + r"\[12\] = (\{.*?\});", src).group(1))]
    mont = lambda words: _val(words, 32) * pow(R, -1, B.P) % B.P
    c = mont(get("G1_SQRT_NEG_Z"))
    assert c * c % B.P == (-HC.G1_Z) % B.P
    assert pow(HC.G1_Z % B.P, (B.P - 1) // 2, B.P) == B.P - 1          # Z is no square, -Z is one (p = 3 mod 4)
    nz = (HC.G2_Z[0] ** 2 + HC.G2_Z[1] ** 2) % B.P
    assert pow(nz, (B.P - 1) // 2, B.P) == B.P - 1                       # N(Z) is no square in Fp <=> Z is none in Fp2
    c = mont(get("G2_SQRT_NEG_NZ3"))
    assert c * c % B.P == (-pow(nz, 3, B.P)) % B.P
