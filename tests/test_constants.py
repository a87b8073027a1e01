"""Device constants (kyber_amd/csrc/*.inc) re-derived independently."""
import os
import re

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
    # FROB[k-1][j] = xi^(j (p^k - 1) / 6): flattened 3 x 6 x 2 limb arrays
    frob = arr["FROB"][0]
    flat = [frob[i * n:(i + 1) * n] for i in range(len(frob) // n)]
    for k in (1, 2, 3):
        for j in range(6):
            c0, c1 = flat[((k - 1) * 6 + j) * 2], flat[((k - 1) * 6 + j) * 2 + 1]
            assert (mont(c0), mont(c1)) == B.f2_pow(B.XI, j * (B.P**k - 1) // 6)
    # beta: phi(P) = (beta x, y) acts as [-x^2] on G1
    beta = mont(arr["BETA"][0])
    g = B.G1_GEN
    assert (g[0] * beta % B.P, g[1]) == B.g1_mul((-B.X_ABS**2) % B.R, g)
    lam3 = _val(arr["LAMBDA3"][0], 32)
    x = -B.X_ABS
    assert lam3 * 3 == (x - 1) ** 2
    l2 = x * lam3
    l1 = x * l2 - lam3
    l0 = x * l1 + 1
    assert l0 + l1 * B.P + l2 * B.P**2 + lam3 * B.P**3 == (B.P**4 - B.P**2 + 1) // B.R


def test_bn256_params():
    from oracle import bn256 as N

    arr, R, w, n = _check_field(N.P, "bn256_params.h")
    ri = pow(R, -1, N.P)
    mont = lambda limbs: _val(limbs, w) * ri % N.P
    assert mont(arr["B1"][0]) == 3
    assert (mont(arr["G1X"][0]), mont(arr["G1Y"][0])) == N.G1_GEN
    flat = lambda a: [a[i * n:(i + 1) * n] for i in range(len(a) // n)]
    b2 = flat(arr["B2"][0])
    assert (mont(b2[0]), mont(b2[1])) == N.TWIST_B
    gx, gy = flat(arr["G2X"][0]), flat(arr["G2Y"][0])
    assert ((mont(gx[0]), mont(gx[1])), (mont(gy[0]), mont(gy[1]))) == N.G2_GEN
    q1x = flat(arr["Q1X"][0])
    assert (mont(q1x[0]), mont(q1x[1])) == N.f2_pow(N.XI, (N.P - 1) // 3)
    assert mont(arr["Q2X"][0]) == N.f2_pow(N.XI, (N.P * N.P - 1) // 3)[0]
