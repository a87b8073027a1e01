#!/usr/bin/env python3
"""Derive the BLS12-381 hash-to-curve isogeny maps (RFC 9380 section 8.8, `BLS12381G1_XMD:SHA-256_SSWU_RO_` /
`BLS12381G2_XMD:SHA-256_SSWU_RO_`) from first principles and write them as constants.

Why: the reference's BLS12-381 hashing lives in un-vendored modules (kilic/g1.go:161-170 -> HashToCurve) and the
~70 rational-map coefficients are not available offline.  They are recomputed here:

  G1: E1': y^2 = x^3 + A1 x + B1 over Fp is 11-isogenous to E: y^2 = x^3 + 4.  E1'(Fp) has a cyclic 11-part of
      order 121, hence ONE rational subgroup of order 11; Velu's formulas on it land on y^2 = x^3 + 4 * 11^6, and
      (x, y) -> (x / 11^2, y / 11^3) finishes the map.
  G2: E2': y^2 = x^3 + 240 i x + 1012 (1 + i) over Fp2 is 3-isogenous to E2: y^2 = x^3 + 4 (1 + i).  The
      3-division polynomial has one root in Fp2 (the kernel points themselves live in Fp4); Velu lands on
      y^2 = x^3 + 4 (1 + i) * 3^6 and (x, y) -> (x / 9, -y / 27) finishes the map.

The remaining freedom (which of the six automorphisms of a j = 0 curve to compose with) is pinned by the
reference's own golden vectors: the drand signatures of pairing/bls12381/kilic/suite_test.go:17-72 verify with
exactly the choice above and with none of the other five (tests/test_oracle_bls12381_h2c.py replays them).

Outputs: oracle/bls12381_h2c_consts.py (integers) and kyber_amd/csrc/bls12381_h2c_params.h (Montgomery limbs).
Run in any container (needs only this repo); deterministic.
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kyber_amd", "csrc"))
from oracle import bls12381 as O  # noqa: E402  (field / curve helpers only)

p, r = O.P, O.R
X = -O.X_ABS


class Fp:
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % p)
    sub = staticmethod(lambda a, b: (a - b) % p)
    mul = staticmethod(lambda a, b: a * b % p)
    muli = staticmethod(lambda a, k: a * k % p)
    inv = staticmethod(lambda a: pow(a, -1, p))
    sqrt = staticmethod(O.fp_sqrt)
    rand = staticmethod(lambda rng: rng.randrange(p))


class Fp2:
    zero, one = (0, 0), (1, 0)
    add, sub, mul, inv = map(staticmethod, (O.f2_add, O.f2_sub, O.f2_mul, O.f2_inv))
    muli = staticmethod(lambda a, k: (a[0] * k % p, a[1] * k % p))
    sqrt = staticmethod(O.f2_sqrt)
    rand = staticmethod(lambda rng: (rng.randrange(p), rng.randrange(p)))


class Poly:
    def __init__(self, F, c):
        self.F, self.c = F, list(c)
        while len(self.c) > 1 and self.c[-1] == F.zero:
            self.c.pop()

    def deg(self):
        return len(self.c) - 1

    def _zip(self, o, f):
        F, n = self.F, max(len(self.c), len(o.c))
        g = lambda c, i: c[i] if i < len(c) else F.zero
        return Poly(F, [f(g(self.c, i), g(o.c, i)) for i in range(n)])

    def __add__(self, o): return self._zip(o, self.F.add)
    def __sub__(self, o): return self._zip(o, self.F.sub)

    def __mul__(self, o):
        F = self.F
        out = [F.zero] * (len(self.c) + len(o.c) - 1)
        for i, a in enumerate(self.c):
            if a != F.zero:
                for j, b in enumerate(o.c):
                    out[i + j] = F.add(out[i + j], F.mul(a, b))
        return Poly(F, out)

    def scale(self, k): return Poly(self.F, [self.F.mul(k, a) for a in self.c])
    def deriv(self): return Poly(self.F, [self.F.muli(self.c[i], i) for i in range(1, len(self.c))] or [self.F.zero])

    def mod(self, m):
        F, a, mc = self.F, list(self.c), m.c
        dm, il = len(mc) - 1, F.inv(mc[-1])
        while len(a) - 1 >= dm:
            if a[-1] != F.zero:
                f, sh = F.mul(a[-1], il), len(a) - 1 - dm
                for i in range(dm + 1):
                    a[sh + i] = F.sub(a[sh + i], F.mul(f, mc[i]))
            a.pop()
        return Poly(F, a or [F.zero])


def ppowmod(a, e, m):
    res = Poly(a.F, [a.F.one])
    while e:
        if e & 1:
            res = (res * a).mod(m)
        a = (a * a).mod(m)
        e >>= 1
    return res


def pgcd(a, b):
    while not (b.deg() == 0 and b.c[0] == b.F.zero):
        a, b = b, a.mod(b)
    return a.scale(a.F.inv(a.c[-1]))


def velu_x_map(F, A, B, kernel_x, rhs):
    """Velu for an odd-degree kernel given by the x-coordinates of one point of each +-pair (their y^2 = rhs(x)
    is all that is needed).  Returns (A'', B'', N, h) with  X = N / h^2,  Y = y (N' h - 2 N h') / h^3."""
    x = Poly(F, [F.zero, F.one])
    h = Poly(F, [F.one])
    terms, v, w = [], F.zero, F.zero
    for xq in kernel_x:
        gx = F.add(F.muli(F.mul(xq, xq), 3), A)
        vq, uq = F.muli(gx, 2), F.muli(rhs(xq), 4)
        v, w = F.add(v, vq), F.add(w, F.add(uq, F.mul(xq, vq)))
        terms.append((xq, vq, uq))
        h = h * Poly(F, [F.sub(F.zero, xq), F.one])
    A2, B2 = F.sub(A, F.muli(v, 5)), F.sub(B, F.muli(w, 7))
    N = x * h * h
    for xq, vq, uq in terms:
        others = Poly(F, [F.one])
        for xr, _, _ in terms:
            if xr != xq:
                others = others * Poly(F, [F.sub(F.zero, xr), F.one])
        N = N + Poly(F, [F.sub(uq, F.mul(vq, xq)), vq]) * others * others
    return A2, B2, N, h


def finish(F, N, h, ux2, uy3):
    """Compose with (x, y) -> (ux2 x, uy3 y) and return the four coefficient lists (low degree first):
    x = xnum / xden, y = y' * ynum / yden."""
    xnum = N.scale(ux2)
    xden = h * h
    ynum = (N.deriv() * h - (N * h.deriv()).scale(F.muli(F.one, 2))).scale(uy3)
    yden = h * h * h
    return xnum.c, xden.c, ynum.c, yden.c


def derive_g1():
    A = 0x144698A3B8E9433D693A02C96D4982B0EA985383EE66A8D8E8981AEFD881AC98936F8DA0E0F97F5CF428082D584C1D
    B = 0x12E2908D11688030018B12E8753EEE3B2016C1F0F24F4070A0B9C14FCEF35EF55A23215A316CEAA5D1CC48E98E172BE0
    rhs = lambda x: (x * x * x + A * x + B) % p

    def add(P1, P2):
        if P1 is None: return P2
        if P2 is None: return P1
        (x1, y1), (x2, y2) = P1, P2
        if x1 == x2:
            if (y1 + y2) % p == 0: return None
            lam = (3 * x1 * x1 + A) * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return (x3, (lam * (x1 - x3) - y1) % p)

    def mul(k, P1):
        acc = None
        for bit in bin(k)[2:]:
            acc = add(acc, acc)
            if bit == "1": acc = add(acc, P1)
        return acc

    n1 = O.H1 * r
    assert n1 % 121 == 0 and n1 % 1331 != 0
    rng = random.Random(1)
    subgroups = set()
    for _ in range(8):
        while True:
            x = rng.randrange(p)
            y = O.fp_sqrt(rhs(x))
            if y is not None: break
        assert mul(n1, (x, y)) is None  # E1' has the order of E(Fp)
        T = mul(n1 // 121, (x, y))
        if T is None: continue
        while mul(11, T) is not None:
            T = mul(11, T)
        subgroups.add(tuple(sorted(mul(j, T)[0] for j in range(1, 6))))
    assert len(subgroups) == 1, "expected a unique rational subgroup of order 11"
    kx = list(subgroups.pop())
    A2, B2, N, h = velu_x_map(Fp, A, B, kx, rhs)
    assert A2 == 0 and B2 == 4 * 11**6
    u = pow(11, -1, p)
    return dict(A=A, B=B, Z=11, maps=finish(Fp, N, h, u * u % p, pow(u, 3, p)))


def derive_g2():
    A, B, Z = (0, 240), (1012, 1012), ((-2) % p, (-1) % p)
    F = Fp2
    rhs = lambda x: F.add(F.add(F.mul(F.mul(x, x), x), F.mul(A, x)), B)
    # 3-division polynomial 3 x^4 + 6 A x^2 + 12 B x - A^2 ; its roots in Fp2
    psi3 = Poly(F, [F.sub(F.zero, F.mul(A, A)), F.muli(B, 12), F.muli(A, 6), F.zero, (3, 0)])
    xp = Poly(F, [F.zero, F.one])
    g = pgcd(psi3, ppowmod(xp, p * p, psi3) - xp)
    assert g.deg() == 1, "expected exactly one rational 3-division abscissa"
    x0 = F.sub(F.zero, g.c[0])
    A2, B2, N, h = velu_x_map(F, A, B, [x0], rhs)
    assert A2 == (0, 0) and B2 == (4 * 3**6, 4 * 3**6)
    i3 = pow(3, -1, p)
    return dict(A=A, B=B, Z=Z, maps=finish(F, N, h, (i3 * i3 % p, 0), ((-pow(i3, 3, p)) % p, 0)))


def main():
    g1, g2 = derive_g1(), derive_g2()
    names = ("XNUM", "XDEN", "YNUM", "YDEN")
    # ---- oracle constants
    L = ['"""generated by tools/derive_bls12381_isogenies.py -- do not edit.', "",
         "BLS12-381 hash-to-curve constants (RFC 9380 section 8.8): SSWU curve parameters and the isogeny maps",
         'x = XNUM(x\') / XDEN(x\'), y = y\' * YNUM(x\') / YDEN(x\'), coefficient lists low degree first."""', ""]
    L.append(f"G1_A = {g1['A']:#x}\nG1_B = {g1['B']:#x}\nG1_Z = {g1['Z']}")
    for nm, c in zip(names, g1["maps"]):
        L.append(f"G1_{nm} = [\n" + "".join(f"    {v:#x},\n" for v in c) + "]")
    L.append(f"G2_A = {g2['A']}\nG2_B = {g2['B']}\nG2_Z = {g2['Z']}")
    for nm, c in zip(names, g2["maps"]):
        L.append(f"G2_{nm} = [\n" + "".join(f"    ({v[0]:#x}, {v[1]:#x}),\n" for v in c) + "]")
    open(os.path.join(ROOT, "oracle", "bls12381_h2c_consts.py"), "w").write("\n".join(L) + "\n")
    # ---- device constants (Montgomery limbs)
    import gen_consts as G

    F = G.MontField(p, 13, 30, 12)
    m1 = lambda v: G._arr(F.mont(v))
    m2 = lambda v: F.mont2(v)
    H = ["// generated by tools/derive_bls12381_isogenies.py -- do not edit", "#pragma once", "#include <stdint.h>",
         "namespace kyb {", "struct Bls12381H2c {"]
    neg_b_over_a = (-g1["B"] * pow(g1["A"], -1, p)) % p
    b_over_za = g1["B"] * pow(g1["Z"] * g1["A"], -1, p) % p
    H += [f"    static constexpr uint32_t TWO384[12] = {m1(pow(2, 384, p))};  // 2^384 mod p (hash_to_field: 512-bit -> Fp)",
          f"    static constexpr uint32_t G1_A[12] = {m1(g1['A'])};",
          f"    static constexpr uint32_t G1_B[12] = {m1(g1['B'])};",
          f"    static constexpr uint32_t G1_Z[12] = {m1(g1['Z'])};",
          f"    static constexpr uint32_t G1_NEG_B_OVER_A[12] = {m1(neg_b_over_a)};",
          f"    static constexpr uint32_t G1_B_OVER_ZA[12] = {m1(b_over_za)};",
          f"    static constexpr uint32_t G1_SQRT_NEG_Z[12] = {m1(pow((-g1['Z']) % p, (p + 1) // 4, p))};  // sqrt(-Z): the constant of sqrt_ratio for p = 3 mod 4 (RFC 9380 F.2.1.2)"]
    for nm, c in zip(names, g1["maps"]):
        H.append(f"    static constexpr int G1_{nm}_LEN = {len(c)};")
        H.append(f"    static constexpr uint32_t G1_{nm}[{len(c)}][12] = {{" + ", ".join(m1(v) for v in c) + "};")
    nba2 = O.f2_mul(O.f2_neg(g2["B"]), O.f2_inv(g2["A"]))
    bza2 = O.f2_mul(g2["B"], O.f2_inv(O.f2_mul(g2["Z"], g2["A"])))
    H += [f"    static constexpr uint32_t G2_A[2][12] = {m2(g2['A'])};",
          f"    static constexpr uint32_t G2_B[2][12] = {m2(g2['B'])};",
          f"    static constexpr uint32_t G2_Z[2][12] = {m2(g2['Z'])};",
          f"    static constexpr uint32_t G2_NEG_B_OVER_A[2][12] = {m2(nba2)};",
          f"    static constexpr uint32_t G2_B_OVER_ZA[2][12] = {m2(bza2)};",
          f"    static constexpr uint32_t G2_SQRT_NEG_NZ3[12] = {m1(pow((-pow(g2['Z'][0] ** 2 + g2['Z'][1] ** 2, 3, p)) % p, (p + 1) // 4, p))};  // sqrt(-N(Z)^3) in Fp: the norm root of g(x2) from the norm root of g(x1) (g2_sswu)"]
    for nm, c in zip(names, g2["maps"]):
        H.append(f"    static constexpr int G2_{nm}_LEN = {len(c)};")
        H.append(f"    static constexpr uint32_t G2_{nm}[{len(c)}][2][12] = {{" + ", ".join(m2(v) for v in c) + "};")
    H += ["};", "}  // namespace kyb", ""]
    open(os.path.join(ROOT, "kyber_amd", "csrc", "bls12381_h2c_params.h"), "w").write("\n".join(H))
    print("wrote oracle/bls12381_h2c_consts.py and kyber_amd/csrc/bls12381_h2c_params.h",
          [len(c) for c in g1["maps"]], [len(c) for c in g2["maps"]])


if __name__ == "__main__":
    main()
