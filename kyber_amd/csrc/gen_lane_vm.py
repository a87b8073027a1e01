#!/usr/bin/env python3
"""Program generator for the lane machine (lane_vm.cuh): G1 / G2 scalar multiplication of the pairing suites.

The lane machine is the per-wave sibling of the cooperative tower machine (tower_vm.cuh, gen_tower_vm.py): the same
instruction --  out = MontReduce(sum X_t Y_t + R sum Z_t)  over slots holding 14 (10) signed 28-bit limbs, operands
c1 S[a] + c2 S[b], lazy sums, no conditional subtraction -- but every WAVE is independent: lane l owns point l (G1), or
lanes 2l / 2l+1 own the real / imaginary halves of the Fp2 coordinates of point l (G2: an Fp2 product is two base-field
products per lane, the partner's halves arrive through LDS or DPP; an Fp2 square is ONE product per lane).  Seven slots
per lane -- five in LDS, two in registers -- hold the whole working state of a Jacobian point addition; the window
table lives in global memory in the lane's own limb format.  No scratch, ~6 KB of code, two waves per SIMD.

Replaces (kilic/g1.go:110-116 G1Elt.Mul -> MulScalarBig, kilic/g2.go likewise; the per-lane `g1_mul_glv` /
`g2_mul_gls` of bls12381.cuh, which held the table and the digits in scratch at 512 registers): the multiplication
proper.  UnmarshalBinary (decompression, subgroup check) stays with the per-lane code for now; the programs start
from validated affine coordinates.

Algorithm (derived for this design; only canonical encodings of the results are observable):
  * GLV on G1 (k = k0 + k1 z^2, z^2 P = (beta x, -y)), GLS on G2 (k = a0 + a1 |z| + a2 z^2 + a3 |z|^3,
    |z| Q = -psi(Q)) -- the splits of bls12381.cuh (plain long divisions) done by the prep kernel;
  * REGULAR signed odd digits (every sub-scalar is made odd by adding 1 or 2, the surplus is subtracted by one final
    addition of -P or -2P): every lane runs the same records whatever its scalar -- no zero digits, no point at
    infinity on the way, per-lane data are only the table index and the sign;
  * affine table of the odd multiples P, 3P .. 15P (+ 2P) by mixed additions and ONE batched inversion; mixed
    Jacobian additions (11 products) in the main loop;
  * exceptional additions (equal or opposite operands) are not branched on: they leave Z = 0, which is final -- the
    encode kernel sees it and recomputes that lane with the per-lane code (k = 0 mod r and adversarial inputs).

This file also holds the program's exact simulator (LProg.simulate: the device's limb arithmetic with overflow
assertions) and the worst-case bound walk (LProg.check_bounds); tests/test_lane_vm_program.py replays the programs
against the oracle.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_tower_vm import VmField, bls12381_field, emit_field, _carr  # noqa: E402

REC_WORDS = 64
NSLOTS = 6
OP_DOT, OP_IN, OP_OUTW, OP_INV, OP_TSTORE, OP_SELDIGIT, OP_CTRSET, OP_CTRADD, OP_ZFLAG = range(9)
(K_MUL, K_LIN, K_MULC, K_LINC, K_MULT, K_LINT, K2_MUL, K2_SQR, K2_MULT, K2_MULC, K2_NORM, K_SQR) = range(12)
KIND_NAMES = ["mul", "lin", "mulc", "linc", "mult", "lint", "mul2", "sqr2", "mul2t", "mul2c", "norm2", "sqr"]
DYN = -1  # table entry chosen by the lane's current digit (OP_SELDIGIT) instead of a static index


class Lin:
    """c1 S[a] + c2 S[b]: at most two slots, int8 coefficients"""
    __slots__ = ("d",)

    def __init__(self, d):
        self.d = {k: v for k, v in d.items() if v}
        assert 1 <= len(self.d) <= 2 and all(0 <= s < NSLOTS and -128 <= c <= 127 for s, c in self.d.items()), d

    def pairs(self):
        it = sorted(self.d.items())
        return it if len(it) == 2 else [it[0], (it[0][0], 0)]


def L(*a):
    """L(s) -> S[s];  L(s, c) -> c S[s];  L(s1, c1, s2, c2)"""
    if len(a) == 1:
        return Lin({a[0]: 1})
    if len(a) == 2:
        return Lin({a[0]: a[1]})
    d = {}
    for s, c in ((a[0], a[1]), (a[2], a[3])):
        d[s] = d.get(s, 0) + c
    return Lin(d)


class LProg:
    def __init__(self, field, pair, ncoord, name):
        self.f, self.pair, self.ncoord, self.name = field, pair, ncoord, name
        self.recs, self.names, self.sched = [], [], []
        self.consts = []
        self._open = None
        self.n_in = self.n_out = 0
        self.ndigits = 0   # digit bytes per lane

    # ---- constants (stored residues; pair constants take two consecutive entries c0, c1)
    def const(self, stored):
        stored %= self.f.p
        if stored not in self.consts:
            self.consts.append(stored)
        return self.consts.index(stored)

    def const2(self, c0, c1):
        self.consts += [c0 % self.f.p, c1 % self.f.p]
        return len(self.consts) - 2

    def mont(self, x):
        return x % self.f.p * self.f.R % self.f.p

    # ---- schedule
    class _Rep:
        def __init__(self, prog, repeat):
            self.prog, self.repeat = prog, repeat

        def __enter__(self):
            assert self.prog._open is None
            self.prog._open = len(self.prog.recs)

        def __exit__(self, *a):
            start = self.prog._open
            self.prog._open = None
            if len(self.prog.recs) > start and self.repeat > 0:
                self.prog.sched.append((start, len(self.prog.recs) - start, self.repeat))

    def repeat(self, n):
        return LProg._Rep(self, n)

    def block(self, start, ln, repeat=1):
        """schedule an already emitted range of records once more"""
        self.sched.append((start, ln, repeat))

    def _emit(self, r, name):
        self.recs.append(r)
        self.names.append(name)
        if self._open is None:
            self.sched.append((len(self.recs) - 1, 1, 1))

    # ---- instructions
    def dot(self, out, terms, raw=False, negodd=False, name=""):
        """terms: ("mul", X, Y) ("lin", X) ("mulc", X, cidx) ("linc", cidx, coef) ("mult", X, coord, signed, entry)
        ("lint", coord, signed, entry) ("mul2", X, Y) ("sqr2", X) ("mul2t", X, coord, signed, entry) ("mul2c", X, cidx)
        ("norm2", X); entry = DYN or a static index"""
        assert 0 <= out < NSLOTS and 1 <= len(terms) <= 30
        if raw:
            assert all(t[0] in ("lin", "linc", "lint") for t in terms)
        assert all(t[0] != "sqr" for t in terms[1:]), "the symmetric square is the first term of its record"
        if not self.pair:
            assert not negodd and all(not t[0].endswith("2") and t[0] not in ("mul2t", "mul2c") for t in terms)
        self._emit(dict(op=OP_DOT, out=out, terms=list(terms), raw=raw, negodd=negodd), name)

    def op(self, op, out=0, arg=0, name=""):
        self._emit(dict(op=op, out=out, arg=arg), name)

    # ---- executed sequence
    def walk(self):
        for start, ln, rep in self.sched:
            for _ in range(rep):
                for i in range(start, start + ln):
                    yield i, self.recs[i]

    def executed(self):
        return sum(ln * rep for _, ln, rep in self.sched)

    def mads(self, karatsuba=False):
        """integer multiply-adds per LANE (a G2 point is two lanes).  karatsuba: what the same program would cost if an
        Fp2 product took three base-field products per point instead of four (1.5 per lane) -- not expressible with one
        coordinate half per lane, reported as the best known count"""
        n2 = self.f.N * self.f.N
        m2 = 1.5 if karatsuba else 2
        tot = 0
        for _, r in self.walk():
            if r["op"] != OP_DOT:
                continue
            for t in r["terms"]:
                k = t[0]
                tot += {"mul": 1, "mulc": 1, "mult": 1, "mul2": m2, "mul2t": m2, "mul2c": m2, "norm2": 2, "sqr2": 1}.get(k, 0) * n2
                if k == "sqr":
                    tot += self.f.N * (self.f.N + 1) // 2
            if not r["raw"]:
                tot += n2
        return int(tot)

    # ---- the device's arithmetic, limb for limb
    def simulate(self, inputs, digits, trace=None):
        """inputs[lane][k]: plain integers in [0, p) (OP_IN);  digits[lane]: bytes (idx | sign << 7).
        Returns (outputs[lane][k] canonical ints, flags[lane], state).  trace: list to receive (record index, [limbs per
        lane]) after every record that stores a slot."""
        f = self.f
        N, p, W = f.N, f.p, f.W
        pl = f.balanced(p)
        lim63 = 1 << 63
        nl = 2 if self.pair else 1

        def sext(x):
            x &= (1 << W) - 1
            return x - (1 << W) if x >> (W - 1) else x

        def normalise(cols):
            r, carry = [], 0
            for c in range(N - 1):
                v = cols[c] + carry
                assert abs(v) < lim63
                carry = (v + (1 << (W - 1))) >> W
                r.append(sext(v))
            top = cols[N - 1] + carry
            assert abs(top) < 1 << 31, "top limb overflow"
            return r + [top]

        def from_int(x):
            return normalise([(x >> (W * j)) & ((1 << W) - 1) for j in range(N)])

        def canon(l):
            v = f.value(l)
            assert abs(v) < 4 * p, "canon_words needs |v| < 4p"
            return v % p

        S = [[[0] * N for _ in range(NSLOTS)] for _ in range(nl)]
        T = [dict() for _ in range(nl)]
        tsel, tneg = [0] * nl, [0] * nl
        widx = 0
        outs = [dict() for _ in range(nl)]
        flags = [0] * nl
        cl = [f.balanced(c) for c in self.consts]

        def operand(lane, lin):
            v = [0] * N
            for s, c in lin.d.items():
                for i in range(N):
                    v[i] += c * S[lane][s][i]
            assert all(abs(x) < 1 << 31 for x in v), "operand limb overflow"
            return v

        def tabval(lane, coord, signed, entry):
            e = tsel[lane] if entry == DYN else entry
            v = list(T[lane][(e, coord)])
            if signed and tneg[lane]:
                v = [-x for x in v]
            return v

        def mac(t, x, y):
            for i in range(N):
                xi = x[i]
                if xi:
                    for j in range(N):
                        t[i + j] += xi * y[j]
            assert all(abs(c) < lim63 for c in t), "column overflow"

        for idx, r in self.walk():
            op = r["op"]
            new = [None] * nl
            if op == OP_DOT:
                for lane in range(nl):
                    odd, part = lane & 1, lane ^ 1
                    t = [0] * (2 * N)
                    for term in r["terms"]:
                        k = term[0]
                        if k == "lin":
                            x = operand(lane, term[1])
                            for i in range(N):
                                t[N + i] += x[i]
                        elif k == "linc":  # (pair mode: an Fp2 constant, entries idx and idx + 1)
                            c = cl[term[1] + (odd if self.pair else 0)]
                            for i in range(N):
                                t[N + i] += c[i] * term[2]
                        elif k == "lint":
                            x = tabval(lane, term[1], term[2], term[3])
                            for i in range(N):
                                t[N + i] += x[i]
                        elif k == "mul":
                            mac(t, operand(lane, term[1]), operand(lane, term[2]))
                        elif k == "sqr":
                            x = operand(lane, term[1])
                            assert all(abs(2 * v) < 1 << 31 for v in x), "operand limb overflow"
                            mac(t, x, x)
                        elif k == "mulc":
                            mac(t, operand(lane, term[1]), cl[term[2]])
                        elif k == "mult":
                            mac(t, operand(lane, term[1]), tabval(lane, term[2], term[3], term[4]))
                        elif k in ("mul2", "mul2t", "mul2c"):
                            xs, xp = operand(lane, term[1]), operand(part, term[1])
                            if k == "mul2":
                                ys, yp = operand(lane, term[2]), operand(part, term[2])
                            elif k == "mul2t":
                                ys, yp = tabval(lane, term[2], term[3], term[4]), tabval(part, term[2], term[3], term[4])
                            else:
                                ys, yp = cl[term[2] + odd], cl[term[2] + 1 - odd]
                            if odd:   # xp ys + xs yp
                                mac(t, xp, ys)
                                mac(t, xs, yp)
                            else:     # xs ys - xp yp
                                mac(t, xs, ys)
                                mac(t, xp, [-v for v in yp])
                        elif k == "sqr2":
                            xs, xp = operand(lane, term[1]), operand(part, term[1])
                            if odd:
                                X, Y = [2 * v for v in xs], xp
                            else:
                                X, Y = [a + b for a, b in zip(xs, xp)], [a - b for a, b in zip(xs, xp)]
                            assert all(abs(v) < 1 << 31 for v in X + Y), "operand limb overflow"
                            mac(t, X, Y)
                        elif k == "norm2":
                            xs, xp = operand(lane, term[1]), operand(part, term[1])
                            mac(t, xs, xs)
                            mac(t, xp, xp)
                        else:
                            raise ValueError(k)
                        assert all(abs(c) < lim63 for c in t), "column overflow"
                    if not r["raw"]:
                        for i in range(N):
                            m = sext((t[i] & 0xffffffff) * f.ninv)
                            for j in range(N):
                                t[i + j] += m * pl[j]
                            assert all(abs(c) < lim63 for c in t), "column overflow in the reduction"
                            assert t[i] & ((1 << W) - 1) == 0
                            t[i + 1] += t[i] >> W
                    v = normalise(t[N:])
                    if r["negodd"] and odd:
                        v = [-x for x in v]
                    new[lane] = v
            elif op == OP_IN:
                for lane in range(nl):
                    new[lane] = from_int(inputs[lane][r["arg"]])
            elif op == OP_OUTW:
                for lane in range(nl):
                    outs[lane][r["arg"]] = canon(S[lane][r["out"]])
            elif op == OP_INV:
                for lane in range(nl):
                    x = canon(S[lane][r["arg"]])
                    new[lane] = from_int(pow(x, -1, p) * f.R1 * f.R1 % p if x else 0)  # mont.cuh fp_inv: x^-1 R1^2
            elif op == OP_TSTORE:
                for lane in range(nl):
                    T[lane][(r["arg"] >> 8, r["arg"] & 0xff)] = list(S[lane][r["out"]])
            elif op == OP_SELDIGIT:
                for lane in range(nl):
                    b = digits[lane][r["arg"] + widx]
                    tsel[lane], tneg[lane] = b & 15, b >> 7
            elif op == OP_CTRSET:
                widx = r["arg"]
            elif op == OP_CTRADD:
                widx += r["arg"]
            elif op == OP_ZFLAG:
                for lane in range(nl):
                    if canon(S[lane][r["out"]]) == 0:
                        flags[lane] |= 1 << r["arg"]
            else:
                raise ValueError(op)
            if new[0] is not None:
                for lane in range(nl):
                    S[lane][r["out"]] = new[lane]
                if trace is not None:
                    trace.append((idx, [list(v) for v in new]))
        self.last_table = T
        return outs, flags, S

    # ---- worst-case bounds for any input
    def check_bounds(self):
        """Per slot: lb = bound on |limb| / 2^(W-1) for limbs 0..N-2 (after a store: 1), vb = bound on |value| / p.
        Checks operand limbs < 2^31, columns < 2^63, top limbs < 2^31, canon inputs < 4p.  Returns the largest
        column seen (log2)."""
        import math
        f = self.f
        N, W, p = f.N, f.W, f.p
        half = 1 << (W - 1)
        pR = p / f.R
        top_unit = p / float(1 << (W * (N - 1)))  # top limb of a value of magnitude p
        vb = [0.0] * NSLOTS
        worst = 0.0
        tab_vb = {}

        def tabb(coord, entry):
            if entry != DYN:
                return tab_vb[(entry, coord)]
            return max(tab_vb[(e, coord)] for e in range(NENTRY))

        def opb(lin):
            c = sum(abs(k) for k in lin.d.values())
            v = sum(abs(k) * vb[s] for s, k in lin.d.items())
            assert c * half < 1 << 31 and v * top_unit + c * 1 < 1 << 31, "operand limb bound"
            return float(c), v

        for idx, r in self.walk():
            op = r["op"]
            if op == OP_DOT:
                col, val, lin = 0.0, 0.0, 0.0
                for term in r["terms"]:
                    k = term[0]
                    if k == "lin":
                        c, v = opb(term[1])
                        lin += v
                        col += c * half
                    elif k == "linc":
                        lin += abs(term[2])
                        col += abs(term[2]) * half
                    elif k == "lint":
                        lin += tabb(term[1], term[3])
                        col += half
                    else:
                        cx, vx = opb(term[1])
                        if k in ("mul", "mul2"):
                            cy, vy = opb(term[2])
                        elif k in ("sqr2", "norm2", "sqr"):
                            cy, vy = cx, vx
                        elif k in ("mult", "mul2t"):
                            cy, vy = 1.0, tabb(term[2], term[4])
                        else:
                            cy, vy = 1.0, 1.0
                        # limb products: N per column; the top limbs are bounded by the value bounds
                        mult = {"mul": 1, "sqr": 1, "mulc": 1, "mult": 1, "mul2": 2, "mul2t": 2, "mul2c": 2, "norm2": 2, "sqr2": 4}[k]
                        lx = max(cx * half, vx * top_unit + cx)
                        ly = max(cy * half, vy * top_unit + cy)
                        col += mult * N * lx * ly
                        val += {"sqr2": 2, "mul2": 2, "mul2t": 2, "mul2c": 2, "norm2": 2}.get(k, 1) * vx * vy
                if not r["raw"]:
                    col += N * half * half
                    out_v = val * p * pR / p + 0.5 + lin + 2 ** -20
                else:
                    assert val == 0
                    out_v = lin
                assert col < 2 ** 63, ("column bound", self.names[idx], math.log2(col))
                worst = max(worst, col)
                assert out_v * top_unit + 1 < 1 << 31, ("top limb bound", self.names[idx])
                assert out_v < 2 ** 10
                vb[r["out"]] = out_v
            elif op == OP_IN:
                vb[r["out"]] = 1.0
            elif op == OP_INV:
                assert vb[r["arg"]] < 4, ("canon bound before the inversion", self.names[idx], vb[r["arg"]])
                vb[r["out"]] = 1.0
            elif op in (OP_OUTW, OP_ZFLAG):
                assert vb[r["out"]] < 4, ("canon bound", self.names[idx], vb[r["out"]])
            elif op == OP_TSTORE:
                tab_vb[(r["arg"] >> 8, r["arg"] & 0xff)] = vb[r["out"]]
        return math.log2(worst)

    # ---- encoding
    def encode(self):
        words = []
        for r in self.recs:
            rec = [0] * REC_WORDS
            if r["op"] == OP_DOT:
                rec[0] = OP_DOT | (r["out"] << 4) | (len(r["terms"]) << 7) | (int(r["raw"]) << 12) | (int(r["negodd"]) << 13)
                for k, t in enumerate(r["terms"]):
                    kind = KIND_NAMES.index(t[0])
                    w0 = kind << 24
                    cs = [0, 0, 0, 0]
                    if t[0] in ("linc",):
                        w0 |= t[1] << 12
                        cs[0] = t[2]
                    elif t[0] == "lint":
                        w0 |= (t[1] << 12) | (int(t[2]) << 28) | (int(t[3] != DYN) << 29)
                        cs[2] = 0 if t[3] == DYN else t[3]
                    else:
                        (x1, c1), (x2, c2) = t[1].pairs()
                        w0 |= x1 | (x2 << 3)
                        cs[0], cs[1] = c1, c2
                        if t[0] in ("mul", "mul2"):
                            (y1, d1), (y2, d2) = t[2].pairs()
                            w0 |= (y1 << 6) | (y2 << 9)
                            cs[2], cs[3] = d1, d2
                        elif t[0] in ("mulc", "mul2c"):
                            w0 |= t[2] << 12
                        elif t[0] in ("mult", "mul2t"):
                            w0 |= (t[2] << 12) | (int(t[3]) << 28) | (int(t[4] != DYN) << 29)
                            cs[2] = 0 if t[4] == DYN else t[4]
                    rec[2 + 2 * k] = w0
                    rec[3 + 2 * k] = sum((c & 0xff) << (8 * i) for i, c in enumerate(cs))
            else:
                rec[0] = r["op"] | (r["out"] << 4)
                rec[1] = r["arg"] & 0xffffffff
            words += rec
        return words, [x for s in self.sched for x in (s[0], s[1], s[2], 0)]

    def emit(self, up):
        prog, sched = self.encode()
        consts = []
        for c in self.consts:
            consts += [d & 0xffffffff for d in self.f.balanced(c)] + [0] * (16 - self.f.N)
        return "\n".join([
            f"// program {up}: {len(self.recs)} stored records, {self.executed()} executed, {self.mads()} multiply-adds per lane",
            f"static __device__ const uint32_t LVM_{up}_PROG[{len(prog)}] = {_carr(prog)};",
            f"static __device__ const uint32_t LVM_{up}_SCHED[{len(sched)}] = {_carr(sched)};",
            f"static constexpr uint32_t LVM_{up}_NSCHED = {len(self.sched)};",
            f"static __device__ const uint32_t LVM_{up}_CONSTS[{len(consts)}] = {_carr(consts)};",
            f"static constexpr uint32_t LVM_{up}_NCONSTS = {len(self.consts)}, LVM_{up}_NIN = {self.n_in}, LVM_{up}_NOUT = {self.n_out};",
            f"static constexpr uint32_t LVM_{up}_NCOORD = {self.ncoord}, LVM_{up}_NENTRY = {NENTRY}, LVM_{up}_NDIGITS = {self.ndigits};",
            f"static constexpr uint64_t LVM_{up}_MADS_PER_LANE = {self.mads()}ull;",
            ""])


# ------------------------------------------------------------------------------------------------ curve programs
# slots: the accumulator T = (X, Y, Z) and three temporaries (five slots live in LDS, the sixth in registers)
SX, SY, SZ, SA, SB, SC = range(6)
NENTRY = 9          # table entries: (2 i + 1) P for i = 0..7, entry 8 = 2 P
E2P = 8


class Curve:
    """Emits the point formulas for one mode: Fp (G1) or lane pairs over Fp2 (G2)."""

    def __init__(self, P):
        self.P, self.pair = P, P.pair

    def mul(self, X, Y):
        return ("mul2", X, Y) if self.pair else ("mul", X, Y)

    def sqr(self, X):
        return ("sqr2", X) if self.pair else ("sqr", X)

    def mult(self, X, coord, signed=False, entry=DYN):
        return ("mul2t", X, coord, signed, entry) if self.pair else ("mult", X, coord, signed, entry)

    def dbl(self, name="dbl"):
        """(X, Y, Z) <- 2 (X, Y, Z), Jacobian, a = 0.  Temporaries A, B, C."""
        P = self.P
        if not self.pair:
            # A = X^2, B = Y^2, Z3 = 2 Y Z, C = X B, X3 = 9 A^2 - 8 C, Y3 = 3 A (4 C - X3) - 8 B^2: 7 products
            P.dot(SA, [self.sqr(L(SX))], name=name + ".A")
            P.dot(SB, [self.sqr(L(SY))], name=name + ".B")
            P.dot(SZ, [self.mul(L(SY, 2), L(SZ))], name=name + ".Z")
            P.dot(SC, [self.mul(L(SX), L(SB))], name=name + ".C")
            P.dot(SX, [self.sqr(L(SA, 3)), ("lin", L(SC, -8))], name=name + ".X")
            P.dot(SY, [self.mul(L(SA, 3), L(SC, 4, SX, -1)), self.mul(L(SB, -8), L(SB))], name=name + ".Y")
        else:
            # squarings are one product per lane, multiplications two: dbl-2009-l (2M + 5S) with D' = D / 2 carried, so
            # that the difference (X + B)^2 - A - C rides on the squaring's reduction: seven records
            P.dot(SA, [self.sqr(L(SX))], name=name + ".A")                      # A = X^2
            P.dot(SB, [self.sqr(L(SY))], name=name + ".B")                      # B = Y^2
            P.dot(SZ, [self.mul(L(SY, 2), L(SZ))], name=name + ".Z")            # Z3 = 2 Y Z
            P.dot(SC, [self.sqr(L(SB))], name=name + ".C")                      # C = B^2
            P.dot(SY, [self.sqr(L(SX, 1, SB, 1)), ("lin", L(SA, -1, SC, -1))], name=name + ".D")  # D' = (X + B)^2 - A - C  (in Y)
            # X3 = (3A)^2 - 4 D' as a PRODUCT of 3A with itself: the one-product square (3 (a0 + a1)) (3 (a0 - a1)) on
            # unnormalised limbs would put 504 x 2^54 in a column; storing 3A first costs a record more than the
            # second pass does
            P.dot(SX, [self.mul(L(SA, 3), L(SA, 3)), ("lin", L(SY, -4))], name=name + ".X3")
            P.dot(SY, [self.mul(L(SA, 3), L(SY, 2, SX, -1)), ("lin", L(SC, -8))], name=name + ".Y3")  # Y3 = 3A (2 D' - X3) - 8 C

    def madd(self, cx, cy, entry=DYN, signed=True, name="madd"):
        """(X, Y, Z) <- (X, Y, Z) + (x2, y2), the second operand affine from the table (coordinates cx, cy of `entry`;
        y2 negated by the digit's sign).  madd-2004-hmv: H = U2 - X1, r = S2 - Y1, Z3 = Z1 H, X3 = r^2 - H^3 - 2 X1 H^2,
        Y3 = r (X1 H^2 - X3) - Y1 H^3.  11 products in ten records (H and r are formed on the reductions of U2 and S2: as
        stored values they keep the working set at six slots), temporaries A, B, C.  H = 0 (equal or opposite operands)
        gives Z3 = 0, which every later step preserves."""
        P = self.P
        P.dot(SA, [self.sqr(L(SZ))], name=name + ".ZZ")                                     # A = Z^2
        P.dot(SB, [self.mult(L(SA), cx, False, entry), ("lin", L(SX, -1))], name=name + ".H")   # B = H = x2 Z^2 - X1
        P.dot(SA, [self.mul(L(SA), L(SZ))], name=name + ".ZZZ")                             # A = Z^3
        P.dot(SC, [self.mult(L(SA), cy, signed, entry), ("lin", L(SY, -1))], name=name + ".r")  # C = r = y2 Z^3 - Y1
        P.dot(SZ, [self.mul(L(SZ), L(SB))], name=name + ".Z3")                              # Z3 = Z H
        P.dot(SA, [self.sqr(L(SB))], name=name + ".HH")                                     # A = H^2
        P.dot(SB, [self.mul(L(SA), L(SB))], name=name + ".HHH")                             # B = H^3
        P.dot(SA, [self.mul(L(SX), L(SA))], name=name + ".V")                               # A = V = X1 H^2
        P.dot(SX, [self.sqr(L(SC)), ("lin", L(SB, -1, SA, -2))], name=name + ".X3")
        P.dot(SY, [self.mul(L(SC), L(SA, 1, SX, -1)), self.mul(L(SY, -1), L(SB))], name=name + ".Y3")

    def load_affine(self, cx, cy, entry=DYN, signed=True, name="load"):
        P = self.P
        P.dot(SX, [("lint", cx, False, entry)], name=name + ".x")
        P.dot(SY, [("lint", cy, signed, entry)], name=name + ".y")
        P.dot(SZ, [("linc", self.c_one, 1)], name=name + ".one")

    def inverse(self, dst, src, tmp, name="inv"):
        """dst <- src^-1 (Montgomery form in, Montgomery form out); src is left alone; tmp is clobbered (pair mode)."""
        P = self.P
        if not self.pair:
            P.op(OP_INV, out=dst, arg=src, name=name + ".inv")
            P.dot(dst, [("mulc", L(dst), self.c_invfix)], name=name + ".fix")
        else:
            P.dot(tmp, [("norm2", L(src))], name=name + ".norm")          # a0^2 + a1^2 (the same in both lanes)
            P.op(OP_INV, out=tmp, arg=tmp, name=name + ".inv")
            P.dot(tmp, [("mulc", L(tmp), self.c_invfix)], name=name + ".fix")
            P.dot(dst, [("mul", L(src), L(tmp))], negodd=True, name=name + ".conj")  # conj(a) / norm

    def consts(self):
        P, f = self.P, self.P.f
        self.c_one = P.const2(f.R % f.p, 0) if self.pair else P.const(f.R % f.p)   # Montgomery one (pair: 1 + 0 i)
        self.c_r2 = P.const(f.R * f.R % f.p)                  # plain x -> x R
        self.c_plain = P.const(1)                             # x R -> x
        # OP_INV: v -> v^-1 R1^2 (mont.cuh, radix R1); for v = a R we want a^-1 R: multiply by R^3 / R1^2 (and divide by R
        # in the reduction)
        self.c_invfix = P.const(pow(f.R, 3, f.p) * pow(f.R1 * f.R1, -1, f.p) % f.p)


def build_mul(field, pair, nsub, npos, variants, name):
    """k P by `nsub` sub-scalars of `npos` regular signed radix-16 digits each over the table variants `variants`
    (list of (cx, cy) coordinate pairs: G1 [(x, y), (beta x, y)]; G2 the four psi^j images).
    Inputs (OP_IN): 0 = x, 1 = y (plain, canonical; a lane of a pair holds its own half).
    Digits per lane: sub-scalar j at bytes [j (npos + 1), (j + 1)(npos + 1)): positions 0 .. npos-1, then the correction.
    Outputs (OP_OUTW): 0 = x, 1 = y of the result (plain, canonical) -- meaningless when the flag says Z = 0."""
    ncoord = 2 * len(variants) if pair else 4
    P = LProg(field, pair, ncoord, name)
    C = Curve(P)
    C.consts()
    P.n_in, P.n_out, P.ndigits = 2, 2, nsub * (npos + 1)
    return P, C


def table_g1(P, C, beta_mont):
    """entries (x, beta x, y, scratch): coordinates 0..3"""
    c_beta = P.const(beta_mont)
    CX, CBX, CY, CS = 0, 1, 2, 3

    def store_affine(entry, sx, sy, tmp):
        P.op(OP_TSTORE, out=sx, arg=(entry << 8) | CX, name="tab%d.x" % entry)
        P.dot(tmp, [("mulc", L(sx), c_beta)], name="tab%d.bx" % entry)
        P.op(OP_TSTORE, out=tmp, arg=(entry << 8) | CBX)
        P.op(OP_TSTORE, out=sy, arg=(entry << 8) | CY)

    # P itself
    P.op(OP_IN, out=SA, arg=0, name="in.x")
    P.dot(SX, [("mulc", L(SA), C.c_r2)], name="x.mont")
    P.op(OP_IN, out=SA, arg=1, name="in.y")
    P.dot(SY, [("mulc", L(SA), C.c_r2)], name="y.mont")
    P.dot(SZ, [("linc", C.c_one, 1)], name="z.one")
    store_affine(0, SX, SY, SA)
    # 2 P, made affine at once: every later addition is a mixed one
    C.dbl("dbl2p")
    C.inverse(SA, SZ, SB, "inv2p")                               # A = 1 / Z
    P.dot(SB, [C.sqr(L(SA))], name="2p.zi2")
    P.dot(SC, [C.mul(L(SX), L(SB))], name="2p.x")
    P.dot(SB, [C.mul(L(SB), L(SA))], name="2p.zi3")
    P.dot(SA, [C.mul(L(SY), L(SB))], name="2p.y")
    store_affine(E2P, SC, SA, SB)
    # T = P, then T += 2P seven times: 3P .. 15P in Jacobian form, parked in the table (X -> x, Z -> beta x, Y -> y)
    C.load_affine(CX, CY, entry=0, signed=False, name="t=p")
    for e in range(1, 8):
        C.madd(CX, CY, entry=E2P, signed=False, name="odd%d" % e)
        P.op(OP_TSTORE, out=SX, arg=(e << 8) | CX)
        P.op(OP_TSTORE, out=SZ, arg=(e << 8) | CBX)
        P.op(OP_TSTORE, out=SY, arg=(e << 8) | CY)
    # batched inversion of Z_1 .. Z_7: prefix products c_e in the scratch coordinate
    P.dot(SC, [("lint", CBX, False, 1)], name="c1")
    P.op(OP_TSTORE, out=SC, arg=(1 << 8) | CS)
    for e in range(2, 8):
        P.dot(SC, [C.mult(L(SC), CBX, False, e)], name="c%d" % e)
        P.op(OP_TSTORE, out=SC, arg=(e << 8) | CS)
    C.inverse(SZ, SC, SA, "invtab")                              # Z = 1 / (Z_1 .. Z_7): the running inverse
    for e in range(7, 0, -1):
        if e > 1:
            P.dot(SA, [C.mult(L(SZ), CS, False, e - 1)], name="zi%d" % e)       # A = 1 / Z_e
            P.dot(SZ, [C.mult(L(SZ), CBX, False, e)], name="run%d" % e)         # Z = 1 / (Z_1 .. Z_{e-1})
        else:
            P.dot(SA, [("lin", L(SZ))], raw=True, name="zi1")
        P.dot(SB, [C.sqr(L(SA))], name="zi2.%d" % e)
        P.dot(SX, [C.mult(L(SB), CX, False, e)], name="x%d" % e)
        P.dot(SB, [C.mul(L(SB), L(SA))], name="zi3.%d" % e)
        P.dot(SY, [C.mult(L(SB), CY, False, e)], name="y%d" % e)
        store_affine(e, SX, SY, SA)
    return [(CX, CY), (CBX, CY)]


def table_g2(P, C, psi):
    """entries: coordinates 2 j, 2 j + 1 = x, y of psi^j of the multiple; psi = dict of Montgomery constants
    cx, cy (Fp2), nx, ny (Fp), cx3, cy3 (Fp2)"""
    c_cx = P.const2(*psi["cx"])
    c_cy = P.const2(*psi["cy"])
    c_cx3 = P.const2(*psi["cx3"])
    c_cy3 = P.const2(*psi["cy3"])
    c_nx = P.const(psi["nx"])
    c_ny = P.const(psi["ny"])

    def store_affine(entry, sx, sy, t1, t2):
        """x, y (left intact) and their images under psi, psi^2, psi^3"""
        P.op(OP_TSTORE, out=sx, arg=(entry << 8) | 0, name="tab%d" % entry)
        P.op(OP_TSTORE, out=sy, arg=(entry << 8) | 1)
        P.dot(t1, [("lin", L(sx))], raw=True, negodd=True, name="conj.x")
        P.dot(t2, [("mul2c", L(t1), c_cx)], name="psi.x")
        P.op(OP_TSTORE, out=t2, arg=(entry << 8) | 2)
        P.dot(t2, [("mul2c", L(t1), c_cx3)], name="psi3.x")
        P.op(OP_TSTORE, out=t2, arg=(entry << 8) | 6)
        P.dot(t1, [("lin", L(sy))], raw=True, negodd=True, name="conj.y")
        P.dot(t2, [("mul2c", L(t1), c_cy)], name="psi.y")
        P.op(OP_TSTORE, out=t2, arg=(entry << 8) | 3)
        P.dot(t2, [("mul2c", L(t1), c_cy3)], name="psi3.y")
        P.op(OP_TSTORE, out=t2, arg=(entry << 8) | 7)
        P.dot(t2, [("mulc", L(sx), c_nx)], name="psi2.x")
        P.op(OP_TSTORE, out=t2, arg=(entry << 8) | 4)
        P.dot(t2, [("mulc", L(sy), c_ny)], name="psi2.y")
        P.op(OP_TSTORE, out=t2, arg=(entry << 8) | 5)

    CX, CY, CZ, CS = 0, 1, 2, 3   # while the multiples are Jacobian: X, Y, Z and the prefix product
    P.op(OP_IN, out=SA, arg=0, name="in.x")
    P.dot(SX, [("mulc", L(SA), C.c_r2)], name="x.mont")
    P.op(OP_IN, out=SA, arg=1, name="in.y")
    P.dot(SY, [("mulc", L(SA), C.c_r2)], name="y.mont")
    P.dot(SZ, [("linc", C.c_one, 1)], name="z.one")
    store_affine(0, SX, SY, SA, SB)
    C.dbl("dbl2p")
    C.inverse(SA, SZ, SB, "inv2p")
    P.dot(SB, [C.sqr(L(SA))], name="2p.zi2")
    P.dot(SC, [C.mul(L(SX), L(SB))], name="2p.x")
    P.dot(SB, [C.mul(L(SB), L(SA))], name="2p.zi3")
    P.dot(SA, [C.mul(L(SY), L(SB))], name="2p.y")
    store_affine(E2P, SC, SA, SB, SX)
    C.load_affine(CX, CY, entry=0, signed=False, name="t=p")
    for e in range(1, 8):
        C.madd(CX, CY, entry=E2P, signed=False, name="odd%d" % e)
        P.op(OP_TSTORE, out=SX, arg=(e << 8) | CX)
        P.op(OP_TSTORE, out=SY, arg=(e << 8) | CY)
        P.op(OP_TSTORE, out=SZ, arg=(e << 8) | CZ)
    P.dot(SC, [("lint", CZ, False, 1)], name="c1")
    P.op(OP_TSTORE, out=SC, arg=(1 << 8) | CS)
    for e in range(2, 8):
        P.dot(SC, [C.mult(L(SC), CZ, False, e)], name="c%d" % e)
        P.op(OP_TSTORE, out=SC, arg=(e << 8) | CS)
    C.inverse(SZ, SC, SA, "invtab")                              # Z: the running inverse
    for e in range(7, 0, -1):
        if e > 1:
            P.dot(SA, [C.mult(L(SZ), CS, False, e - 1)], name="zi%d" % e)
            P.dot(SZ, [C.mult(L(SZ), CZ, False, e)], name="run%d" % e)
        else:
            P.dot(SA, [("lin", L(SZ))], raw=True, name="zi1")
        P.dot(SB, [C.sqr(L(SA))], name="zi2.%d" % e)
        P.dot(SX, [C.mult(L(SB), CX, False, e)], name="x%d" % e)
        P.dot(SB, [C.mul(L(SB), L(SA))], name="zi3.%d" % e)
        P.dot(SY, [C.mult(L(SB), CY, False, e)], name="y%d" % e)
        # (Z, the running inverse, must survive: the images are built in A, B)
        store_affine(e, SX, SY, SA, SB)
    return [(0, 1), (2, 3), (4, 5), (6, 7)]


def ladder(P, C, variants, npos):
    """T = sum_j (digits of sub-scalar j) over the table variants, then the corrections, then the affine result."""
    nsub = len(variants)
    stride = npos + 1
    P.op(OP_CTRSET, arg=npos - 1, name="top")
    for j, (cx, cy) in enumerate(variants):
        P.op(OP_SELDIGIT, arg=j * stride, name="digit%d" % j)
        if j == 0:
            C.load_affine(cx, cy, name="init")
        else:
            C.madd(cx, cy, name="init%d" % j)
    d0 = len(P.recs)
    with P.repeat(4):
        C.dbl("dbl")
    a0 = len(P.recs)
    with P.repeat(1):
        P.op(OP_CTRADD, arg=-1, name="next")
        for j, (cx, cy) in enumerate(variants):
            P.op(OP_SELDIGIT, arg=j * stride, name="digit%d" % j)
            C.madd(cx, cy, name="add%d" % j)
    a1 = len(P.recs)
    for _ in range(npos - 2):
        P.block(d0, a0 - d0, 4)
        P.block(a0, a1 - a0, 1)
    # corrections: the digit byte after the last position
    P.op(OP_CTRSET, arg=npos, name="corr")
    for j, (cx, cy) in enumerate(variants):
        P.op(OP_SELDIGIT, arg=j * stride, name="cdigit%d" % j)
        C.madd(cx, cy, name="corr%d" % j)
    # affine result; Z = 0 <=> infinity or an exceptional addition on the way: flagged, the outputs are then void
    P.op(OP_ZFLAG, out=SZ, arg=0, name="zflag")
    C.inverse(SA, SZ, SB, "invout")
    P.dot(SB, [C.sqr(L(SA))], name="out.zi2")
    P.dot(SC, [C.mul(L(SX), L(SB))], name="out.x")
    P.dot(SB, [C.mul(L(SB), L(SA))], name="out.zi3")
    P.dot(SA, [C.mul(L(SY), L(SB))], name="out.y")
    P.dot(SC, [("mulc", L(SC), C.c_plain)], name="out.x.plain")
    P.dot(SA, [("mulc", L(SA), C.c_plain)], name="out.y.plain")
    P.op(OP_OUTW, out=SC, arg=0)
    P.op(OP_OUTW, out=SA, arg=1)


BLS_P = bls12381_field().p
BLS_BETA = pow(2, (BLS_P - 1) // 3, BLS_P)
G1_NPOS, G2_NPOS = 33, 17
BLS_G1 = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
          0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)
BLS_G2 = ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
           0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
          (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
           0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))


def _f2_mul(a, b, p):
    return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def _f2_pow(a, e, p):
    r = (1, 0)
    while e:
        if e & 1:
            r = _f2_mul(r, a, p)
        a = _f2_mul(a, a, p)
        e >>= 1
    return r


def _f2_inv(a, p):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, p)
    return (a[0] * n % p, -a[1] * n % p)


def bls_psi_consts(f):
    p = f.p
    xi = (1, 1)
    cx = _f2_inv(_f2_pow(xi, (p - 1) // 3, p), p)
    cy = _f2_inv(_f2_pow(xi, (p - 1) // 2, p), p)
    nx = (cx[0] * cx[0] + cx[1] * cx[1]) % p
    ny = (cy[0] * cy[0] + cy[1] * cy[1]) % p
    m = lambda v: v % p * f.R % p
    return dict(cx=(m(cx[0]), m(cx[1])), cy=(m(cy[0]), m(cy[1])), nx=m(nx), ny=m(ny),
                cx3=(m(cx[0] * nx), m(cx[1] * nx)), cy3=(m(cy[0] * ny), m(cy[1] * ny)),
                plain=dict(cx=cx, cy=cy, nx=nx, ny=ny))


def build_bls12381_g1_mul():
    f = bls12381_field()
    P, C = build_mul(f, False, 2, G1_NPOS, [None, None], "bls12381_g1_mul")
    variants = table_g1(P, C, BLS_BETA * f.R % f.p)
    ladder(P, C, variants, G1_NPOS)
    return P


def build_bls12381_g2_mul():
    f = bls12381_field()
    P, C = build_mul(f, True, 4, G2_NPOS, [None] * 4, "bls12381_g2_mul")
    variants = table_g2(P, C, bls_psi_consts(f))
    ladder(P, C, variants, G2_NPOS)
    return P


# ------------------------------------------------------------------------------------------------ host-side recoding
# (the reference for the prep kernel's digit bytes; tests replay the programs with these)
BLS_Z = 0xD201000000010000


def regular_digits(k, npos):
    """k odd, 0 < k < 16^npos: npos signed odd digits d_i (|d_i| <= 15), sum d_i 16^i = k"""
    assert k & 1 and 0 < k < 16 ** npos
    out = []
    for i in range(npos - 1):
        out.append((((k >> (4 * i)) | 1) & 31) - 16)
    out.append((k >> (4 * (npos - 1))) | 1)
    assert sum(d << (4 * i) for i, d in enumerate(out)) == k and out[-1] < 16
    return out


def digit_bytes(sub, npos, flip):
    """one sub-scalar (any non-negative integer below 16^npos - 2) -> npos digit bytes + the correction byte.
    byte = table index | sign << 7.  flip: the sub-scalar multiplies the NEGATIVE of its table variant."""
    odd = sub & 1
    k = sub + (2 if odd else 1)
    out = []
    for d in regular_digits(k, npos):
        neg = (d < 0) != flip
        out.append(((abs(d) - 1) // 2) | (0x80 if neg else 0))
    # surplus: 1 (even sub-scalar: k = sub + 1) or 2 -> subtract entry 0 (P) or entry 8 (2P)
    out.append((E2P if odd else 0) | (0 if flip else 0x80))
    return bytes(out)


def bls_g1_digits(k):
    """k P = k0 P + k1 z^2 P with z^2 P = (beta x, -y): the second variant (beta x, y) enters negated"""
    z2 = BLS_Z * BLS_Z
    return digit_bytes(k % z2, G1_NPOS, False) + digit_bytes(k // z2, G1_NPOS, True)


def bls_g2_digits(k):
    """k Q = a0 Q - a1 psi(Q) + a2 psi^2(Q) - a3 psi^3(Q)"""
    out = b""
    for j in range(4):
        a = k % BLS_Z if j < 3 else k
        k //= BLS_Z
        out += digit_bytes(a, G2_NPOS, bool(j & 1))
    return out


def main():
    out = ["// generated by gen_lane_vm.py -- do not edit", "#pragma once", "#include <stdint.h>", "namespace kyb {"]
    f = bls12381_field()
    out.append(emit_field(f, "Bls12381Lvm"))

    def words(x):
        return "{" + ", ".join("0x%xu" % ((x >> (32 * i)) & 0xffffffff) for i in range(12)) + "}"
    # the generators, plain canonical words: what the prep kernel feeds the machine for elements it does not compute
    out.append("static __device__ const uint32_t LVM_BLS12381_G1_GEN[2][12] = {%s, %s};" % (words(BLS_G1[0]), words(BLS_G1[1])))
    out.append("static __device__ const uint32_t LVM_BLS12381_G2_GEN[4][12] = {%s};  // x.c0, x.c1, y.c0, y.c1"
               % ", ".join(words(v) for v in (BLS_G2[0][0], BLS_G2[0][1], BLS_G2[1][0], BLS_G2[1][1])))
    for up, build in (("BLS12381_G1_MUL", build_bls12381_g1_mul), ("BLS12381_G2_MUL", build_bls12381_g2_mul)):
        P = build()
        out.append(P.emit(up))
        print(up, "records", len(P.recs), "executed", P.executed(), "mads/lane", P.mads(), "column bound 2^%.2f" % P.check_bounds())
    out += ["}  // namespace kyb", ""]
    dst = os.path.join(HERE, "lane_vm_bls12381.inc")
    with open(dst + ".tmp", "w") as fh:
        fh.write("\n".join(out))
    os.replace(dst + ".tmp", dst)


if __name__ == "__main__":
    main()
