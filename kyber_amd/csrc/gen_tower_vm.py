#!/usr/bin/env python3
"""Program generator for the cooperative tower machine (tower_vm.cuh).

A program is data: a list of instructions, each giving every one of the twelve waves of a workgroup one output slot and
a list of terms  out = MontReduce(sum X_t Y_t + R sum Z_t)  over LDS slots (operands = c1 S[a] + c2 S[b]).  This file
builds the programs for the pairings symbolically -- the tower (Fp2 = Fp[i]/(i^2+1), Fp12 = Fp2[w]/(w^6 - xi)) is
expanded into base-field bilinear forms here, never on the device -- and provides

  * Prog.simulate(): the program run on stored residues mod p (exact meaning of every instruction);
  * Prog.simulate_limbs(): the same with the device's arithmetic (14 balanced 28-bit limbs, 64-bit columns, the signed
    Montgomery reduction), asserting that nothing overflows;
  * Prog.check_bounds(): worst-case bounds on operand limbs, accumulator columns and values for ANY input in [0, p);
  * emit(): the generated header kyber_amd/csrc/tower_vm_<suite>.inc.

tests/test_tower_vm_program.py replays the generated programs against the oracle's pairing (the oracle is not used
here: this is product tooling).  What the programs compute, with the reference functions they replace:
  bls12381 PAIR   Suite.Pair              kilic/suite.go:70-75   Miller loop f_{|x|,Q}(P) conjugated + final exponentiation
  bls12381 CHECK  Suite.ValidatePairing   kilic/suite.go:57-68   product of two Miller loops (shared squarings), f^e == 1
Curve formulas: homogeneous projective doubling / mixed addition on the M-type twist y^2 = x^3 + 4 xi with the tangent /
chord evaluated at P as the sparse element l0 + l2 w^2 + l3 w^3 (derivation in DESIGN.md section 4a).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
WAVES = 12
W = 28  # default limb width; a field may choose a narrower one (VmField.W)
REC_WORDS = 64
OP_DOT, OP_IDLE, OP_GLOAD, OP_INV, OP_GT_STORE, OP_IS_ONE, OP_CLOAD, OP_SPILL, OP_FILL, OP_CMP_EQ, OP_GCLOAD, OP_CMP_NZ2 = range(12)
# the comparison records (IS_ONE, CMP_EQ, CMP_NZ2) raise a bit of the lane's result flags: FLAG_VERDICT feeds the boolean of
# a check program / rejects a GT element; FLAG_G2_A / FLAG_G2_B say that the G2 operand of pair A / B is outside the
# order-r subgroup (decided at the end of the Miller loop: bls_g2_member_check)
FLAG_VERDICT, FLAG_G2_A, FLAG_G2_B = 1, 2, 4
# raised when a coefficient of the joint Miller value of a product-form check is NON-zero: a lane that leaves the program
# without it had a zero Miller value -- the one case in which the product form and "two pairings + Equal" differ (bn256's
# G2 accepts points of small order) -- and is handed to the two-pairing program (tower_vm.cuh Args::redo_out / only)
FLAG_MILLER_NONZERO = 8
K_PROD, K_LIN, K_PROD_CONST = 0, 1, 2
GC_ENTRIES = 4  # table entries one OP_GCLOAD record moves into the constant area (64 lanes x 1 word = 4 x 16 words)


# ------------------------------------------------------------------------------------------------ field
class VmField:
    def __init__(self, name, p, n, nw, r1_bits, w=W):
        self.name, self.p, self.N, self.NW, self.W = name, p, n, nw, w
        self.R = 1 << (w * n)
        self.R1 = 1 << r1_bits  # Montgomery radix of the per-lane field code (mont.cuh) that decodes the inputs
        self.ninv = (-pow(p, -1, 1 << w)) % (1 << w)
        assert self.R > 256 * p and 32 * nw <= w * n + 31

    def balanced(self, x):
        """signed digits d_i in [-2^(W-1), 2^(W-1)) (top digit unbounded) with sum d_i 2^(W i) = x"""
        W = self.W
        out = []
        for _ in range(self.N - 1):
            d = x & ((1 << W) - 1)
            if d >= 1 << (W - 1):
                d -= 1 << W
            out.append(d)
            x = (x - d) >> W
        out.append(x)
        assert abs(x) < 1 << 31
        return out

    def value(self, limbs):
        return sum(d << (self.W * i) for i, d in enumerate(limbs))


# ------------------------------------------------------------------------------------------------ symbolic layer
class Lin:
    """integer combination of slots"""
    __slots__ = ("d",)

    def __init__(self, d=None):
        self.d = {k: v for k, v in (d or {}).items() if v}

    @staticmethod
    def slot(s):
        return Lin({s: 1})

    def __add__(self, o):
        d = dict(self.d)
        for k, v in o.d.items():
            d[k] = d.get(k, 0) + v
        return Lin(d)

    def __neg__(self):
        return Lin({k: -v for k, v in self.d.items()})

    def __sub__(self, o):
        return self + (-o)

    def scale(self, c):
        return Lin({k: v * c for k, v in self.d.items()})

    def is_zero(self):
        return not self.d

    def items(self):
        return sorted(self.d.items())

    def __repr__(self):
        return "Lin(%s)" % self.items()


class E2:
    """Fp2 operand re + im i, both integer combinations of slots"""
    __slots__ = ("re", "im")

    def __init__(self, re, im):
        self.re, self.im = re, im

    @staticmethod
    def slots(sre, sim):
        return E2(Lin.slot(sre), Lin.slot(sim))

    def __add__(self, o):
        return E2(self.re + o.re, self.im + o.im)

    def __sub__(self, o):
        return E2(self.re - o.re, self.im - o.im)

    def __neg__(self):
        return E2(-self.re, -self.im)

    def scale(self, c):
        return E2(self.re.scale(c), self.im.scale(c))

    def conj(self):
        return E2(self.re, -self.im)

    def mul_xi(self, xi0):
        """times xi = xi0 + i"""
        return E2(self.re.scale(xi0) - self.im, self.re + self.im.scale(xi0))

    def mul_i(self):
        return E2(-self.im, self.re)


class Acc2:
    """an Fp2 output under construction: Fp-level terms of its real and imaginary parts"""

    def __init__(self):
        self.re, self.im = [], []

    def prod(self, a, b):
        """+= a * b (Fp2 operands)"""
        for lst, x, y in ((self.re, a.re, b.re), (self.re, -a.im, b.im), (self.im, a.re, b.im), (self.im, a.im, b.re)):
            if not x.is_zero() and not y.is_zero():
                lst.append(("p", x, y))
        return self

    def sqr(self, a):
        """+= a^2 with the (re + im)(re - im), 2 re im forms (one product each)"""
        s, d = a.re + a.im, a.re - a.im
        if len(s.d) <= 2 and len(d.d) <= 2:
            self.re.append(("p", s, d))
            if not a.re.is_zero() and not a.im.is_zero():
                self.im.append(("p", a.re.scale(2), a.im))
            return self
        return self.prod(a, a)

    def prod_fp(self, a, b):
        """+= a * b with b a base-field operand (Lin)"""
        if not a.re.is_zero():
            self.re.append(("p", a.re, b))
        if not a.im.is_zero():
            self.im.append(("p", a.im, b))
        return self

    def prod_const(self, a, cre, cim):
        """+= a * (cre + cim i), constants given as (index, stored value) or None when zero"""
        for lst, x, c in ((self.re, a.re, cre), (self.re, -a.im, cim), (self.im, a.re, cim), (self.im, a.im, cre)):
            if c is not None and not x.is_zero():
                lst.append(("c", x, c))
        return self

    def prod_dconst(self, a, g):
        """+= a * G with G = (dre, dim) two DYNAMIC constants: entries of the constant area that an OP_GCLOAD record
        refills from the program's table (Prog.gconsts) -- the constants of a loop body that change from one pass to
        the next (the lines of a fixed point)"""
        dre, dim = g
        for lst, x, c in ((self.re, a.re, dre), (self.re, -a.im, dim), (self.im, a.re, dim), (self.im, a.im, dre)):
            if not x.is_zero():
                lst.append(("d", x, c))
        return self

    def dconst_fp(self, g, b):
        """+= G * b with b a base-field operand (Lin)"""
        self.re.append(("d", b, g[0]))
        self.im.append(("d", b, g[1]))
        return self

    def lin(self, a):
        if not a.re.is_zero():
            self.re.append(("l", a.re))
        if not a.im.is_zero():
            self.im.append(("l", a.im))
        return self


MASK_EXPO = 3  # store mask 3: the lane keeps the old value unless bit (base - repetition * stride) of ITS exponent is set


class Out:
    def __init__(self, dst, terms, scale=1, mask=0, raw=False, ebit=None):
        self.dst, self.terms, self.scale, self.mask, self.raw, self.ebit = dst, terms, scale, mask, raw, ebit
        assert (mask == MASK_EXPO) == (ebit is not None)


def outs2(dst_re, dst_im, acc, scale=1, mask=0, raw=False, ebit=None):
    return [Out(dst_re, acc.re, scale, mask, raw, ebit), Out(dst_im, acc.im, scale, mask, raw, ebit)]


# ------------------------------------------------------------------------------------------------ program
class Prog:
    def __init__(self, field, nslots, xi0, n_inputs, n_gslots):
        self.f, self.nslots, self.xi0 = field, nslots, xi0
        self.n_inputs, self.n_gslots = n_inputs, n_gslots
        self.ins = []      # instruction = list of WAVES records (dict)
        self.names = []
        self.consts = []   # stored values (ints in [0, p))
        self.gconsts = []  # the table of per-repetition constants (stored values), global memory
        self.dyn_base = None  # first of the constant-area entries that OP_GCLOAD refills (after the fixed ones)
        self.ndyn = GC_ENTRIES  # how many there are: one GCLOAD record refills GC_ENTRIES of them, from its "dst" on
        self.sched = []    # (start, len, repeat)
        self._open = None
        self.c_zero = self.const(0)
        self.c_one = self.const(field.R % field.p)  # Montgomery one
        self.c_plain_one = self.const(1)            # stored value 1: DOT by it leaves the plain residue

    # --- constants
    def const(self, stored):
        stored %= self.f.p
        if stored not in self.consts:
            self.consts.append(stored)
        return self.consts.index(stored)

    def mont(self, x):
        """constant holding the field element x (Montgomery form), as (index, stored) for Acc2.prod_const"""
        s = x % self.f.p * self.f.R % self.f.p
        return None if s == 0 else (self.const(s), s)

    def gtable(self, elements):
        """appends field elements (Montgomery form is applied here) to the table; returns the index of the first"""
        base = len(self.gconsts)
        self.gconsts += [x % self.f.p * self.f.R % self.f.p for x in elements]
        return base

    # --- schedule
    class _Rep:
        def __init__(self, prog, repeat):
            self.prog, self.repeat = prog, repeat

        def __enter__(self):
            assert self.prog._open is None
            self.prog._open = len(self.prog.ins)

        def __exit__(self, *a):
            start = self.prog._open
            self.prog._open = None
            if len(self.prog.ins) > start and self.repeat > 0:
                self.prog.sched.append((start, len(self.prog.ins) - start, self.repeat))

    def repeat(self, n):
        return Prog._Rep(self, n)

    def _emit(self, recs, name):
        assert len(recs) <= WAVES
        recs = list(recs) + [dict(op=OP_IDLE)] * (WAVES - len(recs))
        # barrier before the stores when some wave's destination is read by ANOTHER wave in this instruction
        reads = [self._reads(r) for r in recs]
        prebar = False
        for w, r in enumerate(recs):
            if "dst" in r and r["op"] not in (OP_GT_STORE, OP_IS_ONE, OP_SPILL, OP_CMP_EQ, OP_CMP_NZ2):
                for v, rd in enumerate(reads):
                    if v != w and r["dst"] in rd:
                        prebar = True
        dsts = [r["dst"] for r in recs if "dst" in r and r["op"] not in (OP_GT_STORE, OP_IS_ONE, OP_SPILL, OP_CMP_EQ, OP_CMP_NZ2)]
        assert len(dsts) == len(set(dsts)), "two waves write one slot: " + name
        for r in recs:
            r["prebar"] = prebar
        recs = self._balance(recs)
        self.ins.append(recs)
        self.names.append(name)
        if self._open is None:
            self.sched.append((len(self.ins) - 1, 1, 1))

    @staticmethod
    def _cost(r):
        """issue slots of a record, in products (one product = N^2 multiply-adds)"""
        if r["op"] != OP_DOT:
            return 0.3 if r["op"] != OP_IDLE else 0.0
        c = 0.0 if r["raw"] else 1.0  # the reduction
        for t in r["terms"]:
            c += 0.1 if t[0] == "l" else 1.05
        return c + 0.4

    @staticmethod
    def _balance(recs):
        """Wave w of a workgroup runs on SIMD w mod 4 (MI355X_MICROARCH.md: waves go to the SIMDs in a fixed cyclic
        order), three waves per SIMD: deal the records so that the four SIMDs carry equal work -- an Fp12 squaring has
        outputs of 7, 7, 6, 6, ... products, which in natural order load the SIMDs 21 : 18 : 21 : 18.  Spill / fill
        records address global scratch by wave index and stay where they are."""
        if any(r["op"] in (OP_SPILL, OP_FILL) for r in recs):
            return recs
        order = sorted(range(WAVES), key=lambda i: -Prog._cost(recs[i]))
        load, fill = [0.0] * 4, [[] for _ in range(4)]
        for i in order:
            b = min((b for b in range(4) if len(fill[b]) < 3), key=lambda b: load[b])
            fill[b].append(i)
            load[b] += Prog._cost(recs[i])
        out = [None] * WAVES
        for b in range(4):
            for k, i in enumerate(fill[b]):
                out[b + 4 * k] = recs[i]
        return out

    @staticmethod
    def _reads(r):
        s = set()
        if r["op"] == OP_DOT:
            for t in r["terms"]:
                for lin in t[1:3] if t[0] == "p" else t[1:2]:
                    s.update(lin.d.keys())
            if r.get("mask"):
                s.add(r["dst"])
        elif r["op"] in (OP_INV,):
            s.add(r["src"])
        elif r["op"] in (OP_GT_STORE, OP_IS_ONE, OP_SPILL):
            s.add(r["dst"])
        elif r["op"] in (OP_CMP_EQ, OP_CMP_NZ2):
            s.update((r["dst"], r["arg"]))
        return s

    def dot(self, outs, name="", extra=()):
        """one instruction of DOT records (+ `extra` non-DOT records for the waves left over)"""
        recs = []
        for o in outs:
            assert 0 <= o.dst < self.nslots
            for t in o.terms:
                if t[0] == "d":
                    assert 0 <= t[2] < self.ndyn
                for lin in (t[1:3] if t[0] == "p" else t[1:2]):
                    assert 1 <= len(lin.d) <= 2, (name, lin)
                    assert all(0 <= s < self.nslots and -128 <= c <= 127 for s, c in lin.d.items()), (name, lin)
            if o.raw:
                assert all(t[0] == "l" for t in o.terms)
            assert len(o.terms) <= 31 and 1 <= o.scale <= 15
            recs.append(dict(op=OP_DOT, dst=o.dst, terms=o.terms, scale=o.scale, mask=o.mask, raw=o.raw))
            if o.ebit is not None:
                assert 0 <= o.ebit[0] < 1 << 16 and 0 <= o.ebit[1] < 1 << 16
                recs[-1]["ebit"] = o.ebit
        self._emit(recs + list(extra), name)

    def misc(self, recs, name=""):
        self._emit(recs, name)

    # --- value-level simulation on stored residues
    @staticmethod
    def _kept(r, flags, expo, it):
        """does the lane keep the old value of the output slot?  (store masks of tower_vm.cuh)"""
        if not r["mask"]:
            return False
        if r["mask"] == MASK_EXPO:
            bit = r["ebit"][0] - it * r["ebit"][1]
            assert bit >= 0
            return not (expo >> bit) & 1
        return bool((flags >> (r["mask"] - 1)) & 1)

    def simulate(self, inputs, flags=0, expo=0):
        """inputs: list of stored residues (what the decode kernel leaves: a R1 mod p).  Returns (state, results) with
        results = {'gt': {offset: (value, is_c0)}, 'is_one': bool}.  expo: the lane's exponent (GT exponentiation)."""
        f = self.f
        p, Rinv = f.p, pow(f.R, -1, f.p)
        S = [0] * self.nslots
        G = {}
        CL = [None] * self.ndyn  # the dynamic constants; refilled at the END of the instruction that holds the GCLOAD
        res = {"gt": {}, "not_one": False, "flags": 0}

        def raise_flag(r):
            res["flags"] |= r.get("flag", FLAG_VERDICT)
            res["not_one"] = bool(res["flags"] & FLAG_VERDICT)

        for start, ln, rep in self.sched:
            for it in range(rep):
                for ins in self.ins[start:start + ln]:
                    new, newc = {}, {}
                    for w, r in enumerate(ins):
                        op = r["op"]
                        if op == OP_DOT:
                            acc = 0
                            for t in r["terms"]:
                                x = sum(c * S[s] for s, c in t[1].d.items())
                                if t[0] == "p":
                                    y = sum(c * S[s] for s, c in t[2].d.items())
                                    acc += x * y * (1 if r["raw"] else Rinv)
                                elif t[0] == "c":
                                    acc += x * t[2][1] * Rinv
                                elif t[0] == "d":
                                    acc += x * CL[t[2]] * Rinv
                                else:
                                    acc += x
                            v = acc * r["scale"] % p
                            if self._kept(r, flags, expo, it):
                                v = S[r["dst"]]
                            new[r["dst"]] = v
                        elif op == OP_GLOAD:
                            new[r["dst"]] = inputs[r["arg"]] % p
                        elif op == OP_GCLOAD:
                            base = r["arg"][0] + it * r["arg"][1]
                            newc[r.get("dst", 0)] = self.gconsts[base:base + GC_ENTRIES]
                        elif op == OP_CLOAD:
                            new[r["dst"]] = self.consts[r["arg"]]
                        elif op == OP_INV:
                            x = S[r["src"]]
                            new[r["dst"]] = pow(x, -1, p) * f.R1 * f.R1 % p if x else 0  # mont.cuh fp_inv: x^-1 R1^2
                        elif op == OP_SPILL:
                            G[(r["arg"], w)] = S[r["dst"]]
                        elif op == OP_FILL:
                            new[r["dst"]] = G[(r["arg"], w)]
                        elif op == OP_GT_STORE:
                            res["gt"][r["arg"] & 0xffff] = (S[r["dst"]], r["arg"] >> 16)
                        elif op == OP_IS_ONE:
                            if S[r["dst"]] != (1 if r["arg"] >> 16 else 0):
                                raise_flag(r)
                        elif op == OP_CMP_EQ:
                            if S[r["dst"]] != S[r["arg"]]:
                                raise_flag(r)
                        elif op == OP_CMP_NZ2:
                            if S[r["dst"]] == 0 and S[r["arg"]] == 0:
                                raise_flag(r)
                    for k, v in new.items():
                        S[k] = v
                    for d0, ent in newc.items():
                        assert len(ent) == GC_ENTRIES and d0 + GC_ENTRIES <= self.ndyn
                        CL[d0:d0 + GC_ENTRIES] = ent
        return S, res

    # --- the device's arithmetic, limb for limb
    def simulate_limbs(self, inputs, flags=0, expo=0):
        f = self.f
        N, p = f.N, f.p
        pl = f.balanced(p)
        lim63 = 1 << 63

        W = f.W

        def sext28(x):  # (sign extension from the field's limb width)
            x &= (1 << W) - 1
            return x - (1 << W) if x >> (W - 1) else x

        def normalise(cols):
            r, carry = [], 0
            for c in range(N - 1):
                v = cols[c] + carry
                assert abs(v) < lim63
                carry = (v + (1 << (W - 1))) >> W
                r.append(sext28(v))
            top = cols[N - 1] + carry
            assert abs(top) < 1 << 31, "top limb overflow"
            return r + [top]

        def operand(lin):
            v = [0] * N
            for s, c in lin.d.items():
                for i in range(N):
                    v[i] += c * S[s][i]
            assert all(abs(x) < 1 << 31 for x in v), "operand limb overflow"
            return v

        def canon(l):
            v = f.value(l)
            assert abs(v) < 4 * p, "canon_words needs |v| < 4p"
            return v % p

        def from_words(x):
            return normalise([(x >> (W * j)) & ((1 << W) - 1) for j in range(N)])

        S = [[0] * N for _ in range(self.nslots)]
        G = {}
        CL = [None] * self.ndyn
        res = {"gt": {}, "not_one": False, "flags": 0}

        def raise_flag(r):
            res["flags"] |= r.get("flag", FLAG_VERDICT)
            res["not_one"] = bool(res["flags"] & FLAG_VERDICT)

        for start, ln, rep in self.sched:
            for it in range(rep):
                for ins in self.ins[start:start + ln]:
                    new, newc = {}, {}
                    for w, r in enumerate(ins):
                        op = r["op"]
                        if op == OP_DOT:
                            t = [0] * (2 * N)
                            for term in r["terms"]:
                                x = operand(term[1])
                                if term[0] == "l":
                                    for i in range(N):
                                        t[N + i] += x[i]
                                else:
                                    if term[0] == "p":
                                        y = operand(term[2])
                                    elif term[0] == "d":
                                        y = f.balanced(CL[term[2]])
                                    else:
                                        y = f.balanced(term[2][1])
                                    for i in range(N):
                                        for j in range(N):
                                            t[i + j] += x[i] * y[j]
                                assert all(abs(c) < lim63 for c in t), "column overflow"
                            if not r["raw"]:
                                for i in range(N):
                                    m = sext28((t[i] & 0xffffffff) * f.ninv)
                                    for j in range(N):
                                        t[i + j] += m * pl[j]
                                    assert all(abs(c) < lim63 for c in t), "column overflow in the reduction"
                                    assert t[i] & ((1 << W) - 1) == 0
                                    t[i + 1] += t[i] >> W
                            v = normalise(t[N:])
                            if r["scale"] > 1:  # the scale is applied to the normalised limbs, then normalised again
                                v = normalise([c * r["scale"] for c in v])
                            if self._kept(r, flags, expo, it):
                                v = list(S[r["dst"]])
                            new[r["dst"]] = v
                        elif op == OP_GLOAD:
                            new[r["dst"]] = from_words(inputs[r["arg"]] % p)
                        elif op == OP_GCLOAD:
                            base = r["arg"][0] + it * r["arg"][1]
                            newc[r.get("dst", 0)] = self.gconsts[base:base + GC_ENTRIES]
                        elif op == OP_CLOAD:
                            new[r["dst"]] = f.balanced(self.consts[r["arg"]])
                        elif op == OP_INV:
                            x = canon(S[r["src"]])
                            new[r["dst"]] = from_words(pow(x, -1, p) * f.R1 * f.R1 % p if x else 0)
                        elif op == OP_SPILL:
                            G[(r["arg"], w)] = list(S[r["dst"]])
                        elif op == OP_FILL:
                            new[r["dst"]] = list(G[(r["arg"], w)])
                        elif op == OP_GT_STORE:
                            res["gt"][r["arg"] & 0xffff] = (canon(S[r["dst"]]), r["arg"] >> 16)
                        elif op == OP_IS_ONE:
                            if canon(S[r["dst"]]) != (1 if r["arg"] >> 16 else 0):
                                raise_flag(r)
                        elif op == OP_CMP_EQ:
                            if canon(S[r["dst"]]) != canon(S[r["arg"]]):
                                raise_flag(r)
                        elif op == OP_CMP_NZ2:
                            if canon(S[r["dst"]]) == 0 and canon(S[r["arg"]]) == 0:
                                raise_flag(r)
                    for k, v in new.items():
                        S[k] = v
                    for d0, ent in newc.items():
                        assert len(ent) == GC_ENTRIES and d0 + GC_ENTRIES <= self.ndyn
                        CL[d0:d0 + GC_ENTRIES] = ent
        return S, res

    # --- worst-case bounds for any input
    def check_bounds(self):
        """Walks the schedule with, per slot, a bound B on |value| / p (limbs are normalised on every store: |limb| <=
        2^27, top limb = value >> 364).  Checks operand limbs < 2^31, accumulator columns < 2^63, values < 2^8 p and
        |v| < 4p where the canonicaliser needs it.  Returns the largest column magnitude seen (as a power of two)."""
        import math

        f = self.f
        N, p, W = f.N, f.p, f.W
        lb = float(1 << (W - 1))
        pR = p / f.R
        B = [0.0] * self.nslots
        GB = {}
        worst_col, worst_val = 0.0, 0.0

        top_unit = p / 2.0 ** (W * (N - 1))  # top limb of a value v is v >> 28(N-1): at most |v| / that + 1

        def opnd(lin):
            limb = sum(abs(c) for c in lin.d.values()) * lb
            assert limb < 2.0 ** 31, ("operand limbs", lin)
            return limb, sum(abs(c) * B[s] for s, c in lin.d.items())

        def top(lin):
            return sum(abs(c) * (B[s] * top_unit + 1) for s, c in lin.d.items())

        def pair_coef(x, y):
            """sum over slot pairs of the magnitude of the coefficient of S[a]_i S[b]_j in a column of x * y: operands
            over the same slots combine -- (a + b)(a - b) has the antisymmetric cross terms cancel inside every
            column, leaving 2, not 4 (intermediate 64-bit sums may wrap; only the completed column has to fit)"""
            tot, seen = 0.0, set()
            for a, ca in x.d.items():
                for b, cb in y.d.items():
                    if (a, b) in seen:
                        continue
                    if a != b and b in x.d and a in y.d:
                        tot += abs(ca * cb + x.d[b] * y.d[a])  # a_i b_j and b_i a_j sum to the same column form
                        seen.add((b, a))
                    else:
                        tot += abs(ca * cb)
                    seen.add((a, b))
            return tot

        for start, ln, rep in self.sched:
            for _ in range(rep):
                for ii, ins in enumerate(self.ins[start:start + ln]):
                    new = {}
                    for w, r in enumerate(ins):
                        op = r["op"]
                        if op == OP_DOT:
                            col, val_prod, val_lin = 0.0, 0.0, 0.0
                            for t in r["terms"]:
                                lx, bx = opnd(t[1])
                                if t[0] == "l":
                                    col += lx
                                    val_lin += bx
                                else:
                                    if t[0] == "p":
                                        ly, by = opnd(t[2])
                                        # column N-2 has N-1 products of full limbs; column N-1 has N-2 of them and two
                                        # products with a top limb (v >> 28(N-1), small): the larger of the two
                                        full = pair_coef(t[1], t[2]) * lb * lb
                                        col += max((N - 1) * full, (N - 2) * full + lx * top(t[2]) + top(t[1]) * ly)
                                    else:
                                        ly, by = lb, 1.0
                                        col += N * lx * ly
                                    val_prod += bx * by
                            if not r["raw"]:
                                col += N * lb * lb + 2.0 ** 36
                                val = val_prod * pR + val_lin + 0.5 * (1 + 2.0 ** -26)
                            else:
                                val = val_lin
                            assert col < 2.0 ** 63, ("column bound", self.names[start + ii], math.log2(col))
                            assert r["scale"] * lb < 2.0 ** 31  # scaled normalised limbs stay below 2^31
                            worst_col = max(worst_col, col)
                            val *= r["scale"]
                            if r["mask"]:
                                val = max(val, B[r["dst"]])
                            assert val < 1024, ("value bound", self.names[start + ii], val)
                            worst_val = max(worst_val, val)
                            new[r["dst"]] = val
                        elif op in (OP_GLOAD, OP_CLOAD, OP_INV):
                            if op == OP_INV:
                                assert B[r["src"]] < 4, "INV input bound"
                            new[r["dst"]] = 1.0
                        elif op == OP_SPILL:
                            GB[(r["arg"], w)] = B[r["dst"]]
                        elif op == OP_FILL:
                            new[r["dst"]] = GB[(r["arg"], w)]
                        elif op in (OP_GT_STORE, OP_IS_ONE):
                            assert B[r["dst"]] < 4, "canonicaliser input bound"
                        elif op in (OP_CMP_EQ, OP_CMP_NZ2):
                            assert B[r["dst"]] < 4 and B[r["arg"]] < 4, ("canonicaliser input bound", self.names[start + ii])
                    for k, v in new.items():
                        B[k] = v
        return math.log2(worst_col), worst_val

    # --- encoding
    def encode(self):
        """(prog words, sched words) -- identical instructions are stored once"""
        blobs, index, order = [], {}, []
        for ins in self.ins:
            words = []
            for r in ins:
                rec = [0] * REC_WORDS
                op = r["op"]
                hdr = (op << 21) | (int(r["prebar"]) << 12)
                if op == OP_DOT:
                    hdr |= r["dst"] | (len(r["terms"]) << 6) | (int(r["raw"]) << 13) | ((r["scale"] if r["scale"] > 1 else 0) << 14) | (r["mask"] << 19)
                    for k, t in enumerate(r["terms"]):
                        def two(lin):
                            it = lin.items()
                            (s1, c1) = it[0]
                            (s2, c2) = it[1] if len(it) > 1 else (0, 0)
                            return s1, c1 & 0xff, s2, c2 & 0xff
                        x1, cx1, x2, cx2 = two(t[1])
                        if t[0] == "p":
                            y1, cy1, y2, cy2 = two(t[2])
                            w0 = x1 | (x2 << 6) | (y1 << 12) | (y2 << 18) | (K_PROD << 24)
                        elif t[0] == "c":
                            ci = t[2][0]
                            cy1 = cy2 = 0
                            w0 = x1 | (x2 << 6) | ((ci & 0xfff) << 12) | (K_PROD_CONST << 24)
                        elif t[0] == "d":  # a dynamic constant is an entry of the constant area like any other
                            cy1 = cy2 = 0
                            w0 = x1 | (x2 << 6) | (((self.dyn_base + t[2]) & 0xfff) << 12) | (K_PROD_CONST << 24)
                        else:
                            cy1 = cy2 = 0
                            w0 = x1 | (x2 << 6) | (K_LIN << 24)
                        rec[1 + 2 * k] = w0
                        rec[2 + 2 * k] = cx1 | (cx2 << 8) | (cy1 << 16) | (cy2 << 24)
                    if r["mask"] == MASK_EXPO:  # last word (terms end at 62): exponent bit of repetition 0 | stride << 16
                        rec[REC_WORDS - 1] = r["ebit"][0] | (r["ebit"][1] << 16)
                elif op == OP_IDLE:
                    pass
                elif op == OP_GCLOAD:  # word 1: table index of the first entry | its advance per repetition << 16
                    hdr |= self.dyn_base + r.get("dst", 0)
                    assert r["arg"][0] < 1 << 16 and r["arg"][1] < 1 << 16
                    rec[1] = r["arg"][0] | (r["arg"][1] << 16)
                else:
                    hdr |= r["dst"]
                    rec[1] = r["src"] if op == OP_INV else r["arg"]
                    if op in (OP_IS_ONE, OP_CMP_EQ, OP_CMP_NZ2):
                        rec[2] = r.get("flag", FLAG_VERDICT)  # the result-flag bits a failed comparison raises
                rec[0] = hdr
                words.extend(rec)
            key = tuple(words)
            if key not in index:
                index[key] = len(blobs)
                blobs.append(words)
            order.append(index[key])
        # schedule over the de-duplicated store: a block must be contiguous there, so blocks are re-laid in order
        prog, sched, placed = [], [], {}
        for start, ln, rep in self.sched:
            key = tuple(order[start:start + ln])
            if key not in placed:
                placed[key] = len(prog) // (WAVES * REC_WORDS)
                for b in key:
                    prog.extend(blobs[b])
            sched.append((placed[key], ln, rep))
        merged = []
        has_g = [any(r["op"] == OP_GCLOAD or r.get("mask") == MASK_EXPO for r in ins) for ins in self.ins]
        gblock = [any(has_g[start:start + ln]) for start, ln, rep in self.sched]
        for k, s in enumerate(sched):  # consecutive repeats of one block (a table index restarts with its block)
            if merged and merged[-1][0] == s[0] and merged[-1][1] == s[1] and not gblock[k]:
                merged[-1] = (s[0], s[1], merged[-1][2] + s[2])
            else:
                merged.append(s)
        return prog, merged

    def mads(self):
        """64-bit integer multiply-adds (v_mad_i64_i32) one pairing costs on the machine, summed over the twelve waves:
        N^2 per product term, N^2 per Montgomery reduction (the algorithmic count the roofline is priced with)."""
        n2 = self.f.N * self.f.N
        total = 0
        for start, ln, rep in self.sched:
            for ins in self.ins[start:start + ln]:
                for r in ins:
                    if r["op"] == OP_DOT:
                        total += rep * n2 * (sum(1 for t in r["terms"] if t[0] != "l") + (0 if r["raw"] else 1))
        return total

    def stats(self):
        n_exec = sum(ln * rep for _, ln, rep in self.sched)
        prods = 0
        for start, ln, rep in self.sched:
            for ins in self.ins[start:start + ln]:
                mx = max((sum(1 for t in r["terms"] if t[0] != "l") for r in ins if r["op"] == OP_DOT), default=0)
                prods += mx * rep
        return dict(instructions=len(self.ins), executed=n_exec, product_slots_per_wave=prods)


# ------------------------------------------------------------------------------------------------ tower helpers
class Tower:
    """builds instructions on a Prog for a tower with xi = xi0 + i"""

    def __init__(self, prog):
        self.P, self.xi0 = prog, prog.xi0

    @staticmethod
    def reg(base):
        """Fp12 register set at slots base .. base+11 as six E2"""
        return [E2.slots(base + 2 * j, base + 2 * j + 1) for j in range(6)]

    @staticmethod
    def conj12(a):
        return [a[j] if j % 2 == 0 else -a[j] for j in range(6)]

    def _wrap(self, a, i, k):
        """coefficient a_i contributing to w^k: a_i when no wrap, xi a_i after w^6 = xi"""
        return a[i]

    def mul12(self, dst, a, b, name="mul12", mask=0, ebit=None):
        outs = []
        for k in range(6):
            acc = Acc2()
            for j in range(6):
                i = k - j
                if i >= 0:
                    acc.prod(a[i], b[j])
                else:
                    acc.prod(a[i + 6].mul_xi(self.xi0), b[j])
            outs += outs2(dst + 2 * k, dst + 2 * k + 1, acc, mask=mask, ebit=ebit)
        self.P.dot(outs, name)

    def sqr12(self, dst, a, name="sqr12"):
        outs = []
        for k in range(6):
            acc = Acc2()
            for i in range(6):
                for j in range(i, 6):
                    if (i + j) % 6 != k:
                        continue
                    wrap = i + j >= 6
                    if i == j:
                        if not wrap:
                            acc.sqr(a[i])
                        else:  # xi a^2 = (u - v) + (u + v) i for xi0 = 1; in general (xi0 u - v) + (u + xi0 v) i
                            x = a[i]
                            s, d = x.re + x.im, x.re - x.im
                            acc.re.append(("p", s.scale(self.xi0), d))
                            acc.re.append(("p", x.re.scale(-2), x.im))
                            acc.im.append(("p", s, d))
                            acc.im.append(("p", x.re.scale(2 * self.xi0), x.im))
                    else:
                        ai = a[i].scale(2)
                        acc.prod(ai.mul_xi(self.xi0) if wrap else ai, a[j])
            outs += outs2(dst + 2 * k, dst + 2 * k + 1, acc)
        self.P.dot(outs, name)

    def mul_sparse(self, dst, a, ls, mask=0, name="mul_sparse"):
        """a * sum_j ls[j] w^j for a line with three non-zero coefficients (M-type twist: w^0, w^2, w^3; D-type:
        w^0, w^1, w^3)"""
        outs = []
        for k in range(6):
            acc = Acc2()
            for j, l in ls.items():
                i = k - j
                x = a[i] if i >= 0 else a[i + 6].mul_xi(self.xi0)
                if l == 1:  # the coefficient one: a linear term
                    acc.lin(x)
                elif isinstance(l, tuple):  # a coefficient held in two dynamic constants (re, im)
                    acc.prod_dconst(x, l)
                else:
                    acc.prod(x, l)
            outs += outs2(dst + 2 * k, dst + 2 * k + 1, acc, mask=mask)
        self.P.dot(outs, name)

    def cyclo_sqr(self, dst, a, name="cyclo_sqr", refresh=False):
        """Granger-Scott squaring of an element of the cyclotomic subgroup: three Fp4 squarings
        (t0, t1) = (x^2 + xi y^2, 2 x y), then z' = 3 t +- 2 z coefficient-wise.  The +-2 z enter as linear terms
        (unreduced: the value bound doubles per squaring); with `refresh` they are products by the constant one, which
        brings the bound back near p -- cyclo_sqr_run makes every third squaring of a run such a one (the products of a
        squaring contribute ~B^2 / 84 to the bound, so a longer period does not converge)."""
        xi0 = self.xi0

        def fp4(x, y):
            t0, t1 = Acc2(), Acc2()
            # 3 t0 = 3 x^2 + 3 xi y^2; the integer factors are split between the two operands of a product so that
            # no operand limb exceeds 2^31 (xi0 = 3 for bn256)
            sx, dx = x.re + x.im, x.re - x.im
            t0.re.append(("p", sx.scale(3), dx))
            t0.im.append(("p", x.re.scale(6), x.im))
            sy, dy = y.re + y.im, y.re - y.im
            t0.re.append(("p", sy.scale(xi0), dy.scale(3)))
            t0.re.append(("p", y.re.scale(-6), y.im))
            t0.im.append(("p", sy.scale(3), dy))
            t0.im.append(("p", y.re.scale(6), y.im.scale(xi0)))
            # 3 t1 = 6 x y
            t1.prod(x.scale(3), y.scale(2))
            return t0, t1

        # tower positions: c0 = (a0, a2, a4), c1 = (a1, a3, a5) in the w-basis
        c00, c01, c02, c10, c11, c12 = a[0], a[2], a[4], a[1], a[3], a[5]
        t0, t1 = fp4(c00, c11)
        t2, t3 = fp4(c10, c02)
        t4, t5 = fp4(c01, c12)
        # 3 xi t5: multiply the operands of t5's product by xi
        t5x = Acc2().prod(c01.mul_xi(xi0).scale(2), c12.scale(3))
        res = {0: (t0, c00, -2), 3: (t1, c11, 2), 1: (t5x, c10, 2), 4: (t4, c02, -2), 2: (t2, c01, -2), 5: (t3, c12, 2)}
        outs = []
        one = (self.P.c_one, self.P.consts[self.P.c_one])
        for k in range(6):
            acc, z, c = res[k]
            if refresh:
                acc.prod_const(z.scale(c), one, None)
            else:
                acc.lin(z.scale(c))
            outs += outs2(dst + 2 * k, dst + 2 * k + 1, acc)
        self.P.dot(outs, name + ("/refresh" if refresh else ""))

    def cyclo_sqr_run(self, dst, a, run, name):
        """`run` squarings in place: blocks of [refresh, linear, linear]"""
        q, s = divmod(run, 3)
        if q:
            with self.P.repeat(q):
                for i in range(3):
                    self.cyclo_sqr(dst, a, name, refresh=(i == 0))
        if s:
            self.cyclo_sqr(dst, a, name, refresh=True)
            if s > 1:
                with self.P.repeat(s - 1):
                    self.cyclo_sqr(dst, a, name)

    def frob12(self, dst, a, K, gammas, name="frob"):
        """a^(p^K): coefficient j -> conj^K(a_j) * gamma_{K,j}; gammas[j] = (re, im) field elements"""
        outs = []
        for j in range(6):
            x = a[j].conj() if K & 1 else a[j]
            g = gammas[j]
            acc = Acc2().prod_const(x, self.P.mont(g[0]), self.P.mont(g[1]))
            outs += outs2(dst + 2 * j, dst + 2 * j + 1, acc)
        self.P.dot(outs, name)

    def copy12(self, dst, a, name="copy12"):
        outs = []
        for j in range(6):
            outs += outs2(dst + 2 * j, dst + 2 * j + 1, Acc2().lin(a[j]), raw=True)
        self.P.dot(outs, name)

    def spill12(self, base, g, name="spill"):
        self.P.misc([dict(op=OP_SPILL, dst=base + w, arg=g) for w in range(12)], name)

    def fill12(self, base, g, name="fill"):
        self.P.misc([dict(op=OP_FILL, dst=base + w, arg=g) for w in range(12)], name)

    def pow_cyclo(self, acc_base, base_val, exponent, name="pow", cube_base=None):
        """S[acc_base..] = base_val^exponent for base_val in the cyclotomic subgroup, squarings in runs; base_val must
        not live in the acc register set.  Plain square-and-multiply from the top bit, or -- when a register set
        `cube_base` is free to hold base_val^3 -- width-3 NAF digits {+-1, +-3} (inversion is conjugation here, free):
        the result is the same power whatever the chain."""
        acc = self.reg(acc_base)
        if cube_base is None:
            digits = [int(b) for b in bin(exponent)[2:]]
            table = {1: base_val}
        else:
            k, naf = exponent, []
            while k:
                if k & 1:
                    d = k % 8
                    if d > 4:
                        d -= 8
                    if d in (3, -3, 1, -1):
                        pass
                    naf.append(d)
                    k -= d
                else:
                    naf.append(0)
                k >>= 1
            assert sum(d << i for i, d in enumerate(naf)) == exponent and all(d in (0, 1, -1, 3, -3) for d in naf)
            digits = naf[::-1]
            cube = self.reg(cube_base)
            self.cyclo_sqr(cube_base, base_val, name + "/cube_sq", refresh=True)
            self.mul12(cube_base, cube, base_val, name + "/cube")
            table = {1: base_val, -1: self.conj12(base_val), 3: cube, -3: self.conj12(cube)}
        assert digits[0] in (1, 3)
        self.copy12(acc_base, table[digits[0]], name + "/init")
        run = 0
        for d in digits[1:]:
            run += 1
            if d:
                self.cyclo_sqr_run(acc_base, acc, run, name + "/sqr")
                run = 0
                self.mul12(acc_base, acc, table[d], name + "/mul")
        if run:
            self.cyclo_sqr_run(acc_base, acc, run, name + "/sqr")


# ------------------------------------------------------------------------------------------------ BLS12-381
def bls12381_field():
    p = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    return VmField("bls12381", p, 14, 12, 390)


BLS_X_ABS = 0xD201000000010000


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


def frob_gammas(p, xi, K):
    return [_f2_pow(xi, j * (p ** K - 1) // 6, p) for j in range(6)]


def bls_psi(p):
    """psi(x, y) = (cx conj(x), cy conj(y)) on the M-type twist: cx = xi^-((p-1)/3), cy = xi^-((p-1)/2)"""
    def inv(a):
        n = pow(a[0] * a[0] + a[1] * a[1], -1, p)
        return (a[0] * n % p, -a[1] * n % p)
    return inv(_f2_pow((1, 1), (p - 1) // 3, p)), inv(_f2_pow((1, 1), (p - 1) // 2, p))


def g2_member_check(P, TX, TY, TZ, Q, tmp, flag, cx, cy, minus_t, name="member"):
    """Raise `flag` unless (cx conj(xQ), cy conj(yQ)) equals -T (minus_t) or T, T = (X : Y : Z) homogeneous, and Z != 0:
    an endomorphism image of the G2 operand Q held against the point the Miller loop has just finished with.  Q: the
    four slots of (xQ, yQ); tmp: ten free slots."""
    X, Y, Z = E2.slots(*TX), E2.slots(*TY), E2.slots(*TZ)
    xQ, yQ = E2.slots(Q[0], Q[1]), E2.slots(Q[2], Q[3])
    zero = tmp[8]
    o = outs2(tmp[0], tmp[1], Acc2().prod_const(xQ.conj(), P.mont(cx[0]), P.mont(cx[1])))
    o += outs2(tmp[2], tmp[3], Acc2().prod_const(yQ.conj(), P.mont(cy[0]), P.mont(cy[1])))
    P.dot(o, name + "/image", extra=[dict(op=OP_CLOAD, dst=zero, arg=P.c_zero)])
    px, py = E2.slots(tmp[0], tmp[1]), E2.slots(tmp[2], tmp[3])
    mone = P.mont(1)
    o = outs2(tmp[4], tmp[5], Acc2().prod(px, Z).prod_const(-X, mone, None))                      # image_x Z - X
    o += outs2(tmp[6], tmp[7], Acc2().prod(py, Z).prod_const(Y if minus_t else -Y, mone, None))   # image_y Z +- Y
    o += outs2(tmp[0], tmp[1], Acc2().prod_const(Z, mone, None))                                  # Z, bound refreshed
    P.dot(o, name + "/diff")
    P.misc([dict(op=OP_CMP_EQ, dst=tmp[4 + i], arg=zero, flag=flag) for i in range(4)]
           + [dict(op=OP_CMP_NZ2, dst=tmp[0], arg=tmp[1], flag=flag)], name + "/verdict")


def bls_g2_member_check(P, TX, TY, TZ, Q, tmp, flag, name="member"):
    """UnmarshalBinary's r-torsion test of a G2 operand (kilic/g2.go FromCompressed -> InCorrectSubgroup), decided where
    it is free: the Miller loop leaves T = [|x|] Q, and Q has order r exactly when psi(Q) = [x] Q = -T (Scott, eprint
    2021/1130 -- the criterion of the per-lane decode, bls12381.cuh g2_in_subgroup).  With T = (X : Y : Z):
    psi_x Z = X, psi_y Z = -Y, Z != 0.  The loop's formulas send every exceptional step (T = +-Q, T = infinity: only
    possible when the order of Q divides a partial parameter +- 1, never for order r) to Z = 0 for good, so such points
    are caught by the last condition.  A failed condition raises `flag` in the lane's result flags."""
    cx, cy = bls_psi(P.f.p)
    g2_member_check(P, TX, TY, TZ, Q, tmp, flag, cx, cy, True, name)


# slot map shared by the BLS12-381 programs
F_, G_, H_ = 0, 12, 24          # three Fp12 register sets
SPARE = 36                      # 36 .. 44
NSLOTS = 45


def bls_load_inputs(P, f, slots, first_input):
    """slots[i] <- input first_input + i (decoded by the per-lane code: a R1 mod p) converted to a R"""
    k1 = P.const(f.R * f.R * pow(f.R1, -1, f.p))  # x k1 / R = a R1 (R^2 / R1) / R = a R
    for base in range(0, len(slots), 12):
        part = slots[base:base + 12]
        P.misc([dict(op=OP_GLOAD, dst=s, arg=first_input + base + i) for i, s in enumerate(part)], "gload")
        P.dot([Out(s, [("c", Lin.slot(s), (k1, P.consts[k1]))]) for s in part], "to_vm_form")


def bls_dbl_step(P, T, TX, TY, TZ, tmp, L, PX, PY, fset, mask, extra=()):
    """T <- 2T on the twist, f <- f * tangent(P): two product rounds + the sparse multiplication.
    tmp: 10 slots, L: 6 slots."""
    X, Y, Z = E2.slots(*TX), E2.slots(*TY), E2.slots(*TZ)
    XY, B, E, YZ, A3 = (E2.slots(tmp[2 * i], tmp[2 * i + 1]) for i in range(5))
    o = []
    o += outs2(tmp[0], tmp[1], Acc2().prod(X, Y))
    o += outs2(tmp[2], tmp[3], Acc2().sqr(Y))
    # E = 3 b' Z^2 with b' = 4 xi: 12 xi Z^2
    zz = Acc2()
    s, d = Z.re + Z.im, Z.re - Z.im
    zz.re += [("p", s, d), ("p", Z.re.scale(-2), Z.im)]
    zz.im += [("p", s, d), ("p", Z.re.scale(2), Z.im)]
    o += outs2(tmp[4], tmp[5], zz, scale=12)
    o += outs2(tmp[6], tmp[7], Acc2().prod(Y, Z))
    o += outs2(tmp[8], tmp[9], Acc2().sqr(X), scale=3)
    P.dot(o, "dbl/a", extra)  # ten records: two waves are free for a caller's bookkeeping records
    o = []
    o += outs2(TX[0], TX[1], Acc2().prod(XY.scale(2), B - E.scale(3)))                     # 4 X3 = 2 XY (B - 3E)
    o += outs2(TY[0], TY[1], Acc2().sqr(B).prod(E.scale(3), B.scale(2) - E))               # 4 Y3 = B^2 + 3E(2B - E)
    o += outs2(TZ[0], TZ[1], Acc2().prod(B, YZ), scale=8)                                   # 4 Z3 = 8 B YZ
    o += outs2(L[0], L[1], Acc2().lin(B - E), raw=True)                                     # l0 = B - E
    o += outs2(L[2], L[3], Acc2().prod_fp(-A3, Lin.slot(PX)))                               # l2 = -3 X^2 xP
    o += outs2(L[4], L[5], Acc2().prod_fp(YZ.scale(2), Lin.slot(PY)))                       # l3 = 2 Y Z yP
    P.dot(o, "dbl/b")
    f = Tower.reg(fset)
    T.mul_sparse(fset, f, {0: E2.slots(L[0], L[1]), 2: E2.slots(L[2], L[3]), 3: E2.slots(L[4], L[5])}, mask=mask, name="dbl/line")


def bls_add_step(P, T, TX, TY, TZ, Q, tmp, L, PX, PY, fset, mask, extra=()):
    """T <- T + Q (Q affine), f <- f * chord(P).  tmp: 6 slots besides Q's 4 (which are recycled), L: 6 slots."""
    X, Y, Z = E2.slots(*TX), E2.slots(*TY), E2.slots(*TZ)
    xQ, yQ = E2.slots(Q[0], Q[1]), E2.slots(Q[2], Q[3])
    TH, LA = E2.slots(tmp[0], tmp[1]), E2.slots(tmp[2], tmp[3])
    o = []
    o += outs2(tmp[0], tmp[1], Acc2().lin(Y).prod(-yQ, Z))       # theta = Y - yQ Z
    o += outs2(tmp[2], tmp[3], Acc2().lin(X).prod(-xQ, Z))       # lambda = X - xQ Z
    P.dot(o, "add/a", extra)
    C, D = E2.slots(Q[0], Q[1]), E2.slots(Q[2], Q[3])             # recycle Q's slots (it is last read here)
    o = []
    o += outs2(L[0], L[1], Acc2().prod(TH, xQ).prod(-LA, yQ))     # l0 = theta xQ - lambda yQ
    o += outs2(L[2], L[3], Acc2().prod_fp(-TH, Lin.slot(PX)))     # l2 = -theta xP
    o += outs2(L[4], L[5], Acc2().prod_fp(LA, Lin.slot(PY)))      # l3 = lambda yP
    o += outs2(Q[0], Q[1], Acc2().sqr(TH))                        # C = theta^2
    o += outs2(Q[2], Q[3], Acc2().sqr(LA))                        # D = lambda^2
    P.dot(o, "add/b")
    Ee, Ff, Gg = C, D, E2.slots(tmp[4], tmp[5])                   # E, F over C, D in place; G new
    o = []
    o += outs2(Q[0], Q[1], Acc2().prod(LA, D))                    # E = lambda D
    o += outs2(Q[2], Q[3], Acc2().prod(Z, C))                     # F = Z C
    o += outs2(tmp[4], tmp[5], Acc2().prod(X, D))                 # G = X D
    P.dot(o, "add/c")
    o = []
    # E + F is one operand (two slots): five Fp2 products in X3, Y3 instead of seven -- Y3's record is the longest of
    # the instruction, everything else waits for it
    o += outs2(TX[0], TX[1], Acc2().prod(LA, Ee + Ff).prod(LA.scale(-2), Gg))                      # X3 = lambda (E + F - 2G)
    o += outs2(TY[0], TY[1], Acc2().prod(TH.scale(3), Gg).prod(-TH, Ee + Ff).prod(-Ee, Y))          # Y3 = theta (3G - E - F) - E Y
    o += outs2(TZ[0], TZ[1], Acc2().prod(Z, Ee))                                                    # Z3 = Z E
    P.dot(o, "add/d")
    f = Tower.reg(fset)
    T.mul_sparse(fset, f, {0: E2.slots(L[0], L[1]), 2: E2.slots(L[2], L[3]), 3: E2.slots(L[4], L[5])}, mask=mask, name="add/line")


# --- the Miller loop of a FIXED point Q (the G2 generator of bls.Verify): T's walk does not depend on the input, so
# the three line coefficients of every step are constants.  Per step the table holds the factors of xP and yP in l2, l3
# divided by l0 (Fp2 each: four field elements), in the order the loop visits the steps (doubling, then the addition
# where the parameter has a one).  The walk follows bls_dbl_step / bls_add_step formula for formula, so the fixed loop
# multiplies the lines the general loop would, each scaled by an Fp2 constant (which the final exponentiation kills).
BLS_G2_GEN = ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
               0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
              (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
               0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))


def bls_fixed_line_table(p, Q):
    """[(c2, c3)] per Miller step (Fp2 pairs of ints): line = 1 + (c2 xP) w^2 + (c3 yP) w^3 -- the general loop's line
    l0 + (-3X^2 xP) w^2 + (2YZ yP) w^3 divided by l0 (an Fp2 factor dies in the final exponentiation), so that the
    sparse multiplication has one coefficient that costs nothing"""
    def mul(a, b): return _f2_mul(a, b, p)
    def add(a, b): return ((a[0] + b[0]) % p, (a[1] + b[1]) % p)
    def sub(a, b): return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)
    def sc(a, c): return (a[0] * c % p, a[1] * c % p)

    def unit(l0, c2, c3):
        n = pow(l0[0] * l0[0] + l0[1] * l0[1], -1, p)  # l0 = 0 would make Q a point of small order
        inv = (l0[0] * n % p, -l0[1] * n % p)
        return (mul(c2, inv), mul(c3, inv))

    xQ, yQ = Q
    assert sub(mul(yQ, yQ), mul(mul(xQ, xQ), xQ)) == (4, 4), "Q is not on the twist y^2 = x^3 + 4 xi"
    X, Y, Z = xQ, yQ, (1, 0)
    out = []
    for b in bin(BLS_X_ABS)[3:]:
        XY, B, YZ, A3 = mul(X, Y), mul(Y, Y), mul(Y, Z), sc(mul(X, X), 3)
        E = sc(mul(mul(Z, Z), (1, 1)), 12)  # 3 b' Z^2, b' = 4 xi
        out.append(unit(sub(B, E), sc(A3, -1), sc(YZ, 2)))
        X, Y, Z = mul(sc(XY, 2), sub(B, sc(E, 3))), add(mul(B, B), mul(sc(E, 3), sub(sc(B, 2), E))), sc(mul(B, YZ), 8)
        if b == "1":
            TH, LA = sub(Y, mul(yQ, Z)), sub(X, mul(xQ, Z))
            out.append(unit(sub(mul(TH, xQ), mul(LA, yQ)), sc(TH, -1), LA))
            C, D = mul(TH, TH), mul(LA, LA)
            Ee, Ff, Gg = mul(LA, D), mul(Z, C), mul(X, D)
            X, Y, Z = (mul(LA, sub(add(Ee, Ff), sc(Gg, 2))), sub(mul(TH, sub(sc(Gg, 3), add(Ee, Ff))), mul(Ee, Y)), mul(Z, Ee))
    return out


def bls_fixed_load(g, stride, dst=0):
    """the record that moves the step's (c2, c3) -- table entries g .. g + 3, advanced by `stride` per repetition of the
    enclosing block -- into the dynamic constants dst .. dst + 3; it goes into an EARLIER instruction of the same step"""
    return dict(op=OP_GCLOAD, arg=(g, stride), dst=dst)


def bls_fixed_lines(L, PX, PY, d0=0):
    """outputs (c2 xP, c3 yP) -> L[2..5] with (c2, c3) the dynamic constants d0 .. d0 + 3"""
    return (outs2(L[2], L[3], Acc2().dconst_fp((d0, d0 + 1), Lin.slot(PX)))
            + outs2(L[4], L[5], Acc2().dconst_fp((d0 + 2, d0 + 3), Lin.slot(PY))))


def bls_fixed_step(P, T, L, PX, PY, fset, mask, name):
    """f <- f * (1 + c2 xP w^2 + c3 yP w^3) with (c2, c3) the dynamic constants 0 .. 3"""
    P.dot(bls_fixed_lines(L, PX, PY), name + "/l")
    T.mul_sparse(fset, Tower.reg(fset), {0: 1, 2: E2.slots(L[2], L[3]), 3: E2.slots(L[4], L[5])}, mask=mask, name=name + "/line")


def tower_easy_part(P, T, f, gamma2, F_, G_, H_, SPARE):
    """F <- f^((p^6 - 1)(p^2 + 1)) for the symbolic Fp12 value f held in register set F (possibly with signs): the
    inverse through the tower (one base-field inversion), conj(f) f^-1, then times its p^2-Frobenius.  Uses the
    register sets G, H and six spare slots.  The result is in the cyclotomic subgroup."""
    xi0 = P.xi0
    FF = T.reg(F_)
    # ---- easy part: f^(p^6 - 1) = conj(f) / f
    c0, c1 = [f[0], f[2], f[4]], [f[1], f[3], f[5]]

    def f6_prod_terms(accs, x, y, sign=1, times_v=False):
        """accs[0..2] += sign * (x y) [* v] for Fp6 elements x, y (lists of three E2)"""
        for i in range(3):
            for j in range(3):
                k = i + j
                a = x[i].scale(sign)
                if k >= 3:
                    k -= 3
                    a = a.mul_xi(xi0)
                if times_v:
                    k += 1
                    if k == 3:
                        k = 0
                        a = a.mul_xi(xi0)
                accs[k].prod(a, y[j])

    t_s = [(H_ + 2 * i, H_ + 2 * i + 1) for i in range(3)]
    accs = [Acc2(), Acc2(), Acc2()]
    f6_prod_terms(accs, c0, c0)
    f6_prod_terms(accs, c1, c1, sign=-1, times_v=True)
    P.dot(sum((outs2(t_s[i][0], t_s[i][1], accs[i]) for i in range(3)), []), "inv/t")      # t = c0^2 - v c1^2
    t = [E2.slots(*s) for s in t_s]
    abc_s = [(H_ + 6 + 2 * i, H_ + 7 + 2 * i) for i in range(3)]
    A = Acc2().prod(t[0], t[0]).prod(-t[1].mul_xi(xi0), t[2])                               # t0^2 - xi t1 t2
    Bb = Acc2().prod(t[2].mul_xi(xi0), t[2]).prod(-t[0], t[1])                              # xi t2^2 - t0 t1
    Cc = Acc2().prod(t[1], t[1]).prod(-t[0], t[2])                                          # t1^2 - t0 t2
    P.dot(outs2(*abc_s[0], A) + outs2(*abc_s[1], Bb) + outs2(*abc_s[2], Cc), "inv/abc")
    Av, Bv, Cv = (E2.slots(*s) for s in abc_s)
    d_s = (SPARE, SPARE + 1)
    dacc = Acc2().prod(t[0], Av).prod(t[2].mul_xi(xi0), Bv).prod(t[1].mul_xi(xi0), Cv)      # d = t0 A + xi (t2 B + t1 C)
    P.dot(outs2(d_s[0], d_s[1], dacc), "inv/d")
    n_s, ni_s = SPARE + 2, SPARE + 3
    P.dot([Out(n_s, [("p", Lin.slot(d_s[0]), Lin.slot(d_s[0])), ("p", Lin.slot(d_s[1]), Lin.slot(d_s[1]))])], "inv/norm")
    P.misc([dict(op=OP_INV, dst=ni_s, src=n_s)], "inv/fp")
    fld = P.f
    k2 = P.const(pow(fld.R, 3, fld.p) * pow(fld.R1 * fld.R1, -1, fld.p))                    # (x^-1 R1^2) k2 / R = x^-1 R^2
    P.dot([Out(ni_s, [("c", Lin.slot(ni_s), (k2, P.consts[k2]))])], "inv/fix")
    di_s = (SPARE + 4, SPARE + 5)
    P.dot([Out(di_s[0], [("p", Lin.slot(d_s[0]), Lin.slot(ni_s))]), Out(di_s[1], [("p", Lin.slot(d_s[1]).scale(-1), Lin.slot(ni_s))])],
          "inv/dinv")                                                                       # conj(d) / N
    di = E2.slots(*di_s)
    ti_s = [(H_ + 2 * i, H_ + 2 * i + 1) for i in range(3)]                                  # over t (dead after inv/d)
    P.dot(outs2(*ti_s[0], Acc2().prod(Av, di)) + outs2(*ti_s[1], Acc2().prod(Bv, di)) + outs2(*ti_s[2], Acc2().prod(Cv, di)), "inv/tinv")
    ti = [E2.slots(*s) for s in ti_s]
    # f^-1 = (c0 tinv) - (c1 tinv) w  -> register set G (w-basis interleave)
    a0 = [Acc2(), Acc2(), Acc2()]
    a1 = [Acc2(), Acc2(), Acc2()]
    f6_prod_terms(a0, c0, ti)
    f6_prod_terms(a1, c1, ti, sign=-1)
    o = []
    for m in range(3):
        o += outs2(G_ + 4 * m, G_ + 4 * m + 1, a0[m])
        o += outs2(G_ + 4 * m + 2, G_ + 4 * m + 3, a1[m])
    P.dot(o, "inv/finv")
    GG, HH = T.reg(G_), T.reg(H_)
    T.mul12(F_, T.conj12(f), GG, "easy/conj_times_inv")                                     # g1 = conj(f) f^-1  (in F)
    T.frob12(G_, FF, 2, gamma2, "easy/frob2")
    T.mul12(F_, GG, FF, "easy/mul")                                                         # g = g1^(p^2) g1: cyclotomic
    return FF


def bls_final_exp(P, T, f, gammas):
    """F <- f^(3 (p^12 - 1) / r) for the symbolic Fp12 value f held in register set F (possibly with signs).
    Uses G, H, the spare slots and four global spill slots.  Returns the symbolic result (in F)."""
    FF = tower_easy_part(P, T, f, gammas[2], F_, G_, H_, SPARE)
    GG, HH = T.reg(G_), T.reg(H_)
    # ---- hard part, exponent 3 (p^4 - p^2 + 1) / r: the five-exponentiation chain of the reference's kilic backend
    # (github.com/kilic/bls12-381 v0.1.0 finalExp, the zkcrypto chain; restated from memory and confirmed numerically to
    # be exactly the cube of the canonical reduced pairing -- tests/test_oracle_bls12381.py).  With g the easy-part
    # output and a^x = conj(a^|x|):
    #   B = g^x, D = conj(g^2) B, E = D^x, Fv = E^x, Hv = Fv^x B^2, I = Hv^x
    #   result = frob2(B Fv) * frob3(E g) * frob1(Hv conj(g)) * (I conj(D) g)
    # Three register sets and four global slots; `X~` below: the slots hold conj(X).
    X = BLS_X_ABS
    T.pow_cyclo(G_, FF, X, "hard/B")                       # G = B~
    T.spill12(F_, 0)                                       # gs0 = g
    T.cyclo_sqr(H_, FF, "hard/g2", refresh=True)           # H = g^2
    T.mul12(H_, HH, GG, "hard/D")                          # H = g^2 B~ = D~
    T.spill12(G_, 1)                                       # gs1 = B~
    T.spill12(H_, 2)                                       # gs2 = D~
    T.pow_cyclo(G_, T.conj12(HH), X, "hard/E")             # G = D^|x| = E~
    T.mul12(H_, T.conj12(GG), FF, "hard/Eg")               # H = E g
    T.frob12(H_, HH, 3, gammas[3], "hard/frob3")           # H = X2
    T.spill12(H_, 3)                                       # gs3 = R = X2
    T.pow_cyclo(F_, T.conj12(GG), X, "hard/Fv")            # F = E^|x| = Fv~
    T.fill12(H_, 1)                                        # H = B~
    T.mul12(H_, HH, FF, "hard/BFv")                        # H = (B Fv)~
    T.frob12(H_, T.conj12(HH), 2, gammas[2], "hard/frob2") # H = X1
    T.fill12(G_, 3)
    T.mul12(G_, GG, HH, "hard/R1")                         # R = X2 X1
    T.spill12(G_, 3)
    T.pow_cyclo(G_, T.conj12(FF), X, "hard/Hv0")           # G = Fv^|x| = (Fv^x)~
    T.fill12(H_, 1)                                        # H = B~
    T.cyclo_sqr(H_, HH, "hard/B2", refresh=True)           # H = (B^2)~
    T.mul12(G_, GG, HH, "hard/Hv")                         # G = Hv~
    T.fill12(F_, 0)                                        # F = g
    T.mul12(H_, GG, FF, "hard/Hvg")                        # H = Hv~ g = (Hv conj(g))~
    T.frob12(H_, T.conj12(HH), 1, gammas[1], "hard/frob1") # H = X3
    T.spill12(H_, 1)                                       # gs1 = X3
    T.pow_cyclo(H_, T.conj12(GG), X, "hard/I")             # H = Hv^|x| = I~
    T.fill12(G_, 2)                                        # G = D~ = conj(D)
    T.mul12(H_, T.conj12(HH), GG, "hard/ID")               # H = I conj(D)
    T.mul12(H_, HH, FF, "hard/X4")                         # H = I conj(D) g
    T.fill12(F_, 3)                                        # F = R
    T.fill12(G_, 1)                                        # G = X3
    T.mul12(F_, FF, GG, "hard/R2")
    T.mul12(F_, FF, HH, "hard/R3")
    return FF


def gt_layout_bls(j, c):
    """byte offset of coefficient (w^j, c = 0 real / 1 imaginary) in the 576-byte GT encoding (oracle gt_to_bytes:
    Fp12.c1 then c0; within Fp6 c2, c1, c0; within Fp2 c1, c0)"""
    h, m = j & 1, j >> 1
    return ((1 - h) * 3 + (2 - m)) * 96 + (0 if c == 1 else 48)


def bls_sched_miller(P, step, add):
    bits = bin(BLS_X_ABS)[3:]
    run = 0
    for b in bits:
        run += 1
        if b == "1":
            with P.repeat(run):
                step()
            run = 0
            add()
    if run:
        with P.repeat(run):
            step()


def build_bls12381_pair():
    f = bls12381_field()
    P = Prog(f, NSLOTS, 1, n_inputs=6, n_gslots=4)
    T = Tower(P)
    xi = (1, 1)
    gam = {K: frob_gammas(f.p, xi, K) for K in (1, 2, 3)}
    TX, TY, TZ = (12, 13), (14, 15), (16, 17)
    tmp = list(range(18, 28))
    L = list(range(28, 34))
    PX, PY = 34, 35
    Q = [36, 37, 38, 39]
    QT = [40, 41, 42, 43]  # the add step recycles its copy of Q
    bls_load_inputs(P, f, [PX, PY] + Q, 0)
    # f = 1, T = (xQ, yQ, 1)
    P.misc([dict(op=OP_CLOAD, dst=F_ + i, arg=P.c_one if i == 0 else P.c_zero) for i in range(12)], "f=1")
    o = [Out(TX[0], [("l", Lin.slot(Q[0]))], raw=True), Out(TX[1], [("l", Lin.slot(Q[1]))], raw=True),
         Out(TY[0], [("l", Lin.slot(Q[2]))], raw=True), Out(TY[1], [("l", Lin.slot(Q[3]))], raw=True)]
    P.misc([dict(op=OP_DOT, dst=x.dst, terms=x.terms, scale=1, mask=0, raw=True) for x in o]
           + [dict(op=OP_CLOAD, dst=TZ[0], arg=P.c_one), dict(op=OP_CLOAD, dst=TZ[1], arg=P.c_zero)], "T=Q")
    FF = T.reg(F_)

    def step():
        T.sqr12(F_, FF, "miller/sqr")
        bls_dbl_step(P, T, TX, TY, TZ, tmp, L, PX, PY, F_, 0)

    def add():
        P.dot([Out(QT[i], [("l", Lin.slot(Q[i]))], raw=True) for i in range(4)], "add/copyQ")
        bls_add_step(P, T, TX, TY, TZ, QT, tmp, L, PX, PY, F_, 0)

    bls_sched_miller(P, step, add)
    bls_g2_member_check(P, TX, TY, TZ, Q, tmp, FLAG_G2_A)
    res = bls_final_exp(P, T, T.conj12(FF), gam)
    # plain residues, then canonical bytes
    one = (P.c_plain_one, 1)
    P.dot(sum((outs2(F_ + 2 * j, F_ + 2 * j + 1, Acc2().prod_const(res[j], one, None)) for j in range(6)), []), "to_plain")
    P.misc([dict(op=OP_GT_STORE, dst=F_ + 2 * j + c, arg=gt_layout_bls(j, c) | ((1 if (j == 0 and c == 0) else 0) << 16))
            for j in range(6) for c in range(2)], "gt_store")
    return P


GT_EXP_BITS = 256  # GT exponentiation walks all 256 bits of the 32-byte scalar, most significant first


def _gt_store(P, base, layout):
    """register set `base` (Montgomery form, bound < 2^8 p) -> plain residues -> canonical bytes of the GT encoding"""
    T = Tower(P)
    X = T.reg(base)
    one = (P.c_plain_one, 1)
    P.dot(sum((outs2(base + 2 * j, base + 2 * j + 1, Acc2().prod_const(X[j], one, None)) for j in range(6)), []), "to_plain")
    P.misc([dict(op=OP_GT_STORE, dst=base + 2 * j + c, arg=layout(j, c) | ((1 if (j == 0 and c == 0) else 0) << 16))
            for j in range(6) for c in range(2)], "gt_store")


def build_bls12381_gtmul():
    """GTElt.Mul (kilic/gt.go:79-84: GT.Exp after the element was unmarshalled and found in the order-r subgroup) on the
    machine: inputs 0..11 = the twelve Fp coefficients (w-basis, re / im; decoded by the per-lane code: a R1 mod p), the
    lane's exponent enters through store mask 3.
      membership  f^r = 1, decided without an r-exponentiation: f conj(f) = 1 (f is non-zero and unitary), f^(p^4) f =
                  f^(p^2) (f^(p^4 - p^2 + 1) = 1: the cyclotomic subgroup, where inversion is conjugation and the
                  Granger-Scott squaring is valid) and conj(f)^p = f^|x| (f^(p - x) = 1, x < 0): the order of f divides
                  gcd(Phi_12(p), p - x) = r (the gcd is checked below).  A failed comparison sets the lane's flag; the
                  kernel stores zeros and status 2 for it.
      power       square-and-multiply from bit 255, cyclotomic squarings in [refresh, linear, linear] blocks like
                  cyclo_sqr_run, the multiplication by f masked by the exponent bit."""
    from math import gcd
    f = bls12381_field()
    p = f.p
    r_order = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    assert (p ** 4 - p ** 2 + 1) % r_order == 0 and gcd(p ** 4 - p ** 2 + 1, p + BLS_X_ABS) == r_order
    P = Prog(f, NSLOTS, 1, n_inputs=12, n_gslots=0)
    T = Tower(P)
    gam = {K: frob_gammas(p, (1, 1), K) for K in (1, 2, 4)}
    ONE, ZERO = SPARE, SPARE + 1
    bls_load_inputs(P, f, list(range(F_, F_ + 12)), 0)
    FF, GG, HH = T.reg(F_), T.reg(G_), T.reg(H_)
    P.misc([dict(op=OP_CLOAD, dst=ONE, arg=P.c_one), dict(op=OP_CLOAD, dst=ZERO, arg=P.c_zero)], "member/consts")

    def same(a_base, b_of, name):
        P.misc([dict(op=OP_CMP_EQ, dst=a_base + i, arg=b_of(i)) for i in range(12)], name)

    T.mul12(H_, T.conj12(FF), FF, "member/norm")
    same(H_, lambda i: ONE if i == 0 else ZERO, "member/norm==1")
    T.frob12(G_, FF, 2, gam[2], "member/frob2")
    T.frob12(H_, FF, 4, gam[4], "member/frob4")
    T.mul12(H_, HH, FF, "member/f^(p^4+1)")
    same(H_, lambda i: G_ + i, "member/cyclotomic")
    T.pow_cyclo(G_, FF, BLS_X_ABS, "member/pow")
    mone = P.mont(1)
    P.dot(sum((outs2(G_ + 2 * j, G_ + 2 * j + 1, Acc2().prod_const(GG[j], mone, None)) for j in range(6)), []), "member/refresh")
    T.frob12(H_, T.conj12(FF), 1, gam[1], "member/frob1")
    same(H_, lambda i: G_ + i, "member/f^p==f^x")
    # ---- the power
    P.misc([dict(op=OP_CLOAD, dst=G_ + i, arg=P.c_one if i == 0 else P.c_zero) for i in range(12)], "exp/acc=1")
    q, rem = divmod(GT_EXP_BITS, 3)
    with P.repeat(q):
        for j in range(3):
            T.cyclo_sqr(G_, GG, "exp/sqr", refresh=(j == 0))
            T.mul12(G_, GG, FF, "exp/mul", mask=MASK_EXPO, ebit=(GT_EXP_BITS - 1 - j, 3))
    for j in range(rem):
        T.cyclo_sqr(G_, GG, "exp/sqr", refresh=(j == 0))
        T.mul12(G_, GG, FF, "exp/mul", mask=MASK_EXPO, ebit=(rem - 1 - j, 0))
    _gt_store(P, G_, gt_layout_bls)
    return P


def build_bls12381_check():
    """inputs: P1 (2), Q1 (4), P2 (2), Q2 (4) with P2 already negated by the caller's decode kernel:
    ok = (f_{Q1}(P1) f_{Q2}(P2))^e == 1.  Flag bit 0 / 1: pair A / B has an operand at infinity (contributes 1)."""
    f = bls12381_field()
    P = Prog(f, NSLOTS, 1, n_inputs=12, n_gslots=4)
    T = Tower(P)
    xi = (1, 1)
    gam = {K: frob_gammas(f.p, xi, K) for K in (1, 2, 3)}
    T1 = [(12, 13), (14, 15), (16, 17)]
    T2 = [(18, 19), (20, 21), (22, 23)]
    tmp = list(range(24, 34))
    L = list(range(34, 40))
    P1, P2 = (40, 41), (42, 43)
    # Q1, Q2 converted once and parked in global slots 2 and 3 (waves 0..3), filled into tmp[0..3] at the add steps
    Qs = tmp[0:4]
    for g, first in ((2, 2), (3, 8)):
        bls_load_inputs(P, f, Qs, first)
        P.misc([dict(op=OP_SPILL, dst=Qs[i], arg=g) for i in range(4)], "park Q")
    bls_load_inputs(P, f, [P1[0], P1[1]], 0)
    bls_load_inputs(P, f, [P2[0], P2[1]], 6)
    P.misc([dict(op=OP_CLOAD, dst=F_ + i, arg=P.c_one if i == 0 else P.c_zero) for i in range(12)], "f=1")
    for TT, g in ((T1, 2), (T2, 3)):
        P.misc([dict(op=OP_FILL, dst=TT[0][0], arg=g), dict(op=OP_FILL, dst=TT[0][1], arg=g), dict(op=OP_FILL, dst=TT[1][0], arg=g),
                dict(op=OP_FILL, dst=TT[1][1], arg=g), dict(op=OP_CLOAD, dst=TT[2][0], arg=P.c_one),
                dict(op=OP_CLOAD, dst=TT[2][1], arg=P.c_zero)], "T=Q")
    FF = T.reg(F_)

    def step():
        T.sqr12(F_, FF, "miller/sqr")
        bls_dbl_step(P, T, T1[0], T1[1], T1[2], tmp, L, P1[0], P1[1], F_, 1)
        bls_dbl_step(P, T, T2[0], T2[1], T2[2], tmp, L, P2[0], P2[1], F_, 2)

    def add():
        for TT, g, PP, mask in ((T1, 2, P1, 1), (T2, 3, P2, 2)):
            P.misc([dict(op=OP_FILL, dst=Qs[i], arg=g) for i in range(4)], "add/fillQ")
            bls_add_step(P, T, TT[0], TT[1], TT[2], Qs, tmp[4:10], L, PP[0], PP[1], F_, mask)

    bls_sched_miller(P, step, add)
    for TT, g, flag in ((T1, 2, FLAG_G2_A), (T2, 3, FLAG_G2_B)):
        P.misc([dict(op=OP_FILL, dst=Qs[i], arg=g) for i in range(4)], "member/fillQ")
        bls_g2_member_check(P, TT[0], TT[1], TT[2], Qs, tmp[4:10] + L[0:4], flag)
    res = bls_final_exp(P, T, T.conj12(FF), gam)
    one = (P.c_plain_one, 1)
    P.dot(sum((outs2(F_ + 2 * j, F_ + 2 * j + 1, Acc2().prod_const(res[j], one, None)) for j in range(6)), []), "to_plain")
    P.misc([dict(op=OP_IS_ONE, dst=F_ + 2 * j + c, arg=(1 if (j == 0 and c == 0) else 0) << 16) for j in range(6) for c in range(2)],
           "is_one")
    return P


def build_bls12381_verify():
    """CHECK with the second pair's G2 operand fixed to the generator (bls.Verify with signatures on G1, sign/bls
    bls.go:83-96: e(H(m), pk) e(-sig, g2) == 1): inputs P1 (2), Q1 (4), P2 (2, negated by the decode kernel); the lines
    of the second Miller loop come from the table, its point is never computed."""
    f = bls12381_field()
    P = Prog(f, NSLOTS, 1, n_inputs=8, n_gslots=4)
    T = Tower(P)
    xi = (1, 1)
    gam = {K: frob_gammas(f.p, xi, K) for K in (1, 2, 3)}
    T1 = [(12, 13), (14, 15), (16, 17)]
    tmp = list(range(24, 34))
    L = list(range(34, 40))
    P1, P2 = (40, 41), (42, 43)
    Qs = tmp[0:4]
    bls_load_inputs(P, f, Qs, 2)
    P.misc([dict(op=OP_SPILL, dst=Qs[i], arg=2) for i in range(4)], "park Q")
    bls_load_inputs(P, f, [P1[0], P1[1]], 0)
    bls_load_inputs(P, f, [P2[0], P2[1]], 6)
    P.misc([dict(op=OP_CLOAD, dst=F_ + i, arg=P.c_one if i == 0 else P.c_zero) for i in range(12)], "f=1")
    P.misc([dict(op=OP_FILL, dst=T1[0][0], arg=2), dict(op=OP_FILL, dst=T1[0][1], arg=2), dict(op=OP_FILL, dst=T1[1][0], arg=2),
            dict(op=OP_FILL, dst=T1[1][1], arg=2), dict(op=OP_CLOAD, dst=T1[2][0], arg=P.c_one),
            dict(op=OP_CLOAD, dst=T1[2][1], arg=P.c_zero)], "T=Q")
    FF = T.reg(F_)
    lines = bls_fixed_line_table(f.p, BLS_G2_GEN)
    cur = [P.gtable([c for line in lines for z in line for c in z])]
    assert cur[0] == 0

    def step():
        T.sqr12(F_, FF, "miller/sqr")
        bls_dbl_step(P, T, T1[0], T1[1], T1[2], tmp, L, P1[0], P1[1], F_, 1, extra=[bls_fixed_load(cur[0], 4)])
        bls_fixed_step(P, T, L, P2[0], P2[1], F_, 2, "fixdbl")

    def add():
        P.misc([dict(op=OP_FILL, dst=Qs[i], arg=2) for i in range(4)], "add/fillQ")
        bls_add_step(P, T, T1[0], T1[1], T1[2], Qs, tmp[4:10], L, P1[0], P1[1], F_, 1, extra=[bls_fixed_load(cur[0], 0)])
        bls_fixed_step(P, T, L, P2[0], P2[1], F_, 2, "fixadd")
        cur[0] += 4

    def sched(step, add):  # bls_sched_miller with the table cursor moved past every run of doublings
        run = 0
        for b in bin(BLS_X_ABS)[3:]:
            run += 1
            if b == "1":
                with P.repeat(run):
                    step()
                cur[0] += 4 * run
                run = 0
                add()
        if run:
            with P.repeat(run):
                step()
            cur[0] += 4 * run

    sched(step, add)
    assert cur[0] == len(P.gconsts) and len(lines[0]) * 2 == GC_ENTRIES
    P.misc([dict(op=OP_FILL, dst=Qs[i], arg=2) for i in range(4)], "member/fillQ")
    bls_g2_member_check(P, T1[0], T1[1], T1[2], Qs, tmp[4:10] + L[0:4], FLAG_G2_A)
    res = bls_final_exp(P, T, T.conj12(FF), gam)
    one = (P.c_plain_one, 1)
    P.dot(sum((outs2(F_ + 2 * j, F_ + 2 * j + 1, Acc2().prod_const(res[j], one, None)) for j in range(6)), []), "to_plain")
    P.misc([dict(op=OP_IS_ONE, dst=F_ + 2 * j + c, arg=(1 if (j == 0 and c == 0) else 0) << 16) for j in range(6) for c in range(2)],
           "is_one")
    P.dyn_base = len(P.consts)  # the dynamic constants follow the fixed ones in the constant area
    return P


def build_bls12381_verify_same_key(key=None):
    """bls.Verify for MANY messages under ONE public key (sign/bls/bls.go:82-96 called in a loop with the same X: a
    drand chain, sign/tbls/tbls.go:100-107): e(H(m), X) e(-sig, g2) == 1 where BOTH G2 operands are the same for every
    lane -- so both Miller loops take their lines from tables: the generator's (constants of the program, as in VERIFY)
    and the key's (computed once per key by bls12381_key_lines_kernel with bls_fixed_line_table's formulas; the launch
    hands the machine one table [generator | key]).  No point is walked at all: a step is f^2 and two sparse
    multiplications.  inputs: P1 = H(m) (2), P2 = -sig (2).  `key`: the G2 point whose lines fill the key half of the
    EMITTED table (any valid point: the device overwrites that half; the simulators use it)."""
    f = bls12381_field()
    P = Prog(f, NSLOTS, 1, n_inputs=4, n_gslots=4)
    P.ndyn = 2 * GC_ENTRIES
    T = Tower(P)
    xi = (1, 1)
    gam = {K: frob_gammas(f.p, xi, K) for K in (1, 2, 3)}
    LK = list(range(24, 30))   # the key's line (pair A)
    LG = list(range(34, 40))   # the generator's line (pair B)
    P1, P2 = (40, 41), (42, 43)
    bls_load_inputs(P, f, [P1[0], P1[1]], 0)
    bls_load_inputs(P, f, [P2[0], P2[1]], 2)
    P.misc([dict(op=OP_CLOAD, dst=F_ + i, arg=P.c_one if i == 0 else P.c_zero) for i in range(12)], "f=1")
    FF = T.reg(F_)
    gen_lines = bls_fixed_line_table(f.p, BLS_G2_GEN)
    key_lines = bls_fixed_line_table(f.p, key if key is not None else BLS_G2_GEN)
    g0 = P.gtable([c for line in gen_lines for z in line for c in z])
    k0 = P.gtable([c for line in key_lines for z in line for c in z])
    assert g0 == 0 and k0 == 4 * len(gen_lines)
    P.key_table_base = k0
    cur = [0]

    def step(stride):
        # the two loads ride in an instruction of their own (two waves; the constants change at its END)
        P.misc([bls_fixed_load(k0 + cur[0], stride, 0), bls_fixed_load(g0 + cur[0], stride, GC_ENTRIES)], "samekey/load")
        P.dot(bls_fixed_lines(LK, P1[0], P1[1], 0) + bls_fixed_lines(LG, P2[0], P2[1], GC_ENTRIES), "samekey/l")

    def lines(name):
        T.mul_sparse(F_, FF, {0: 1, 2: E2.slots(LK[2], LK[3]), 3: E2.slots(LK[4], LK[5])}, mask=1, name=name + "/key")
        T.mul_sparse(F_, FF, {0: 1, 2: E2.slots(LG[2], LG[3]), 3: E2.slots(LG[4], LG[5])}, mask=2, name=name + "/gen")

    run = 0
    for b in bin(BLS_X_ABS)[3:]:
        run += 1
        if b == "1":
            with P.repeat(run):
                step(4)
                T.sqr12(F_, FF, "miller/sqr")
                lines("fixdbl")
            cur[0] += 4 * run
            run = 0
            step(0)
            lines("fixadd")
            cur[0] += 4
    if run:
        with P.repeat(run):
            step(4)
            T.sqr12(F_, FF, "miller/sqr")
            lines("fixdbl")
        cur[0] += 4 * run
    assert cur[0] == 4 * len(gen_lines) and len(P.gconsts) == 8 * len(gen_lines)
    res = bls_final_exp(P, T, T.conj12(FF), gam)
    one = (P.c_plain_one, 1)
    P.dot(sum((outs2(F_ + 2 * j, F_ + 2 * j + 1, Acc2().prod_const(res[j], one, None)) for j in range(6)), []), "to_plain")
    P.misc([dict(op=OP_IS_ONE, dst=F_ + 2 * j + c, arg=(1 if (j == 0 and c == 0) else 0) << 16) for j in range(6) for c in range(2)],
           "is_one")
    P.dyn_base = len(P.consts)  # the dynamic constants follow the fixed ones in the constant area
    return P


# ------------------------------------------------------------------------------------------------ BN curves
# pairing/bn256 (dclxvi parameters, xi = 3 + i) and pairing/bn254 (alt_bn128, xi = 9 + i; the same package with other
# constants): p = 36u^4 + 36u^3 + 24u^2 + 6u + 1, D-type twist y^2 = x^3 + 3/xi, optimal ate loop over the reference's
# signed digits of 6u + 2 with the two Frobenius steps (optate.go:126-213) and the final exponentiation's addition chain
# (optate.go:215-264) -- the GT bytes must equal the reference's, so the exponent is exactly the chain's; the line
# functions and the digit form are free (any Fp2 multiple of a line dies in the final exponentiation).
class BnCurve:
    def __init__(self, name, u, xi0, digits, w=W, strict_g2=False):
        self.name, self.u, self.xi0, self.digits, self.w = name, u, xi0, digits, w
        self.strict_g2 = strict_g2  # the package's UnmarshalBinary rejects G2 points outside the order-n subgroup
        self.p = 36 * u ** 4 + 36 * u ** 3 + 24 * u ** 2 + 6 * u + 1
        assert sum(d << i for i, d in enumerate(digits)) == 6 * u + 2

    def field(self):
        return VmField(self.name, self.p, 10, 8, 261, self.w)


BN256 = BnCurve("bn256", 6518589491078791937, 3,   # constants.go:17, optate.go:117-122
                [0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, -1, 0, 1, 0,
                 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, -1,
                 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, -1, 0, 0, 0,
                 0, 1, 0, 0, 0, 1])
BN254 = BnCurve("bn254", 4965661367192848881, 9,   # pairing/bn254/constants.go:17, optate.go:117-120
                [0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0,
                 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0, 1, 1,
                 1, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1,
                 1, 0, 0, -1, 0, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, 1, 1],
                w=27, strict_g2=True)  # xi0 = 9 puts sums of 10-fold coefficients into the unreduced columns: 27-bit limbs keep them below 2^63
BN_NSLOTS = 63
BN_A, BN_B, BN_C, BN_D, BN_E = 0, 12, 24, 36, 48   # five Fp12 register sets; slots 60..62 spare


def bn256_field():
    return BN256.field()


def bn_dbl_step(P, T, TX, TY, TZ, tmp, L, PX, PY, fset, mask):
    """T <- 2T on the D-type twist (b' = 3 / xi), f <- f * tangent(P).  With Bh = 10 Y^2 and Ep = 9 conj(xi) Z^2
    (10 = norm(xi): Bh / 10 = Y^2, Ep / 10 = 3 b' Z^2) the point, scaled by 100, is
      X3 = 10 [2 XY (Bh - 3 Ep)],  Y3 = Bh^2 + 3 Ep (2 Bh - Ep),  Z3 = 10 [8 Bh YZ]
    and the tangent, scaled by 10:  l0 = 10 [2 YZ yP],  l1 = 10 [-3 X^2 xP],  l3 = Bh - Ep."""
    xi0 = P.xi0
    X, Y, Z = E2.slots(*TX), E2.slots(*TY), E2.slots(*TZ)
    XY, Bh, Ep, YZ, A3 = (E2.slots(tmp[2 * i], tmp[2 * i + 1]) for i in range(5))
    if xi0 * xi0 + 1 > 15:
        return bn_dbl_step_xi(P, T, TX, TY, TZ, tmp, L, PX, PY, fset, mask)
    o = []
    o += outs2(tmp[0], tmp[1], Acc2().prod(X, Y))
    o += outs2(tmp[2], tmp[3], Acc2().sqr(Y), scale=xi0 * xi0 + 1)
    zz = Acc2()  # conj(xi) Z^2 = (xi0 u + v) + (xi0 v - u) i with Z^2 = u + v i
    s, d = Z.re + Z.im, Z.re - Z.im
    zz.re += [("p", s.scale(xi0), d), ("p", Z.re.scale(2), Z.im)]
    zz.im += [("p", Z.re.scale(2 * xi0), Z.im), ("p", -s, d)]
    o += outs2(tmp[4], tmp[5], zz, scale=9)
    o += outs2(tmp[6], tmp[7], Acc2().prod(Y, Z))
    o += outs2(tmp[8], tmp[9], Acc2().sqr(X), scale=3)
    P.dot(o, "dbl/a")
    o = []
    o += outs2(TX[0], TX[1], Acc2().prod(XY.scale(2), Bh - Ep.scale(3)), scale=10)
    o += outs2(TY[0], TY[1], Acc2().sqr(Bh).prod(Ep.scale(3), Bh.scale(2) - Ep))
    o += outs2(TZ[0], TZ[1], Acc2().prod(Bh.scale(8), YZ), scale=10)
    o += outs2(L[0], L[1], Acc2().prod_fp(YZ.scale(2), Lin.slot(PY)), scale=10)      # l0
    o += outs2(L[2], L[3], Acc2().prod_fp(-A3, Lin.slot(PX)), scale=10)             # l1
    o += outs2(L[4], L[5], Acc2().lin(Bh - Ep), raw=True)                           # l3
    P.dot(o, "dbl/b")
    f = Tower.reg(fset)
    T.mul_sparse(fset, f, {0: E2.slots(L[0], L[1]), 1: E2.slots(L[2], L[3]), 3: E2.slots(L[4], L[5])}, mask=mask, name="dbl/line")


def bn_dbl_step_xi(P, T, TX, TY, TZ, tmp, L, PX, PY, fset, mask):
    """The doubling step when norm(xi) = xi0^2 + 1 exceeds the post-scale range (bn254: 82).  3 b' = 9 / xi, so with
    Bt = xi Y^2 and Et = 9 Z^2 (B = Bt / xi, E = 3 b' Z^2 = Et / xi) the point, scaled by xi^2, is
      X3 = 2 (xi XY) (Bt - 3 Et),  Y3 = Bt^2 + 3 Et (2 Bt - Et),  Z3 = 8 Bt (xi YZ)
    and the tangent, scaled by xi:  l0 = 2 (xi YZ) yP,  l1 = -3 (xi X^2) xP,  l3 = Bt - Et.
    The multiplications by xi cost nothing: xi Y (xi X) is an OPERAND -- the slot combination (xi0 y0 - y1, y0 + xi0 y1) --
    of the products X (xi Y), Y (xi Y), (xi Y) Z, X (xi X)."""
    xi0 = P.xi0
    X, Y, Z = E2.slots(*TX), E2.slots(*TY), E2.slots(*TZ)
    XY, Bt, Et, YZ, A3 = (E2.slots(tmp[2 * i], tmp[2 * i + 1]) for i in range(5))
    xiY, xiX = Y.mul_xi(xi0), X.mul_xi(xi0)
    o = []
    o += outs2(tmp[0], tmp[1], Acc2().prod(X, xiY))
    o += outs2(tmp[2], tmp[3], Acc2().prod(Y, xiY))
    o += outs2(tmp[4], tmp[5], Acc2().sqr(Z), scale=9)
    o += outs2(tmp[6], tmp[7], Acc2().prod(xiY, Z))
    o += outs2(tmp[8], tmp[9], Acc2().prod(X, xiX), scale=3)
    P.dot(o, "dbl/a")
    o = []
    o += outs2(TX[0], TX[1], Acc2().prod(XY.scale(2), Bt - Et.scale(3)))
    o += outs2(TY[0], TY[1], Acc2().sqr(Bt).prod(Et.scale(3), Bt.scale(2) - Et))
    o += outs2(TZ[0], TZ[1], Acc2().prod(Bt.scale(8), YZ))
    o += outs2(L[0], L[1], Acc2().prod_fp(YZ.scale(2), Lin.slot(PY)))              # l0
    o += outs2(L[2], L[3], Acc2().prod_fp(-A3, Lin.slot(PX)))                     # l1
    o += outs2(L[4], L[5], Acc2().lin(Bt - Et), raw=True)                         # l3
    P.dot(o, "dbl/b")
    f = Tower.reg(fset)
    T.mul_sparse(fset, f, {0: E2.slots(L[0], L[1]), 1: E2.slots(L[2], L[3]), 3: E2.slots(L[4], L[5])}, mask=mask, name="dbl/line")


def bn_add_step(P, T, TX, TY, TZ, Q, sign, tmp, L, PX, PY, fset, mask):
    """T <- T + sign Q (Q = four slots, affine; they are recycled), f <- f * chord(P), D-type line placement:
    l0 = lambda yP, l1 = -theta xP, l3 = theta xQ - lambda yQ."""
    X, Y, Z = E2.slots(*TX), E2.slots(*TY), E2.slots(*TZ)
    xQ, yQ = E2.slots(Q[0], Q[1]), E2.slots(Q[2], Q[3]).scale(sign)
    TH, LA = E2.slots(tmp[0], tmp[1]), E2.slots(tmp[2], tmp[3])
    o = []
    o += outs2(tmp[0], tmp[1], Acc2().lin(Y).prod(-yQ, Z))
    o += outs2(tmp[2], tmp[3], Acc2().lin(X).prod(-xQ, Z))
    P.dot(o, "add/a")
    C, D = E2.slots(Q[0], Q[1]), E2.slots(Q[2], Q[3])
    o = []
    o += outs2(L[0], L[1], Acc2().prod_fp(LA, Lin.slot(PY)))
    o += outs2(L[2], L[3], Acc2().prod_fp(-TH, Lin.slot(PX)))
    o += outs2(L[4], L[5], Acc2().prod(TH, xQ).prod(-LA, yQ))
    o += outs2(Q[0], Q[1], Acc2().sqr(TH))
    o += outs2(Q[2], Q[3], Acc2().sqr(LA))
    P.dot(o, "add/b")
    Ee, Ff, Gg = C, D, E2.slots(tmp[4], tmp[5])
    o = []
    o += outs2(Q[0], Q[1], Acc2().prod(LA, D))
    o += outs2(Q[2], Q[3], Acc2().prod(Z, C))
    o += outs2(tmp[4], tmp[5], Acc2().prod(X, D))
    P.dot(o, "add/c")
    o = []
    o += outs2(TX[0], TX[1], Acc2().prod(LA, Ee + Ff).prod(LA.scale(-2), Gg))              # (E + F: one two-slot operand)
    o += outs2(TY[0], TY[1], Acc2().prod(TH.scale(3), Gg).prod(-TH, Ee + Ff).prod(-Ee, Y))
    o += outs2(TZ[0], TZ[1], Acc2().prod(Z, Ee))
    P.dot(o, "add/d")
    f = Tower.reg(fset)
    T.mul_sparse(fset, f, {0: E2.slots(L[0], L[1]), 1: E2.slots(L[2], L[3]), 3: E2.slots(L[4], L[5])}, mask=mask, name="add/line")


def bn_g2_member_check(P, curve, TX, TY, TZ, Q, tmp, flag, name="member"):
    """pairing/bn254's UnmarshalBinary rejects G2 points outside the order-n subgroup (twist.go:47-66).  The ate loop ends
    at T = [6u+2] Q + pi(Q) - pi^2(Q), and (6u+2) + pi - pi^2 + pi^3 = 0 on G2 (the optimal ate relation): Q is a member
    exactly when T = -pi^3(Q) = (conj(x) k1x k2x, conj(y) k1y) -- exact on alt_bn128 because the polynomial is non-zero at
    both eigenvalues of pi modulo each of the four primes of the cofactor (tests/test_constants.py), the twist group
    being cyclic.  Exceptional steps end in Z = 0 as on BLS12-381 (bls_g2_member_check)."""
    p = P.f.p
    xi = (curve.xi0, 1)
    k1x, k1y = _f2_pow(xi, (p - 1) // 3, p), _f2_pow(xi, (p - 1) // 2, p)
    k2x = _f2_pow(xi, (p * p - 1) // 3, p)
    g2_member_check(P, TX, TY, TZ, Q, tmp, flag, _f2_mul(k1x, k2x, p), k1y, False, name)


def bn_miller(P, T, f, first_input, mask, curve=None, member_flag=None):
    """F (set A) <- miller(Q, P) of optate.go:126-213 up to factors that the final exponentiation removes.
    Inputs first_input .. +5: P.x, P.y, Q.x.re, Q.x.im, Q.y.re, Q.y.im."""
    curve = curve or BN256
    p = f.p
    xi = (curve.xi0, 1)
    digits = curve.digits
    TX, TY, TZ = (12, 13), (14, 15), (16, 17)
    tmp = list(range(18, 28))
    L = list(range(28, 34))
    PX, PY = 34, 35
    Q = [36, 37, 38, 39]
    QT = [40, 41, 42, 43]
    bls_load_inputs(P, f, [PX, PY] + Q, first_input)
    P.misc([dict(op=OP_CLOAD, dst=BN_A + i, arg=P.c_one if i == 0 else P.c_zero) for i in range(12)], "f=1")
    P.misc([dict(op=OP_DOT, dst=d, terms=[("l", Lin.slot(q))], scale=1, mask=0, raw=True) for d, q in
            ((TX[0], Q[0]), (TX[1], Q[1]), (TY[0], Q[2]), (TY[1], Q[3]))]
           + [dict(op=OP_CLOAD, dst=TZ[0], arg=P.c_one), dict(op=OP_CLOAD, dst=TZ[1], arg=P.c_zero)], "T=Q")
    FF = T.reg(BN_A)

    def step():
        T.sqr12(BN_A, FF, "miller/sqr")
        bn_dbl_step(P, T, TX, TY, TZ, tmp, L, PX, PY, BN_A, mask)

    def add(src, sign):
        P.dot([Out(QT[i], [("l", Lin.slot(src[i]))], raw=True) for i in range(4)], "add/copyQ")
        bn_add_step(P, T, TX, TY, TZ, QT, sign, tmp, L, PX, PY, BN_A, mask)

    n = len(digits)
    run = 0
    for i in range(n - 1, 0, -1):
        run += 1
        d = digits[i - 1]
        if d:
            with P.repeat(run):
                step()
            run = 0
            add(Q, d)
    if run:
        with P.repeat(run):
            step()
    # Q1 = pi(Q) = (conj(x) xi^((p-1)/3), conj(y) xi^((p-1)/2)),  -Q2 = -pi^2(Q) = (x xi^((p^2-1)/3), y)
    k1x, k1y = _f2_pow(xi, (p - 1) // 3, p), _f2_pow(xi, (p - 1) // 2, p)
    k2x = _f2_pow(xi, (p * p - 1) // 3, p)
    assert k2x[1] == 0
    Q1 = [44, 45, 46, 47]
    xQ, yQ = E2.slots(Q[0], Q[1]), E2.slots(Q[2], Q[3])
    o = outs2(Q1[0], Q1[1], Acc2().prod_const(xQ.conj(), P.mont(k1x[0]), P.mont(k1x[1])))
    o += outs2(Q1[2], Q1[3], Acc2().prod_const(yQ.conj(), P.mont(k1y[0]), P.mont(k1y[1])))
    P.dot(o, "frobQ1")
    add(Q1, 1)
    o = outs2(Q1[0], Q1[1], Acc2().prod_const(xQ, P.mont(k2x[0]), None))
    o += outs2(Q1[2], Q1[3], Acc2().prod_const(yQ, (P.c_one, P.consts[P.c_one]), None))
    P.dot(o, "frobQ2")
    add(Q1, 1)
    if member_flag:
        bn_g2_member_check(P, curve, TX, TY, TZ, Q, tmp, member_flag)
    return FF


def bn_final_exp(P, T, fval, gam, curve=None):
    """set A <- finalExponentiation(f) (optate.go:215-264), the same exponent: with g the easy-part output and
    fu = g^u, fu2 = fu^u, fu3 = fu2^u
      y0 = g^p g^(p^2) g^(p^3), y1 = conj(g), y2 = fu2^(p^2), y3 = conj(fu^p), y4 = conj(fu fu2^p), y5 = conj(fu2),
      y6 = conj(fu3 fu3^p);  t0 = y6^2 y4 y5, t1 = y3 y5 t0, t0 = t0 y2, t1 = (t1^2 t0)^2, t0 = t1 y1, t1 = t1 y0,
      result = t0^2 t1.
    Every element after the easy part is in the cyclotomic subgroup, where Granger-Scott squaring equals the squaring.
    Five register sets, four global slots; `X~`: the slots hold conj(X)."""
    A_, B_, C_, D_, E_ = BN_A, BN_B, BN_C, BN_D, BN_E
    AA = tower_easy_part(P, T, fval, gam[2], A_, B_, C_, D_)   # spare slots: the first six of set D
    BB, CC, DD, EE = T.reg(B_), T.reg(C_), T.reg(D_), T.reg(E_)
    U = (curve or BN256).u
    T.frob12(B_, AA, 1, gam[1], "hard/gp")
    T.frob12(C_, AA, 2, gam[2], "hard/gp2")
    T.mul12(B_, BB, CC, "hard/y0a")
    T.frob12(C_, AA, 3, gam[3], "hard/gp3")
    T.mul12(B_, BB, CC, "hard/y0")
    T.spill12(B_, 0)                                        # gs0 = y0
    T.pow_cyclo(B_, AA, U, "hard/fu", cube_base=E_)          # B = fu
    T.frob12(C_, BB, 1, gam[1], "hard/fup")                 # C = fu^p = y3~
    T.spill12(C_, 1)                                        # gs1 = y3~
    T.pow_cyclo(C_, BB, U, "hard/fu2", cube_base=E_)         # C = fu2 = y5~
    T.frob12(D_, CC, 1, gam[1], "hard/fu2p")
    T.mul12(D_, BB, DD, "hard/y4")                          # D = fu fu2^p = y4~
    T.spill12(D_, 2)                                        # gs2 = y4~
    T.frob12(B_, CC, 2, gam[2], "hard/y2")                  # B = y2
    T.spill12(B_, 3)                                        # gs3 = y2
    T.pow_cyclo(B_, CC, U, "hard/fu3", cube_base=E_)         # B = fu3
    T.frob12(D_, BB, 1, gam[1], "hard/fu3p")
    T.mul12(D_, BB, DD, "hard/y6")                          # D = y6~
    T.cyclo_sqr(E_, DD, "hard/y6sq", refresh=True)          # E = (y6^2)~
    T.fill12(D_, 2)
    T.mul12(E_, EE, DD, "hard/t0a")                         # E = (y6^2 y4)~
    T.mul12(E_, EE, CC, "hard/t0b")                         # E = t0~ = (y6^2 y4 y5)~
    T.fill12(D_, 1)
    T.mul12(D_, DD, CC, "hard/t1a")                         # D = (y3 y5)~
    T.mul12(D_, DD, EE, "hard/t1b")                         # D = t1~
    T.fill12(B_, 3)
    T.mul12(E_, T.conj12(EE), BB, "hard/t0c")               # E = t0 y2
    T.cyclo_sqr(D_, DD, "hard/t1sq", refresh=True)          # D = (t1^2)~
    T.mul12(D_, T.conj12(DD), EE, "hard/t1c")               # D = t1^2 t0
    T.cyclo_sqr(D_, DD, "hard/t1d", refresh=True)           # D = t1 (new)
    T.mul12(E_, DD, T.conj12(AA), "hard/t0d")               # E = t1 y1
    T.fill12(B_, 0)
    T.mul12(D_, DD, BB, "hard/t1e")                         # D = t1 y0
    T.cyclo_sqr(E_, EE, "hard/t0sq", refresh=True)
    T.mul12(A_, EE, DD, "hard/out")
    return AA


def gt_layout_bn(j, c):
    """byte offset of coefficient (w^j, c) in pointGT.MarshalBinary (point.go:630-662; oracle gt_marshal)"""
    h, m = j & 1, j >> 1
    return ((1 - h) * 3 + (2 - m)) * 64 + (0 if c == 1 else 32)


def build_bn_pair(curve):
    f = curve.field()
    P = Prog(f, BN_NSLOTS, curve.xi0, n_inputs=6, n_gslots=4)
    T = Tower(P)
    gam = {K: frob_gammas(f.p, (curve.xi0, 1), K) for K in (1, 2, 3)}
    FF = bn_miller(P, T, f, 0, 0, curve, FLAG_G2_A if curve.strict_g2 else None)
    res = bn_final_exp(P, T, FF, gam, curve)
    one = (P.c_plain_one, 1)
    P.dot(sum((outs2(BN_A + 2 * j, BN_A + 2 * j + 1, Acc2().prod_const(res[j], one, None)) for j in range(6)), []), "to_plain")
    P.misc([dict(op=OP_GT_STORE, dst=BN_A + 2 * j + c, arg=gt_layout_bn(j, c) | ((1 if (j == 0 and c == 0) else 0) << 16))
            for j in range(6) for c in range(2)], "gt_store")
    return P


def build_bn_check(curve):
    """Suite.ValidatePairing (bn256 suite.go:105-107, bn254 suite.go:134-140): Pair(p1, p2).Equal(Pair(inv1, inv2)) -- two
    whole pairings compared coefficient by coefficient (bn256 accepts G2 points outside the order-n subgroup, for which
    the product form e(p1, p2) e(-inv1, inv2) == 1 need not be equivalent).  Inputs 0..5: pair A, 6..11: pair B.  Flag bit
    0 / 1: pair A / B has an operand at infinity and pairs to one (optate.go:270-272)."""
    f = curve.field()
    P = Prog(f, BN_NSLOTS, curve.xi0, n_inputs=12, n_gslots=5)
    T = Tower(P)
    gam = {K: frob_gammas(f.p, (curve.xi0, 1), K) for K in (1, 2, 3)}
    one = (P.c_plain_one, 1)
    for k in range(2):
        FF = bn_miller(P, T, f, 6 * k, 0, curve)
        res = bn_final_exp(P, T, FF, gam, curve)
        # plain residues into set E preloaded with the identity; lanes whose pair is dead keep the identity
        P.misc([dict(op=OP_CLOAD, dst=BN_E + i, arg=P.c_plain_one if i == 0 else P.c_zero) for i in range(12)], "one")
        P.dot(sum((outs2(BN_E + 2 * j, BN_E + 2 * j + 1, Acc2().prod_const(res[j], one, None), mask=k + 1) for j in range(6)), []),
              "to_plain")
        if k == 0:
            T.spill12(BN_E, 4)
    T.fill12(BN_D, 4)
    P.misc([dict(op=OP_CMP_EQ, dst=BN_E + i, arg=BN_D + i) for i in range(12)], "equal")
    return P


def build_bn_check_product(curve, member=True, zero_flag=False):
    """ValidatePairing as ONE final exponentiation: ok = (f_{Q1}(P1) f_{Q2}(-P2))^e == 1, the two Miller loops sharing the
    squarings of f.  Equivalent to the reference's two pairings + Equal whenever both G2 operands lie in the order-n
    subgroup -- which pairing/bn254's UnmarshalBinary guarantees (twist.go:47-66); bn256, whose G2 is unchecked, keeps
    build_bn_check.  (e(-P, Q) = e(P, Q)^-1: negating y_P negates the w^0 coefficient of every line, i.e. the Miller
    value is conjugated up to sign, and conjugation inverts after the easy part.)  Inputs 0..5: P1, Q1; 6..11: P2 with y
    ALREADY NEGATED by the operand kernel, Q2.  Flag bit 0 / 1: pair A / B has an operand at infinity and contributes 1."""
    f = curve.field()
    P = Prog(f, BN_NSLOTS, curve.xi0, n_inputs=12, n_gslots=4)
    T = Tower(P)
    p = f.p
    xi = (curve.xi0, 1)
    gam = {K: frob_gammas(p, xi, K) for K in (1, 2, 3)}
    Ts = [((12, 13), (14, 15), (16, 17)), ((18, 19), (20, 21), (22, 23))]
    tmp = list(range(24, 34))
    L = list(range(34, 40))
    Ps = [(40, 41), (42, 43)]
    Qs = [[44, 45, 46, 47], [48, 49, 50, 51]]
    QT = [52, 53, 54, 55]
    QF = [56, 57, 58, 59]
    for k in range(2):
        bls_load_inputs(P, f, [Ps[k][0], Ps[k][1]] + Qs[k], 6 * k)
    P.misc([dict(op=OP_CLOAD, dst=BN_A + i, arg=P.c_one if i == 0 else P.c_zero) for i in range(12)], "f=1")
    for k in range(2):
        (TX, TY, TZ), Q = Ts[k], Qs[k]
        P.misc([dict(op=OP_DOT, dst=d, terms=[("l", Lin.slot(q))], scale=1, mask=0, raw=True) for d, q in
                ((TX[0], Q[0]), (TX[1], Q[1]), (TY[0], Q[2]), (TY[1], Q[3]))]
               + [dict(op=OP_CLOAD, dst=TZ[0], arg=P.c_one), dict(op=OP_CLOAD, dst=TZ[1], arg=P.c_zero)], "T=Q")
    FF = T.reg(BN_A)

    def step():
        T.sqr12(BN_A, FF, "miller/sqr")
        for k in range(2):
            bn_dbl_step(P, T, *Ts[k], tmp, L, Ps[k][0], Ps[k][1], BN_A, k + 1)

    def add(srcs, sign):
        for k in range(2):
            P.dot([Out(QT[i], [("l", Lin.slot(srcs[k][i]))], raw=True) for i in range(4)], "add/copyQ")
            bn_add_step(P, T, *Ts[k], QT, sign, tmp, L, Ps[k][0], Ps[k][1], BN_A, k + 1)

    digits = curve.digits
    run = 0
    for i in range(len(digits) - 1, 0, -1):
        run += 1
        d = digits[i - 1]
        if d:
            with P.repeat(run):
                step()
            run = 0
            add(Qs, d)
    if run:
        with P.repeat(run):
            step()
    k1x, k1y = _f2_pow(xi, (p - 1) // 3, p), _f2_pow(xi, (p - 1) // 2, p)
    k2x = _f2_pow(xi, (p * p - 1) // 3, p)
    for which in (1, 2):  # pi(Q), then -pi^2(Q), for both pairs in turn (QF is recycled)
        for k in range(2):
            xQ, yQ = E2.slots(Qs[k][0], Qs[k][1]), E2.slots(Qs[k][2], Qs[k][3])
            if which == 1:
                o = outs2(QF[0], QF[1], Acc2().prod_const(xQ.conj(), P.mont(k1x[0]), P.mont(k1x[1])))
                o += outs2(QF[2], QF[3], Acc2().prod_const(yQ.conj(), P.mont(k1y[0]), P.mont(k1y[1])))
            else:
                o = outs2(QF[0], QF[1], Acc2().prod_const(xQ, P.mont(k2x[0]), None))
                o += outs2(QF[2], QF[3], Acc2().prod_const(yQ, (P.c_one, P.consts[P.c_one]), None))
            P.dot(o, "frobQ%d" % which)
            P.dot([Out(QT[i], [("l", Lin.slot(QF[i]))], raw=True) for i in range(4)], "add/copyQ")
            bn_add_step(P, T, *Ts[k], QT, 1, tmp, L, Ps[k][0], Ps[k][1], BN_A, k + 1)
    if curve.strict_g2 and member:
        for k in range(2):
            bn_g2_member_check(P, curve, *Ts[k], Qs[k], tmp, FLAG_G2_B if k else FLAG_G2_A)
    one = (P.c_plain_one, 1)
    if zero_flag:
        # Is the joint Miller value zero?  (Only a G2 operand with a component of tiny order makes a line vanish; then
        # e(p1, p2) = 0 or e(inv1, inv2) = 0 in the reference's terms, and "0 == 0" is TRUE there while the product is
        # never one.)  The twelve coefficients in plain form (set B is free: the points are dead), each raising
        # FLAG_MILLER_NONZERO unless it is zero.
        P.dot(sum((outs2(BN_B + 2 * j, BN_B + 2 * j + 1, Acc2().prod_const(FF[j], one, None)) for j in range(6)), []), "miller/to_plain")
        P.misc([dict(op=OP_IS_ONE, dst=BN_B + i, arg=0, flag=FLAG_MILLER_NONZERO) for i in range(12)], "miller/nonzero")
    res = bn_final_exp(P, T, FF, gam, curve)
    P.dot(sum((outs2(BN_A + 2 * j, BN_A + 2 * j + 1, Acc2().prod_const(res[j], one, None)) for j in range(6)), []), "to_plain")
    P.misc([dict(op=OP_IS_ONE, dst=BN_A + 2 * j + c, arg=(1 if (j == 0 and c == 0) else 0) << 16) for j in range(6) for c in range(2)],
           "is_one")
    return P


def build_bn_gtmul(curve):
    """pointGT.Mul / gfP12.Exp (pairing/bn256/point.go:613, gfp12.go:177-192): a^k for ANY element of Fp12 -- the
    reference's UnmarshalBinary checks no membership, so the squarings are general ones.  Inputs 0..11 = the twelve
    coefficients (w-basis, re / im), the lane's exponent through store mask 3; square-and-multiply from bit 255."""
    f = curve.field()
    P = Prog(f, BN_NSLOTS, curve.xi0, n_inputs=12, n_gslots=0)
    T = Tower(P)
    bls_load_inputs(P, f, list(range(BN_A, BN_A + 12)), 0)
    AA, BB = T.reg(BN_A), T.reg(BN_B)
    P.misc([dict(op=OP_CLOAD, dst=BN_B + i, arg=P.c_one if i == 0 else P.c_zero) for i in range(12)], "exp/acc=1")
    with P.repeat(GT_EXP_BITS):
        T.sqr12(BN_B, BB, "exp/sqr")
        T.mul12(BN_B, BB, AA, "exp/mul", mask=MASK_EXPO, ebit=(GT_EXP_BITS - 1, 1))
    _gt_store(P, BN_B, gt_layout_bn)
    return P


def build_bn256_pair(): return build_bn_pair(BN256)
def build_bn256_gtmul(): return build_bn_gtmul(BN256)
def build_bn254_gtmul(): return build_bn_gtmul(BN254)
def build_bn256_check(): return build_bn_check(BN256)
def build_bn254_pair(): return build_bn_pair(BN254)
def build_bn254_check(): return build_bn_check_product(BN254)
def build_bn256_check_product(): return build_bn_check_product(BN256, zero_flag=True)


# ------------------------------------------------------------------------------------------------ emission
def _carr(vals, fmt="0x%xu", per=16):
    lines = []
    for i in range(0, len(vals), per):
        lines.append(", ".join(fmt % v for v in vals[i:i + per]))
    return "{\n" + ",\n".join(lines) + "}"


def emit_field(f, struct_name):
    p = f.p

    def digits(x):
        assert x >> (f.W * (f.N + 1)) == 0
        return [(x >> (f.W * i)) & ((1 << f.W) - 1) for i in range(f.N + 1)]

    def arr(v):
        return ", ".join("0x%xu" % d for d in v)

    return "\n".join([
        f"struct {struct_name} {{",
        f"    static constexpr int N = {f.N}, NW = {f.NW}, W = {f.W};  // limbs, packed words, bits per limb",
        f"    static constexpr int32_t P[{f.N}] = {{{', '.join(str(d) for d in f.balanced(p))}}};  // balanced W-bit digits of p",
        f"    static constexpr uint32_t NINV = 0x{f.ninv:x}u;  // -p^-1 mod 2^W",
        f"    static constexpr bool OPAQUE_P = {'true' if any(abs(d) > 1 and abs(d) & (abs(d) - 1) == 0 for d in f.balanced(p)) else 'false'};  // a digit of p is +-2^k (tower_vm.cuh mont_reduce)",
        f"    static constexpr uint32_t P1[{f.N + 1}] = {{{arr(digits(p))}}};  // p, 2p, 4p: unsigned W-bit digits",
        f"    static constexpr uint32_t P2[{f.N + 1}] = {{{arr(digits(2 * p))}}};",
        f"    static constexpr uint32_t P4[{f.N + 1}] = {{{arr(digits(4 * p))}}};",
        "};"])


def emit_prog(P, name):
    prog, sched = P.encode()
    consts = []
    for c in P.consts:
        consts += [d & 0xffffffff for d in P.f.balanced(c)] + [0] * (16 - P.f.N)
    flat = []
    for s in sched:
        flat += [s[0], s[1], s[2], 0]
    table = []
    for c in P.gconsts:
        table += [d & 0xffffffff for d in P.f.balanced(c)] + [0] * (16 - P.f.N)
    gtab = [f"static __device__ const uint32_t TVM_{name}_GCONSTS[{len(table)}] = {_carr(table)};  // {len(P.gconsts)} per-repetition constants",
            ""] if table else []
    return "\n".join(gtab + [
        f"// program {name}: {len(prog) // (WAVES * REC_WORDS)} stored instructions, {sum(s[1] * s[2] for s in sched)} executed",
        f"static __device__ const uint32_t TVM_{name}_PROG[{len(prog)}] = {_carr(prog)};",
        f"static __device__ const uint32_t TVM_{name}_SCHED[{len(flat)}] = {_carr(flat)};",
        f"static constexpr uint32_t TVM_{name}_NSCHED = {len(sched)};",
        f"static __device__ const uint32_t TVM_{name}_CONSTS[{len(consts)}] = {_carr(consts)};",
        f"static constexpr uint32_t TVM_{name}_NCONSTS = {len(P.consts)}, TVM_{name}_NDYNCONSTS = {P.ndyn if P.dyn_base is not None else 0};",
        f"static constexpr uint32_t TVM_{name}_NGSLOTS = {P.n_gslots}, TVM_{name}_NINPUTS = {P.n_inputs}, TVM_{name}_NSLOTS = {P.nslots};",
        f"static constexpr uint64_t TVM_{name}_MADS_PER_UNIT = {P.mads()}ull;  // integer MADs per pairing (check), all waves",
        ""])


def main():
    for suite, struct, builders in (("bls12381", "Bls12381Vm", (build_bls12381_pair, build_bls12381_check)),
                                    ("bn256", "Bn256Vm", (build_bn256_pair, build_bn256_check)),
                                    ("bn254", "Bn254Vm", (build_bn254_pair, build_bn254_check))):
        pair, check = builders[0](), builders[1]()
        up = suite.upper()
        out = ["// generated by gen_tower_vm.py -- do not edit", "#pragma once", "#include <stdint.h>", "namespace kyb {",
               emit_field(pair.f, struct), emit_prog(pair, up + "_PAIR"), emit_prog(check, up + "_CHECK")]
        gtmul = {"bls12381": build_bls12381_gtmul, "bn256": build_bn256_gtmul, "bn254": build_bn254_gtmul}[suite]()
        out.append(emit_prog(gtmul, up + "_GTMUL"))  # GT exponentiation (store mask 3: the lane's exponent bits)
        if suite == "bn256":  # the product form, for calls whose G2 operands the caller vouches for (bn_pair.inc)
            out.append(emit_prog(build_bn256_check_product(), up + "_CHECKP"))
        if suite == "bls12381":  # CHECK with the second G2 operand fixed to the generator (bls.Verify on G1)
            out.append(emit_prog(build_bls12381_verify(), up + "_VERIFY"))
            vk = build_bls12381_verify_same_key()  # both G2 operands fixed: the key's lines come from a per-key table
            out.append(emit_prog(vk, up + "_VERIFYK"))
            out.append(f"static constexpr uint32_t TVM_{up}_VERIFYK_KEY_TABLE_BASE = {vk.key_table_base};  // entries; the key's half starts here")
        out += ["}  // namespace kyb", ""]
        dst = os.path.join(HERE, "tower_vm_%s.inc" % suite)   # written whole, then renamed: a reader never sees half a file
        with open(dst + ".tmp", "w") as f:
            f.write("\n".join(out))
        os.replace(dst + ".tmp", dst)
        for n, P in (("pair", pair), ("check", check), ("gtmul", gtmul)):
            print(suite, n, P.stats(), "bounds (log2 column, value/p):", P.check_bounds())


if __name__ == "__main__":
    main()
