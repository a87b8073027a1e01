// Short-Weierstrass (a = 0) group arithmetic in Jacobian coordinates over any field of the tower
// (F = Fp<C> for G1, Fp2<T> for G2), one point per lane.
//
// Replaces: pairing/bn256 curvePoint/twistPoint Add/Double/Mul/MakeAffine (curve.go:69,156,189,205;
// twist.go:162) and the g1/g2 layers of the external BLS12-381 backends (kilic/g1.go:110-116 ->
// MulScalarBig).  Only canonical affine encodings are observable, so the scalar multiplication is
// free to use a signed radix-16 window instead of the reference's bit-serial double-and-add.
#pragma once
#include "tower.cuh"
#include "fp_limbs.cuh"

namespace kyb {

// Code size.  The instruction cache of a CU pair holds 64 KB; an inlined base-field multiplication is 4-5 KB of code
// (BLS12-381), so a point addition over Fp2 with every multiplier inlined is a 200 KB straight line and a window step
// of the ladder twice that: each wave then streams its instructions from L2 / HBM on every iteration (measured on
// MI355X, profiles/r03_codesize_*: G2 subgroup check 6.7 -> 4.1 ms per 2^16, bn256 G2 ladder 2x).  The point formulas
// therefore CALL the Fp2 multiplier and squarer (one 13 KB / 8 KB copy each, operands through the lane's scratch, which
// stays in L1 / L2), and the Fp ones too for fields of KYB_OUTLINE_FP_MIN_WORDS words and up, so that a whole ladder
// iteration fits the cache.
#ifndef KYB_OUTLINE_FP_MIN_WORDS
#define KYB_OUTLINE_FP_MIN_WORDS 99
#endif
template <class C> KYB_HD_NOINLINE void fp_mul_c(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) { fp_mul(r, a, b); }
template <class C> KYB_HD_NOINLINE void fp_sqr_c(Fp<C>& r, const Fp<C>& a) { fp_sqr(r, a); }

// ---- uniform names over Fp / Fp2 so the point formulas are written once
template <class C> KYB_HD void f_add(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) { fp_add(r, a, b); }
template <class C> KYB_HD void f_sub(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) { fp_sub(r, a, b); }
template <class C> KYB_HD void f_mul(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
    if constexpr (C::NWORDS >= KYB_OUTLINE_FP_MIN_WORDS) fp_mul_c(r, a, b);
    else fp_mul(r, a, b);
}
template <class C> KYB_HD void f_sqr(Fp<C>& r, const Fp<C>& a) {
    if constexpr (C::NWORDS >= KYB_OUTLINE_FP_MIN_WORDS) fp_sqr_c(r, a);
    else fp_sqr(r, a);
}
template <class C> KYB_HD void f_dbl(Fp<C>& r, const Fp<C>& a) { fp_dbl(r, a); }
template <class C> KYB_HD void f_neg(Fp<C>& r, const Fp<C>& a) { fp_neg(r, a); }
template <class C> KYB_HD void f_inv(Fp<C>& r, const Fp<C>& a) { fp_inv(r, a); }
template <class C> KYB_HD void f_zero(Fp<C>& r) { fp_zero(r); }
template <class C> KYB_HD void f_one(Fp<C>& r) { fp_one(r); }
template <class C> KYB_HD bool f_is_zero(const Fp<C>& a) { return fp_is_zero(a); }
template <class C> KYB_HD bool f_eq(const Fp<C>& a, const Fp<C>& b) { return fp_eq(a, b); }
template <class C> KYB_HD void f_cmov(Fp<C>& r, const Fp<C>& a, bool c) { fp_cmov(r, a, c); }
template <class T> KYB_HD void f_add(Fp2<T>& r, const Fp2<T>& a, const Fp2<T>& b) { fp2_add(r, a, b); }
template <class T> KYB_HD void f_sub(Fp2<T>& r, const Fp2<T>& a, const Fp2<T>& b) { fp2_sub(r, a, b); }
#ifdef KYB_OUTLINE_FP2
template <class T> KYB_HD void f_mul(Fp2<T>& r, const Fp2<T>& a, const Fp2<T>& b) { fp2_mul_c(r, a, b); }
template <class T> KYB_HD void f_sqr(Fp2<T>& r, const Fp2<T>& a) { fp2_sqr_c(r, a); }
#else
template <class T> KYB_HD void f_mul(Fp2<T>& r, const Fp2<T>& a, const Fp2<T>& b) { fp2_mul(r, a, b); }
template <class T> KYB_HD void f_sqr(Fp2<T>& r, const Fp2<T>& a) { fp2_sqr(r, a); }
#endif
template <class T> KYB_HD void f_dbl(Fp2<T>& r, const Fp2<T>& a) { fp2_dbl(r, a); }
template <class T> KYB_HD void f_neg(Fp2<T>& r, const Fp2<T>& a) { fp2_neg(r, a); }
template <class T> KYB_HD void f_inv(Fp2<T>& r, const Fp2<T>& a) { fp2_inv(r, a); }
template <class T> KYB_HD void f_zero(Fp2<T>& r) { fp2_zero(r); }
template <class T> KYB_HD void f_one(Fp2<T>& r) { fp2_one(r); }
template <class T> KYB_HD bool f_is_zero(const Fp2<T>& a) { return fp2_is_zero(a); }
template <class T> KYB_HD bool f_eq(const Fp2<T>& a, const Fp2<T>& b) { return fp2_eq(a, b); }
template <class T> KYB_HD void f_cmov(Fp2<T>& r, const Fp2<T>& a, bool c) { fp2_cmov(r, a, c); }

template <class F>
struct Jac {  // (X : Y : Z), x = X/Z^2, y = Y/Z^3, infinity <=> Z = 0
    F X, Y, Z;
};
template <class F>
struct Aff {
    F x, y;
    bool inf;
};

template <class F> KYB_HD void jac_set_inf(Jac<F>& r) { f_one(r.X); f_one(r.Y); f_zero(r.Z); }
template <class F> KYB_HD bool jac_is_inf(const Jac<F>& p) { return f_is_zero(p.Z); }
template <class F>
KYB_HD void jac_from_aff(Jac<F>& r, const Aff<F>& a) {
    r.X = a.x;
    r.Y = a.y;
    f_one(r.Z);
    if (a.inf) jac_set_inf(r);
}
template <class F>
KYB_HD void jac_cmov(Jac<F>& r, const Jac<F>& a, bool c) {
    f_cmov(r.X, a.X, c);
    f_cmov(r.Y, a.Y, c);
    f_cmov(r.Z, a.Z, c);
}
template <class F> KYB_HD void jac_neg(Jac<F>& r, const Jac<F>& p) { r.X = p.X; f_neg(r.Y, p.Y); r.Z = p.Z; }

// dbl-2009-l (a = 0): 2M + 5S.  Maps infinity to infinity and 2-torsion points (Y = 0) to infinity.
template <class F>
KYB_HD void jac_dbl_inl(Jac<F>& r, const Jac<F>& p) {
    F A, B, C, D, E, G, t;
    f_sqr(A, p.X);
    f_sqr(B, p.Y);
    f_sqr(C, B);
    f_add(t, p.X, B);
    f_sqr(t, t);
    f_sub(t, t, A);
    f_sub(t, t, C);
    f_dbl(D, t);  // 4 X Y^2
    f_dbl(E, A);
    f_add(E, E, A);  // 3 X^2
    f_sqr(G, E);
    f_mul(t, p.Y, p.Z);
    f_dbl(r.Z, t);  // uses p.Y, p.Z before they are overwritten (r may alias p)
    f_dbl(t, D);
    f_sub(r.X, G, t);
    f_sub(t, D, r.X);
    f_mul(t, E, t);
    f_dbl(C, C);
    f_dbl(C, C);
    f_dbl(C, C);
    f_sub(r.Y, t, C);
}
template <class F>
KYB_HD_NOINLINE void jac_dbl(Jac<F>& r, const Jac<F>& p) {
    jac_dbl_inl(r, p);
}
// r = 2^n p: a run of doublings inside ONE out-of-line call, so the point stays in registers across the run
// instead of crossing the call boundary (scratch) 2n times -- the window steps of the scalar multiplications (n = 4)
// and the zero runs of the sparse curve parameter in the subgroup checks (up to 32).
template <class F>
KYB_HD_NOINLINE void jac_dbl_n(Jac<F>& r, const Jac<F>& p, int n) {
    Jac<F> x = p;
#pragma unroll 1
    for (int i = 0; i < n; i++) jac_dbl_inl(x, x);
    r = x;
}

// add-2007-bl with the exceptional cases handled (either operand infinity, P = Q, P = -Q).
// (CALL_DBL = false inlines the doubling of the P = Q case too: a kernel without any out-of-line callee keeps its own
// register budget -- the cooperative tail kernels of the MSM)
template <class F, bool CALL_DBL = true>
KYB_HD void jac_add_inl(Jac<F>& r, const Jac<F>& p, const Jac<F>& q) {
    const bool pinf = jac_is_inf(p), qinf = jac_is_inf(q);
    F Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t;
    f_sqr(Z1Z1, p.Z);
    f_sqr(Z2Z2, q.Z);
    f_mul(U1, p.X, Z2Z2);
    f_mul(U2, q.X, Z1Z1);
    f_mul(t, q.Z, Z2Z2);
    f_mul(S1, p.Y, t);
    f_mul(t, p.Z, Z1Z1);
    f_mul(S2, q.Y, t);
    f_sub(H, U2, U1);
    f_sub(rr, S2, S1);
    if (!pinf && !qinf && f_is_zero(H)) {  // same x: doubling or cancellation (rare; divergence is fine)
        if (f_is_zero(rr)) {
            if constexpr (CALL_DBL) jac_dbl(r, p);
            else jac_dbl_inl(r, p);
        } else {
            jac_set_inf(r);
        }
        return;
    }
    Jac<F> o;
    f_dbl(rr, rr);
    f_dbl(I, H);
    f_sqr(I, I);
    f_mul(J, H, I);
    f_mul(V, U1, I);
    f_sqr(o.X, rr);
    f_sub(o.X, o.X, J);
    f_sub(o.X, o.X, V);
    f_sub(o.X, o.X, V);
    f_sub(t, V, o.X);
    f_mul(t, rr, t);
    f_mul(S1, S1, J);
    f_dbl(S1, S1);
    f_sub(o.Y, t, S1);
    f_add(t, p.Z, q.Z);
    f_sqr(t, t);
    f_sub(t, t, Z1Z1);
    f_sub(t, t, Z2Z2);
    f_mul(o.Z, t, H);
    jac_cmov(o, q, pinf);
    jac_cmov(o, p, qinf);
    r = o;
}
template <class F>
KYB_HD_NOINLINE void jac_add(Jac<F>& r, const Jac<F>& p, const Jac<F>& q) {
    jac_add_inl(r, p, q);
}
// t = d * P from the table tab[j] = (j + 1) P, d a signed radix-16 digit (|d| <= 8; d = 0 returns tab[0], unused)
template <class F>
KYB_HD void jac_select8(Jac<F>& t, const Jac<F> (&tab)[8], int d) {
    const int ad = d < 0 ? -d : d;
    t = tab[ad ? ad - 1 : 0];
    F ny;
    f_neg(ny, t.Y);
    f_cmov(t.Y, ny, d < 0);
}
// The same from an AFFINE table (jac_table8_to_affine: Z is one, or zero for an entry at infinity): only the two
// coordinates a mixed addition takes and the flag -- no copy of Z through the lane's scratch.
template <class F>
KYB_HD void jac_select8_xy(F& x, F& y, bool& inf, const Jac<F> (&tab)[8], int d) {
    const int ad = d < 0 ? -d : d;
    const Jac<F>& e = tab[ad ? ad - 1 : 0];
    x = e.X;
    inf = f_is_zero(e.Z);
    F ny;
    f_neg(ny, e.Y);
    y = e.Y;
    f_cmov(y, ny, d < 0);
}
// Mixed addition r = p + (x2, y2) with the second operand affine (madd-2007-bl, 7M + 4S), exceptional
// cases handled: p at infinity, q at infinity (q_inf), p = q (doubling), p = -q (infinity).
template <class F>
KYB_HD void jac_madd_inl(Jac<F>& r, const Jac<F>& p, const F& x2, const F& y2, bool q_inf) {
    const bool pinf = jac_is_inf(p);
    F Z1Z1, U2, S2, H, HH, I, J, rr, V, t;
    f_sqr(Z1Z1, p.Z);
    f_mul(U2, x2, Z1Z1);
    f_mul(t, p.Z, Z1Z1);
    f_mul(S2, y2, t);
    f_sub(H, U2, p.X);
    f_sub(rr, S2, p.Y);
    if (!pinf && !q_inf && f_is_zero(H)) {
        if (f_is_zero(rr)) {
            jac_dbl(r, p);
        } else {
            jac_set_inf(r);
        }
        return;
    }
    Jac<F> o;
    f_sqr(HH, H);
    f_dbl(I, HH);
    f_dbl(I, I);
    f_mul(J, H, I);
    f_dbl(rr, rr);
    f_mul(V, p.X, I);
    f_sqr(o.X, rr);
    f_sub(o.X, o.X, J);
    f_sub(o.X, o.X, V);
    f_sub(o.X, o.X, V);
    f_sub(t, V, o.X);
    f_mul(t, rr, t);
    f_mul(J, p.Y, J);
    f_dbl(J, J);
    f_sub(o.Y, t, J);
    f_add(t, p.Z, H);
    f_sqr(t, t);
    f_sub(t, t, Z1Z1);
    f_sub(o.Z, t, HH);
    Jac<F> q;
    q.X = x2;
    q.Y = y2;
    f_one(q.Z);
    jac_cmov(o, q, pinf);
    jac_cmov(o, p, q_inf);
    r = o;
}
template <class F>
KYB_HD_NOINLINE void jac_madd(Jac<F>& r, const Jac<F>& p, const F& x2, const F& y2, bool q_inf) {
    jac_madd_inl(r, p, x2, y2, q_inf);
}

// acc <- take ? acc + (x2, y2) : acc in ONE out-of-line call: the accumulator crosses the call boundary (the lane's
// scratch) once in and once out.  As jac_madd(s, acc, ...) followed by jac_cmov(acc, s, take) at the call site it crossed
// five times (the sum out, both operands of the select back in, the accumulator out again): 770 of the ~1 200 bytes a
// window addition of the G2 ladders moved, with 158 KB of such traffic per bn256 G2 multiplication measured
// (profiles/r04_final_bn256_{fetch,write}.txt).
template <class F>
KYB_HD_NOINLINE void jac_madd_if(Jac<F>& acc, const F& x2, const F& y2, bool q_inf, bool take) {
    Jac<F> o;
    jac_madd_inl(o, acc, x2, y2, q_inf);
    jac_cmov(acc, o, take);
}
// The eight-entry window table tab[j] = (j + 1) P of the fixed-window ladders, brought to AFFINE form in place (Z = 1;
// an entry at infinity keeps Z = 0) with ONE shared inversion (Montgomery's trick: 21 multiplications + the
// inversion + 4 per entry): every table addition of the ladder is then a mixed addition, 7M + 4S instead of
// 11M + 5S -- 64 to 72 additions per scalar multiplication against ~95 multiplication-equivalents for the conversion.
// (Inlined into the ladder that owns the table: as an out-of-line function writing the caller's private array through
// the reference, the gfx950 build faulted with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION -- every ladder, both
// fields, ROCm 7.2; the same body inlined, or the same call with the body compiled out, runs.)
template <class F>
KYB_HD void jac_table8_to_affine(Jac<F> (&tab)[8]) {
    F c[8], one, inv, zi, zi2;
    f_one(one);
    uint32_t infm = 0;
#pragma unroll 1
    for (int j = 0; j < 8; j++) {
        const bool z0 = f_is_zero(tab[j].Z);
        infm |= (z0 ? 1u : 0u) << j;
        f_cmov(tab[j].Z, one, z0);  // a placeholder keeps the running product invertible
        if (j == 0) c[0] = tab[0].Z;
        else f_mul(c[j], c[j - 1], tab[j].Z);
    }
    f_inv(inv, c[7]);
#pragma unroll 1
    for (int j = 7; j >= 0; j--) {
        if (j > 0) {
            f_mul(zi, inv, c[j - 1]);      // 1 / Z_j
            f_mul(inv, inv, tab[j].Z);     // 1 / (Z_0 ... Z_{j-1})
        } else {
            zi = inv;
        }
        f_sqr(zi2, zi);
        f_mul(tab[j].X, tab[j].X, zi2);
        f_mul(zi2, zi2, zi);
        f_mul(tab[j].Y, tab[j].Y, zi2);
        tab[j].Z = one;
        if ((infm >> j) & 1u) f_zero(tab[j].Z);
    }
}
// One window against an affine table (jac_table8_to_affine): acc = 16 acc (when dbl) + d P, in one out-of-line call.
template <class F>
KYB_HD_NOINLINE void jac_window_step_aff(Jac<F>& acc, const Jac<F> (&tab)[8], int d, bool dbl) {
    Jac<F> x = acc, t, s;
    if (dbl) {  // uniform across the grid
#pragma unroll 1
        for (int i = 0; i < 4; i++) jac_dbl_inl(x, x);
    }
    jac_select8(t, tab, d);
    jac_madd_inl(s, x, t.X, t.Y, jac_is_inf(t));
    jac_cmov(x, s, d != 0);
    acc = x;
}

// Extended Jacobian ("XYZZ") accumulator for RUNS of mixed additions (the MSM's bucket pieces): x = X / ZZ,
// y = Y / ZZZ with ZZ^3 = ZZZ^2, infinity <=> ZZ = 0.  A mixed addition is 8M + 2S and seven subtractions
// (madd-2008-s) against 7M + 4S and fifteen for madd-2007-bl; the rare cases are BRANCHES instead of selects (the
// accumulator is at infinity only before its first point or after a cancellation, equal x only for repeated or
// opposite points), so the common path carries no conditional moves at all.  Leaving the form costs 2M.
template <class F>
struct Xyzz {
    F X, Y, ZZ, ZZZ;
};
template <class F> KYB_HD void xyzz_set_inf(Xyzz<F>& r) { f_one(r.X); f_one(r.Y); f_zero(r.ZZ); f_zero(r.ZZZ); }
template <class F> KYB_HD bool xyzz_is_inf(const Xyzz<F>& p) { return f_is_zero(p.ZZ); }
// (X ZZ : Y ZZZ : ZZ) is the same point in Jacobian coordinates (Z = ZZ: Z^2 = ZZ^2, Z^3 = ZZZ^2).
template <class F>
KYB_HD void xyzz_to_jac(Jac<F>& r, const Xyzz<F>& p) {
    f_mul(r.X, p.X, p.ZZ);
    f_mul(r.Y, p.Y, p.ZZZ);
    r.Z = p.ZZ;
}
// r += (x2, y2), a finite affine point (the caller skips points at infinity).
template <class F>
KYB_HD void xyzz_madd(Xyzz<F>& r, const F& x2, const F& y2) {
    if (xyzz_is_inf(r)) {  // first point of a run, or the run cancelled so far
        r.X = x2;
        r.Y = y2;
        f_one(r.ZZ);
        f_one(r.ZZZ);
        return;
    }
    F U2, S2, P, R, PP, PPP, Q, t;
    f_mul(U2, x2, r.ZZ);
    f_mul(S2, y2, r.ZZZ);
    f_sub(P, U2, r.X);
    f_sub(R, S2, r.Y);
    if (f_is_zero(P)) {  // same x: the point itself (double it) or its inverse (cancel)
        if (f_is_zero(R)) {
            Jac<F> j;
            j.X = x2;
            j.Y = y2;
            f_one(j.Z);
            jac_dbl(j, j);
            r.X = j.X;
            r.Y = j.Y;
            f_sqr(r.ZZ, j.Z);
            f_mul(r.ZZZ, r.ZZ, j.Z);
        } else {
            xyzz_set_inf(r);
        }
        return;
    }
    f_sqr(PP, P);
    f_mul(PPP, P, PP);
    f_mul(Q, r.X, PP);
    f_sqr(t, R);
    f_sub(t, t, PPP);
    f_sub(t, t, Q);
    f_sub(t, t, Q);  // X3 = R^2 - PPP - 2 Q
    f_sub(Q, Q, t);
    f_mul(Q, R, Q);
    f_mul(S2, r.Y, PPP);
    r.X = t;
    f_sub(r.Y, Q, S2);  // Y3 = R (Q - X3) - Y1 PPP
    f_mul(r.ZZ, r.ZZ, PP);
    f_mul(r.ZZZ, r.ZZZ, PPP);
}

// The XYZZ accumulator in limb form (fp_limbs.cuh) for base fields with headroom (BLS12-381 Fp): a run of mixed
// additions never packs, never compares with p and never selects.  Lazy bounds, as multiples of p (R / p > 2^9):
//   X < 8, Y < 2, ZZ < 2, ZZZ < 2 between additions (the first point of a run: x2, y2 < 1, ZZ = ZZZ = one);
//   U2 = x2 ZZ, S2 = y2 ZZZ < 2;  P = U2 - X + 8p < 10;  R = +-S2 - Y + 4p < 6 (the point's sign is applied HERE, to S2,
//   instead of negating y2);  PP = P^2 (100), PPP = P PP (20), Q = X PP (16), R^2 (36): all < 2;
//   X3 = R^2 - PPP - 2Q + 6p < 8;  Q - X3 + 8p < 10;  2p - PPP in (0, 2];
//   Y3 = R (Q - X3) + Y (2p - PPP): ONE reduction over both products (60 + 4 < R / p), < 2;  ZZ PP, ZZZ PPP < 2.
// 6M + 2S + one two-product multiplication = 3 055 multiply-adds (3 224 for 8M + 2S) and five limb sweeps; the packed
// form's 20 unpacks, 10 packs and 17 conditional subtractions are gone (accumulate_kernel's loop: 5 779 -> ~4 700
// instructions).  Equal x (P = 0 mod p) is decided exactly by fpl_is_zero_mod_p; infinity is a flag.
template <class C>
struct XyzzL {
    FpL<C> X, Y, ZZ, ZZZ;
    uint32_t inf;
};
template <class C> KYB_HD void xyzzl_set_inf(XyzzL<C>& r) {
    fpl_one(r.X);
    fpl_one(r.Y);
    fpl_one(r.ZZ);
    fpl_one(r.ZZZ);
    r.inf = 1u;
}
template <class C>
KYB_HD void xyzzl_to_jac(Jac<Fp<C>>& r, const XyzzL<C>& p) {
    if (p.inf) {
        jac_set_inf(r);
        return;
    }
    FpL<C> t;
    fpl_mul(t, p.X, p.ZZ);
    fpl_finish(r.X, t);
    fpl_mul(t, p.Y, p.ZZZ);
    fpl_finish(r.Y, t);
    fpl_finish(r.Z, p.ZZ);
}
// r += +-(x2, y2), a finite affine point in the packed, reduced form it is stored in
template <class C>
KYB_HD void xyzzl_madd(XyzzL<C>& r, const Fp<C>& x2, const Fp<C>& y2, bool neg) {
    static_assert(fpl_supported<C>(), "the limb-form accumulator needs R / p >= 2^9");
    FpL<C> lx, ly;
    fpl_unpack(lx, x2);
    fpl_unpack(ly, y2);
    if (r.inf) {  // first point of a run, or the run cancelled so far
        FpL<C> z, ny;
#pragma unroll
        for (int j = 0; j < C::N; j++) z.l[j] = 0;
        fpl_sub<1>(ny, z, ly);  // p - y2 (y2 = 0 does not occur: the curves have no point of order two)
        r.X = lx;
#pragma unroll
        for (int j = 0; j < C::N; j++) r.Y.l[j] = neg ? ny.l[j] : ly.l[j];
        fpl_one(r.ZZ);
        fpl_one(r.ZZZ);
        r.inf = 0u;
        return;
    }
    FpL<C> U2, S2, P, R, PP, PPP, Q, t, u;
    fpl_mul(U2, lx, r.ZZ);
    fpl_mul(S2, ly, r.ZZZ);
    fpl_sub<8>(P, U2, r.X);
    fpl_sub_signed<4>(R, S2, neg, r.Y);
    if (fpl_is_zero_mod_p<9>(P)) {  // same x: the point itself (double it) or its inverse (cancel)
        if (fpl_is_zero_mod_p<5>(R)) {
            Jac<Fp<C>> j;
            j.X = x2;
            j.Y = y2;
            if (neg) f_neg(j.Y, y2);
            f_one(j.Z);
            jac_dbl_inl(j, j);  // inlined: an out-of-line callee would set the register budget of the whole loop
            Fp<C> zz, zzz;
            f_sqr(zz, j.Z);
            f_mul(zzz, zz, j.Z);
            fpl_unpack(r.X, j.X);
            fpl_unpack(r.Y, j.Y);
            fpl_unpack(r.ZZ, zz);
            fpl_unpack(r.ZZZ, zzz);
        } else {
            xyzzl_set_inf(r);
        }
        return;
    }
    fpl_sqr(PP, P);
    fpl_mul(PPP, P, PP);
    fpl_mul(Q, r.X, PP);
    fpl_sqr(t, R);
    fpl_add_2x(u, PPP, Q);
    fpl_sub<6>(t, t, u);  // X3 = R^2 - (PPP + 2 Q)
    fpl_sub<8>(Q, Q, t);            // Q - X3
#pragma unroll
    for (int j = 0; j < C::N; j++) u.l[j] = 0;
    fpl_sub<2>(u, u, PPP);  // -PPP
    r.X = t;
    fpl_mul2sum(r.Y, R, Q, r.Y, u);  // Y3 = R (Q - X3) - Y1 PPP
    fpl_mul(r.ZZ, r.ZZ, PP);
    fpl_mul(r.ZZZ, r.ZZZ, PPP);
}

// XyzzSel<F>: the accumulator a RUN of mixed additions uses over field F -- the limb form where the base field has
// the headroom, the packed form otherwise (Fp2, the BN fields).  -DKYB_XYZZ_PACKED keeps the packed form everywhere
// (A/B builds).
template <class F>
struct XyzzSel {
    using type = Xyzz<F>;
    KYB_HD static void identity(type& a) { xyzz_set_inf(a); }
    KYB_HD static void madd(type& acc, const F& x, const F& py, bool neg) {
        F y = py, ny;
        f_neg(ny, py);
        f_cmov(y, ny, neg);
        xyzz_madd(acc, x, y);
    }
    KYB_HD static void finish(Jac<F>& r, const type& a) { xyzz_to_jac(r, a); }
};
#ifndef KYB_XYZZ_PACKED
template <class C>
struct XyzzSelL {
    using type = XyzzL<C>;
    KYB_HD static void identity(type& a) { xyzzl_set_inf(a); }
    KYB_HD static void madd(type& acc, const Fp<C>& x, const Fp<C>& y, bool neg) { xyzzl_madd(acc, x, y, neg); }
    KYB_HD static void finish(Jac<Fp<C>>& r, const type& a) { xyzzl_to_jac(r, a); }
};
template <class C, bool L = fpl_supported<C>()>
struct XyzzSelFp : XyzzSel<const Fp<C>> {};
template <class C>
struct XyzzSelFp<C, true> : XyzzSelL<C> {};
template <class C>
struct XyzzSel<const Fp<C>> {  // the packed form under another name (the primary template is specialised below)
    using F = Fp<C>;
    using type = Xyzz<F>;
    KYB_HD static void identity(type& a) { xyzz_set_inf(a); }
    KYB_HD static void madd(type& acc, const F& x, const F& py, bool neg) {
        F y = py, ny;
        f_neg(ny, py);
        f_cmov(y, ny, neg);
        xyzz_madd(acc, x, y);
    }
    KYB_HD static void finish(Jac<F>& r, const type& a) { xyzz_to_jac(r, a); }
};
template <class C>
struct XyzzSel<Fp<C>> : XyzzSelFp<C> {};
#endif

// Signed radix-16 digits of a 256-bit scalar given as eight little-endian words:
// e[0..63] in [-8, 8), e[64] in {0, 1}.
KYB_HD void recode16_u256(int8_t (&e)[65], const uint32_t (&k)[8]) {
    int carry = 0;
#pragma unroll
    for (int i = 0; i < 64; i++) {
        int d = (int)((k[i >> 3] >> ((i & 7) * 4)) & 15) + carry;
        carry = (d + 8) >> 4;
        d -= carry << 4;
        e[i] = (int8_t)d;
    }
    e[64] = (int8_t)carry;
}

// r = k * p, k a plain 256-bit integer (eight little-endian words).  Uniform control flow: every
// window does 4 doublings and one addition whose result is kept only when the digit is non-zero.
template <class F>
KYB_HD_NOINLINE void jac_mul_u256(Jac<F>& r, const Jac<F>& p, const uint32_t (&k)[8]) {
    Jac<F> tab[8];  // (j + 1) * p
    tab[0] = p;
    jac_dbl(tab[1], p);
#pragma unroll 1
    for (int j = 2; j < 8; j++) jac_add(tab[j], tab[j - 1], p);
    jac_table8_to_affine(tab);
    int8_t e[65];
    recode16_u256(e, k);
    Jac<F> acc;
    jac_set_inf(acc);
#pragma unroll 1
    for (int i = 64; i >= 0; i--) jac_window_step_aff(acc, tab, e[i], i != 64);
    r = acc;
}
// Short public multiplier (e.g. the curve parameter |x|), MSB-first double-and-add; uniform.
template <class F>
KYB_HD_NOINLINE void jac_mul_u64(Jac<F>& r, const Jac<F>& p, uint64_t k) {
    Jac<F> acc;
    jac_set_inf(acc);
    int run = 0;
#pragma unroll 1
    for (int i = 63; i >= 0; i--) {
        run++;
        if (((k >> i) & 1) || i == 0) {
            jac_dbl_n(acc, acc, run);
            run = 0;
            if ((k >> i) & 1) jac_add(acc, acc, p);
        }
    }
    r = acc;
}

// The same for an AFFINE base: the multiplier in non-adjacent form (a third of the digits non-zero instead of half:
// bn256's u has 29 one bits and 21 NAF digits), every addition a mixed one (7M + 4S against 11M + 5S), no doublings
// before the first digit.  The subgroup tests of the BN twists (bn_suite.inc g2_in_subgroup: [u]Q) live on this.
template <class F>
KYB_HD_NOINLINE void jac_mul_u64_aff(Jac<F>& r, const Aff<F>& p, uint64_t k) {
    uint64_t pos = 0, neg = 0;  // digit i of the form is +1 / -1; a 65th digit can only be +1 (`top`)
    bool top = false;
    {
        unsigned __int128 x = k;
#pragma unroll 1
        for (int i = 0; x != 0; i++) {
            if (x & 1) {
                if ((x & 3) == 1) {
                    if (i < 64) pos |= uint64_t(1) << i;
                    else top = true;
                    x -= 1;
                } else {
                    neg |= uint64_t(1) << i;
                    x += 1;
                }
            }
            x >>= 1;
        }
    }
    F ny;
    f_neg(ny, p.y);
    Jac<F> acc;
    jac_set_inf(acc);
    int run = 0;
    bool started = false;
#pragma unroll 1
    for (int i = 64; i >= 0; i--) {
        run++;
        const bool dp = i == 64 ? top : ((pos >> i) & 1) != 0, dn = i < 64 && ((neg >> i) & 1) != 0;
        if (dp || dn || i == 0) {
            if (started) jac_dbl_n(acc, acc, run);
            run = 0;
            if (dp || dn) {
                jac_madd(acc, acc, p.x, dp ? p.y : ny, p.inf);
                started = true;
            }
        }
    }
    r = acc;
}

template <class F>
KYB_HD_NOINLINE void jac_to_aff(Aff<F>& a, const Jac<F>& p) {
    a.inf = jac_is_inf(p);
    F zi, zi2;
    f_inv(zi, p.Z);
    f_sqr(zi2, zi);
    f_mul(a.x, p.X, zi2);
    f_mul(zi2, zi2, zi);
    f_mul(a.y, p.Y, zi2);
}
// Equality of the points represented (both finite or both infinite).
template <class F>
KYB_HD bool jac_eq(const Jac<F>& p, const Jac<F>& q) {
    const bool pinf = jac_is_inf(p), qinf = jac_is_inf(q);
    F z1, z2, a, b;
    f_sqr(z1, p.Z);
    f_sqr(z2, q.Z);
    f_mul(a, p.X, z2);
    f_mul(b, q.X, z1);
    bool ok = f_eq(a, b);
    f_mul(z1, z1, p.Z);
    f_mul(z2, z2, q.Z);
    f_mul(a, p.Y, z2);
    f_mul(b, q.Y, z1);
    ok = ok & f_eq(a, b);
    return (pinf || qinf) ? (pinf && qinf) : ok;
}

}  // namespace kyb
