// Extension-field tower Fp2 / Fp6 / Fp12 over a Montgomery base field (mont.cuh), shared by the
// BLS12-381 and BN suites (G2 arithmetic over Fp2, GT exponentiation over Fp12):
//   Fp2  = Fp[i]/(i^2 + 1)            elements c0 + c1 i
//   Fp6  = Fp2[v]/(v^3 - xi)          xi = XI0 + i   (BLS12-381: 1 + i, bn256: 3 + i)
//   Fp12 = Fp6[w]/(w^2 - v)
//
// Replaces: pairing/bn256 gfP2/gfP6/gfP12 (gfp2.go:11, gfp6.go:11, gfp12.go:15 -- same tower,
// different variable names: gfP2{x,y} = x i + y, gfP6{x,y,z} = x v^2 + y v + z, gfP12{x,y} = x w + y)
// and the fp2/fp6/fp12 layers of the external BLS12-381 backends (go.mod:6-8).
//
// Lazy pre-additions (mont.cuh fp_add_nr): the sums a Karatsuba step forms only to multiply them skip the conditional
// subtraction.  Bounds, as multiples of p, of what reaches fp_mul: one level of lazy sums per tower level doubles the
// bound, so with LAZY12 (all three levels) operands are < 8p (product 64 < R/p = 2^9 for BLS12-381); bn256 has
// R/p = 2^5.8 and stops at two levels (LAZY12 = false, operands < 4p).  KFP2 = bound of an Fp2-level operand.
//
// A tower configuration T provides: typedef F (field config for mont.cuh), XI0, LAZY12.
#pragma once
#include "mont.cuh"

namespace kyb {

template <class T>
struct Fp2 {
    Fp<typename T::F> c0, c1;
};
template <class T>
struct Fp6 {
    Fp2<T> c0, c1, c2;
};
template <class T>
struct Fp12 {
    Fp6<T> c0, c1;
};

// ----------------------------------------------------------------------- Fp2
template <class T> KYB_HD void fp2_zero(Fp2<T>& r) { fp_zero(r.c0); fp_zero(r.c1); }
template <class T> KYB_HD void fp2_one(Fp2<T>& r) { fp_one(r.c0); fp_zero(r.c1); }
template <class T> KYB_HD bool fp2_is_zero(const Fp2<T>& a) { return fp_is_zero(a.c0) & fp_is_zero(a.c1); }
template <class T> KYB_HD bool fp2_eq(const Fp2<T>& a, const Fp2<T>& b) { return fp_eq(a.c0, b.c0) & fp_eq(a.c1, b.c1); }
template <class T> KYB_HD void fp2_cmov(Fp2<T>& r, const Fp2<T>& a, bool c) { fp_cmov(r.c0, a.c0, c); fp_cmov(r.c1, a.c1, c); }
template <class T> KYB_HD void fp2_add(Fp2<T>& r, const Fp2<T>& a, const Fp2<T>& b) { fp_add(r.c0, a.c0, b.c0); fp_add(r.c1, a.c1, b.c1); }
template <class T> KYB_HD void fp2_sub(Fp2<T>& r, const Fp2<T>& a, const Fp2<T>& b) { fp_sub(r.c0, a.c0, b.c0); fp_sub(r.c1, a.c1, b.c1); }
template <class T> KYB_HD void fp2_dbl(Fp2<T>& r, const Fp2<T>& a) { fp_dbl(r.c0, a.c0); fp_dbl(r.c1, a.c1); }
template <class T> KYB_HD void fp2_neg(Fp2<T>& r, const Fp2<T>& a) { fp_neg(r.c0, a.c0); fp_neg(r.c1, a.c1); }
template <class T> KYB_HD void fp2_conj(Fp2<T>& r, const Fp2<T>& a) { r.c0 = a.c0; fp_neg(r.c1, a.c1); }
template <class T> KYB_HD void fp2_add_nr(Fp2<T>& r, const Fp2<T>& a, const Fp2<T>& b) { fp_add_nr(r.c0, a.c0, b.c0); fp_add_nr(r.c1, a.c1, b.c1); }
template <class T> constexpr int kfp2() { return T::LAZY12 ? 4 : 2; }

// Karatsuba: 3 base-field multiplications
template <class T>
KYB_HD void fp2_mul(Fp2<T>& r, const Fp2<T>& a, const Fp2<T>& b) {
    Fp<typename T::F> t0, t1, t2, s0, s1;
    fp_mul(t0, a.c0, b.c0);
    fp_mul(t1, a.c1, b.c1);
    fp_add_nr(s0, a.c0, a.c1);
    fp_add_nr(s1, b.c0, b.c1);
    fp_mul(t2, s0, s1);
    fp_sub(t2, t2, t0);
    fp_sub(r.c1, t2, t1);
    fp_sub(r.c0, t0, t1);
}
// (c0 + c1)(c0 - c1), 2 c0 c1: 2 base-field multiplications
template <class T>
KYB_HD void fp2_sqr(Fp2<T>& r, const Fp2<T>& a) {
    Fp<typename T::F> s, d, m;
    fp_add_nr(s, a.c0, a.c1);
    fp_sub_nr<kfp2<T>()>(d, a.c0, a.c1);
    fp_mul(m, a.c0, a.c1);
    fp_mul(r.c0, s, d);
    fp_dbl(r.c1, m);
}
// by an element of the base field
template <class T>
KYB_HD void fp2_mul_fp(Fp2<T>& r, const Fp2<T>& a, const Fp<typename T::F>& b) {
    fp_mul(r.c0, a.c0, b);
    fp_mul(r.c1, a.c1, b);
}
// r = a * xi, xi = XI0 + i:  (XI0 a0 - a1) + (a0 + XI0 a1) i
template <class T>
KYB_HD void fp2_mul_xi(Fp2<T>& r, const Fp2<T>& a) {
    Fp<typename T::F> x0 = a.c0, x1 = a.c1, t0 = a.c0, t1 = a.c1;
#pragma unroll
    for (int k = 1; k < T::XI0; k++) {
        fp_add(t0, t0, x0);
        fp_add(t1, t1, x1);
    }
    fp_sub(r.c0, t0, x1);
    fp_add(r.c1, t1, x0);
}
template <class T>
KYB_HD_NOINLINE void fp2_inv(Fp2<T>& r, const Fp2<T>& a) {
    Fp<typename T::F> n, t;
    fp_sqr(n, a.c0);
    fp_sqr(t, a.c1);
    fp_add(n, n, t);
    fp_inv(n, n);
    fp_mul(r.c0, a.c0, n);
    fp_mul(t, a.c1, n);
    fp_neg(r.c1, t);
}

// Out-of-line copies for code that runs once or a few times per element (inversions, Frobenius maps,
// decoding): keeps the number of inlined 338-MAD multiplier bodies -- and the compile time -- down.
template <class T> KYB_HD_NOINLINE void fp2_mul_c(Fp2<T>& r, const Fp2<T>& a, const Fp2<T>& b) { fp2_mul(r, a, b); }
template <class T> KYB_HD_NOINLINE void fp2_sqr_c(Fp2<T>& r, const Fp2<T>& a) { fp2_sqr(r, a); }
// The Fp2 products inside Fp6 / Fp12 arithmetic: inlined (an Fp12 multiplication is then ~290 KB of straight-line
// code for BLS12-381), or calls to the copies above with -DKYB_OUTLINE_TOWER (curve.cuh "code size").
#ifdef KYB_OUTLINE_TOWER
template <class T> KYB_HD void fp2_mulx(Fp2<T>& r, const Fp2<T>& a, const Fp2<T>& b) { fp2_mul_c(r, a, b); }
template <class T> KYB_HD void fp2_sqrx(Fp2<T>& r, const Fp2<T>& a) { fp2_sqr_c(r, a); }
#else
template <class T> KYB_HD void fp2_mulx(Fp2<T>& r, const Fp2<T>& a, const Fp2<T>& b) { fp2_mul(r, a, b); }
template <class T> KYB_HD void fp2_sqrx(Fp2<T>& r, const Fp2<T>& a) { fp2_sqr(r, a); }
#endif

// ----------------------------------------------------------------------- Fp6
template <class T> KYB_HD void fp6_zero(Fp6<T>& r) { fp2_zero(r.c0); fp2_zero(r.c1); fp2_zero(r.c2); }
template <class T> KYB_HD void fp6_one(Fp6<T>& r) { fp2_one(r.c0); fp2_zero(r.c1); fp2_zero(r.c2); }
template <class T> KYB_HD void fp6_add(Fp6<T>& r, const Fp6<T>& a, const Fp6<T>& b) { fp2_add(r.c0, a.c0, b.c0); fp2_add(r.c1, a.c1, b.c1); fp2_add(r.c2, a.c2, b.c2); }
template <class T> KYB_HD void fp6_sub(Fp6<T>& r, const Fp6<T>& a, const Fp6<T>& b) { fp2_sub(r.c0, a.c0, b.c0); fp2_sub(r.c1, a.c1, b.c1); fp2_sub(r.c2, a.c2, b.c2); }
template <class T> KYB_HD void fp6_neg(Fp6<T>& r, const Fp6<T>& a) { fp2_neg(r.c0, a.c0); fp2_neg(r.c1, a.c1); fp2_neg(r.c2, a.c2); }
template <class T> KYB_HD bool fp6_eq(const Fp6<T>& a, const Fp6<T>& b) { return fp2_eq(a.c0, b.c0) & fp2_eq(a.c1, b.c1) & fp2_eq(a.c2, b.c2); }
// r = a * v
template <class T>
KYB_HD void fp6_mul_v(Fp6<T>& r, const Fp6<T>& a) {
    Fp2<T> t;
    fp2_mul_xi(t, a.c2);
    r.c2 = a.c1;
    r.c1 = a.c0;
    r.c0 = t;
}
// Karatsuba, 6 Fp2 multiplications
template <class T>
KYB_HD void fp6_mul(Fp6<T>& r, const Fp6<T>& a, const Fp6<T>& b) {
    Fp2<T> v0, v1, v2, s, u, t0, t1, t2;
    fp2_mulx(v0, a.c0, b.c0);
    fp2_mulx(v1, a.c1, b.c1);
    fp2_mulx(v2, a.c2, b.c2);
    fp2_add_nr(s, a.c1, a.c2);
    fp2_add_nr(u, b.c1, b.c2);
    fp2_mulx(t0, s, u);
    fp2_sub(t0, t0, v1);
    fp2_sub(t0, t0, v2);
    fp2_mul_xi(t0, t0);
    fp2_add(t0, t0, v0);  // c0
    fp2_add_nr(s, a.c0, a.c1);
    fp2_add_nr(u, b.c0, b.c1);
    fp2_mulx(t1, s, u);
    fp2_sub(t1, t1, v0);
    fp2_sub(t1, t1, v1);
    fp2_mul_xi(s, v2);
    fp2_add(t1, t1, s);  // c1
    fp2_add_nr(s, a.c0, a.c2);
    fp2_add_nr(u, b.c0, b.c2);
    fp2_mulx(t2, s, u);
    fp2_sub(t2, t2, v0);
    fp2_sub(t2, t2, v2);
    fp2_add(t2, t2, v1);  // c2
    r.c0 = t0;
    r.c1 = t1;
    r.c2 = t2;
}
template <class T>
KYB_HD void fp6_sqr(Fp6<T>& r, const Fp6<T>& a) {
    // Chung-Hasan SQR2: 2 multiplications + 3 squarings in Fp2
    Fp2<T> s0, s1, s2, s3, s4, t;
    fp2_sqrx(s0, a.c0);
    fp2_mulx(t, a.c0, a.c1);
    fp2_dbl(s1, t);
    fp2_sub(t, a.c0, a.c1);
    fp2_add(t, t, a.c2);
    fp2_sqrx(s2, t);
    fp2_mulx(t, a.c1, a.c2);
    fp2_dbl(s3, t);
    fp2_sqrx(s4, a.c2);
    // c0 = s0 + xi s3 ; c1 = s1 + xi s4 ; c2 = s1 + s2 + s3 - s0 - s4
    fp2_mul_xi(t, s3);
    fp2_add(r.c0, s0, t);
    fp2_mul_xi(t, s4);
    fp2_add(r.c1, s1, t);
    fp2_add(t, s1, s2);
    fp2_add(t, t, s3);
    fp2_sub(t, t, s0);
    fp2_sub(r.c2, t, s4);
}

// Fp6 sum feeding an Fp6 multiplication: lazy where the field has the headroom for a third level
template <class T>
KYB_HD void fp6_add_pre(Fp6<T>& r, const Fp6<T>& a, const Fp6<T>& b) {
    if constexpr (T::LAZY12) {
        fp2_add_nr(r.c0, a.c0, b.c0);
        fp2_add_nr(r.c1, a.c1, b.c1);
        fp2_add_nr(r.c2, a.c2, b.c2);
    } else {
        fp6_add(r, a, b);
    }
}
template <class T> KYB_HD_NOINLINE void fp6_mul_c(Fp6<T>& r, const Fp6<T>& a, const Fp6<T>& b) { fp6_mul(r, a, b); }
template <class T> KYB_HD_NOINLINE void fp6_sqr_c(Fp6<T>& r, const Fp6<T>& a) { fp6_sqr(r, a); }

// ---------------------------------------------------------------------- Fp12
// (what GT exponentiation needs; Pair / ValidatePairing run on the tower machine, tower_vm.cuh, which expands the
//  tower into base-field bilinear forms in its generator and shares nothing with this file)
template <class T> KYB_HD void fp12_one(Fp12<T>& r) { fp6_one(r.c0); fp6_zero(r.c1); }
template <class T> KYB_HD bool fp12_eq(const Fp12<T>& a, const Fp12<T>& b) { return fp6_eq(a.c0, b.c0) & fp6_eq(a.c1, b.c1); }
template <class T>
KYB_HD bool fp12_is_one(const Fp12<T>& a) {
    Fp12<T> o;
    fp12_one(o);
    return fp12_eq(a, o);
}
template <class T>
KYB_HD_NOINLINE void fp12_mul(Fp12<T>& r, const Fp12<T>& a, const Fp12<T>& b) {
    Fp6<T> v0, v1, s, u, t;
    fp6_mul(v0, a.c0, b.c0);
    fp6_mul(v1, a.c1, b.c1);
    fp6_add_pre(s, a.c0, a.c1);
    fp6_add_pre(u, b.c0, b.c1);
    fp6_mul(t, s, u);
    fp6_sub(t, t, v0);
    fp6_sub(r.c1, t, v1);
    fp6_mul_v(t, v1);
    fp6_add(r.c0, v0, t);
}
template <class T>
KYB_HD void fp12_sqr_inl(Fp12<T>& r, const Fp12<T>& a) {
    // complex squaring: 2 Fp6 multiplications
    Fp6<T> ab, s, u, t;
    fp6_mul(ab, a.c0, a.c1);
    fp6_add_pre(s, a.c0, a.c1);
    fp6_mul_v(t, a.c1);
    fp6_add_pre(u, a.c0, t);
    fp6_mul(s, s, u);  // (a0 + a1)(a0 + v a1) = a0^2 + v a1^2 + (1 + v) a0 a1
    fp6_sub(s, s, ab);
    fp6_mul_v(t, ab);
    fp6_sub(r.c0, s, t);
    fp6_add(r.c1, ab, ab);
}
template <class T>
KYB_HD_NOINLINE void fp12_sqr(Fp12<T>& r, const Fp12<T>& a) {
    fp12_sqr_inl(r, a);
}

template <class T>
KYB_HD void fp2_load_const(Fp2<T>& r, const uint32_t (&c)[2][T::F::NWORDS]) {
#pragma unroll
    for (int l = 0; l < T::F::NWORDS; l++) {
        r.c0.v[l] = c[0][l];
        r.c1.v[l] = c[1][l];
    }
}

}  // namespace kyb
