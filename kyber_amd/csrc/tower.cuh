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
// ------------------------------------------------------------------ Fp6 / Fp12
// Only the containers remain per lane (GT encodings are decoded into them and handed to the tower machine coefficient
// by coefficient: *_pair.hip gt_unpack kernels); the arithmetic of the tower lives in the machine's generated programs
// (gen_tower_vm.py), GT exponentiation included since round 3.
template <class T>
KYB_HD void fp2_load_const(Fp2<T>& r, const uint32_t (&c)[2][T::F::NWORDS]) {
#pragma unroll
    for (int l = 0; l < T::F::NWORDS; l++) {
        r.c0.v[l] = c[0][l];
        r.c1.v[l] = c[1][l];
    }
}

}  // namespace kyb
