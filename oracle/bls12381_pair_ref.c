/* CPU ORACLE (test infrastructure, never linked into the product): Suite.Pair on BLS12-381 (kilic/suite.go:70-75) for
 * bench.py's cpu_baseline leg of BASELINE.json configs[3] (batched pairings).
 *
 * The pairing arithmetic of the reference's BLS12-381 backends is NOT in the reference tree (github.com/kilic/bls12-381
 * v0.1.0, cloudflare/circl v1.6.3, consensys/gnark-crypto v0.19.2; go.mod:6-8).  This file is a PORT of the published
 * algorithm those backends implement -- optimal ate pairing, Miller loop over |x| = 0xd201000000010000 on the M-type
 * twist with projective line functions, final exponentiation as easy part + the five-exponentiation hard part with
 * Granger-Scott cyclotomic squarings -- on 6 x 64-bit Montgomery limbs, the tower Fp2 = Fp[i]/(i^2 + 1),
 * Fp6 = Fp2[v]/(v^3 - (1 + i)), Fp12 = Fp6[w]/(w^2 - v).  What it must reproduce is the VALUE and its bytes: the cube
 * of the canonical reduced pairing in kilic's 576-byte layout, which oracle/bls12381.py pins with the reference's IBE
 * interop vector (encrypt/ibe/ibe_test.go:202-245).  Held byte for byte against oracle/bls12381.py by
 * tests/test_oracle_bls12381_c.py.  Points travel in the ZCash uncompressed form (no square roots). */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;
#define NL 6
typedef struct { u64 v[NL]; } fq;
typedef struct { fq c0, c1; } fq2;
typedef struct { fq2 c0, c1, c2; } fq6;
typedef struct { fq6 c0, c1; } fq12;

static const u64 Q[NL] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const u64 QINV0 = 0x89f3fffcfffcfffdull; /* -q^-1 mod 2^64 */
#define X_ABS 0xd201000000010000ull
static fq Q_R2, Q_ONE;
static pthread_once_t pair_once = PTHREAD_ONCE_INIT;

/* ------------------------------------------------------------------ Fp */
static int fq_is_zero(const fq *a) { u64 o = 0; for (int i = 0; i < NL; i++) o |= a->v[i]; return o == 0; }
static int q_geq(const u64 *a) {
    for (int i = NL - 1; i >= 0; i--) {
        if (a[i] > Q[i]) return 1;
        if (a[i] < Q[i]) return 0;
    }
    return 1;
}
static void q_sub(u64 *a) {
    u64 b = 0;
    for (int i = 0; i < NL; i++) {
        u128 d = (u128)a[i] - Q[i] - b;
        a[i] = (u64)d;
        b = (u64)(d >> 64) & 1;
    }
}
static void fq_add(fq *r, const fq *a, const fq *b) {
    u64 c = 0, t[NL];
    for (int i = 0; i < NL; i++) {
        u128 s = (u128)a->v[i] + b->v[i] + c;
        t[i] = (u64)s;
        c = (u64)(s >> 64);
    }
    if (c || q_geq(t)) q_sub(t);
    memcpy(r->v, t, sizeof t);
}
static void fq_sub(fq *r, const fq *a, const fq *b) {
    u64 bo = 0, t[NL];
    for (int i = 0; i < NL; i++) {
        u128 d = (u128)a->v[i] - b->v[i] - bo;
        t[i] = (u64)d;
        bo = (u64)(d >> 64) & 1;
    }
    if (bo) {
        u64 c = 0;
        for (int i = 0; i < NL; i++) {
            u128 s = (u128)t[i] + Q[i] + c;
            t[i] = (u64)s;
            c = (u64)(s >> 64);
        }
    }
    memcpy(r->v, t, sizeof t);
}
static void fq_neg(fq *r, const fq *a) { fq z; memset(&z, 0, sizeof z); fq_sub(r, &z, a); }
static void fq_mul(fq *r, const fq *a, const fq *b) {
    u64 t[NL + 2];
    memset(t, 0, sizeof t);
    for (int i = 0; i < NL; i++) {
        u128 c = 0;
        for (int j = 0; j < NL; j++) {
            c += (u128)a->v[j] * b->v[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[NL];
        t[NL] = (u64)c;
        t[NL + 1] = (u64)(c >> 64);
        const u64 m = t[0] * QINV0;
        c = (u128)m * Q[0] + t[0];
        c >>= 64;
        for (int j = 1; j < NL; j++) {
            c += (u128)m * Q[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[NL];
        t[NL - 1] = (u64)c;
        t[NL] = t[NL + 1] + (u64)(c >> 64);
    }
    if (t[NL] || q_geq(t)) q_sub(t);
    memcpy(r->v, t, NL * sizeof(u64));
}
static void fq_sqr(fq *r, const fq *a) { fq_mul(r, a, a); }
static void fq_inv(fq *r, const fq *a) { /* a^(q-2) */
    u64 e[NL];
    memcpy(e, Q, sizeof e);
    e[0] -= 2;
    fq acc = Q_ONE, base = *a;
    for (int i = 0; i < 64 * NL; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) fq_mul(&acc, &acc, &base);
        fq_mul(&base, &base, &base);
    }
    *r = acc;
}
static int fq_from_be(fq *r, const uint8_t *in, int mask_top) { /* 0 if the value is >= q */
    fq t;
    for (int i = 0; i < NL; i++) {
        u64 w = 0;
        for (int k = 0; k < 8; k++) w = (w << 8) | in[8 * (NL - 1 - i) + k];
        t.v[i] = w;
    }
    if (mask_top) t.v[NL - 1] &= 0x1fffffffffffffffull;
    if (q_geq(t.v)) return 0;
    fq_mul(r, &t, &Q_R2);
    return 1;
}
static void fq_to_be(uint8_t *out, const fq *a) {
    fq one = {{1, 0, 0, 0, 0, 0}}, t;
    fq_mul(&t, a, &one);
    for (int i = 0; i < NL; i++)
        for (int k = 0; k < 8; k++) out[8 * (NL - 1 - i) + k] = (uint8_t)(t.v[i] >> (56 - 8 * k));
}

/* ------------------------------------------------------------------ Fp2 */
static void f2_add(fq2 *r, const fq2 *a, const fq2 *b) { fq_add(&r->c0, &a->c0, &b->c0); fq_add(&r->c1, &a->c1, &b->c1); }
static void f2_sub(fq2 *r, const fq2 *a, const fq2 *b) { fq_sub(&r->c0, &a->c0, &b->c0); fq_sub(&r->c1, &a->c1, &b->c1); }
static void f2_dbl(fq2 *r, const fq2 *a) { f2_add(r, a, a); }
static void f2_neg(fq2 *r, const fq2 *a) { fq_neg(&r->c0, &a->c0); fq_neg(&r->c1, &a->c1); }
static void f2_conj(fq2 *r, const fq2 *a) { r->c0 = a->c0; fq_neg(&r->c1, &a->c1); }
static int f2_is_zero(const fq2 *a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }
static void f2_mul(fq2 *r, const fq2 *a, const fq2 *b) { /* Karatsuba: 3 multiplications */
    fq t0, t1, s0, s1, t2;
    fq_mul(&t0, &a->c0, &b->c0);
    fq_mul(&t1, &a->c1, &b->c1);
    fq_add(&s0, &a->c0, &a->c1);
    fq_add(&s1, &b->c0, &b->c1);
    fq_mul(&t2, &s0, &s1);
    fq_sub(&r->c0, &t0, &t1);
    fq_sub(&t2, &t2, &t0);
    fq_sub(&r->c1, &t2, &t1);
}
static void f2_sqr(fq2 *r, const fq2 *a) { /* (a0 + a1)(a0 - a1), 2 a0 a1 */
    fq s, d, m;
    fq_add(&s, &a->c0, &a->c1);
    fq_sub(&d, &a->c0, &a->c1);
    fq_mul(&m, &a->c0, &a->c1);
    fq_mul(&r->c0, &s, &d);
    fq_add(&r->c1, &m, &m);
}
static void f2_mul_fq(fq2 *r, const fq2 *a, const fq *b) { fq_mul(&r->c0, &a->c0, b); fq_mul(&r->c1, &a->c1, b); }
static void f2_mul_xi(fq2 *r, const fq2 *a) { /* (1 + i) a */
    fq t0, t1;
    fq_sub(&t0, &a->c0, &a->c1);
    fq_add(&t1, &a->c0, &a->c1);
    r->c0 = t0;
    r->c1 = t1;
}
static void f2_inv(fq2 *r, const fq2 *a) {
    fq n, t;
    fq_sqr(&n, &a->c0);
    fq_sqr(&t, &a->c1);
    fq_add(&n, &n, &t);
    fq_inv(&n, &n);
    fq_mul(&r->c0, &a->c0, &n);
    fq_mul(&t, &a->c1, &n);
    fq_neg(&r->c1, &t);
}

/* ------------------------------------------------------------------ Fp6 = Fp2[v] / (v^3 - xi) */
static void f6_add(fq6 *r, const fq6 *a, const fq6 *b) { f2_add(&r->c0, &a->c0, &b->c0); f2_add(&r->c1, &a->c1, &b->c1); f2_add(&r->c2, &a->c2, &b->c2); }
static void f6_sub(fq6 *r, const fq6 *a, const fq6 *b) { f2_sub(&r->c0, &a->c0, &b->c0); f2_sub(&r->c1, &a->c1, &b->c1); f2_sub(&r->c2, &a->c2, &b->c2); }
static void f6_neg(fq6 *r, const fq6 *a) { f2_neg(&r->c0, &a->c0); f2_neg(&r->c1, &a->c1); f2_neg(&r->c2, &a->c2); }
static void f6_mul(fq6 *r, const fq6 *a, const fq6 *b) { /* Karatsuba: 6 Fp2 multiplications */
    fq2 v0, v1, v2, t0, t1, t2, s0, s1;
    f2_mul(&v0, &a->c0, &b->c0);
    f2_mul(&v1, &a->c1, &b->c1);
    f2_mul(&v2, &a->c2, &b->c2);
    f2_add(&s0, &a->c1, &a->c2);
    f2_add(&s1, &b->c1, &b->c2);
    f2_mul(&t0, &s0, &s1);
    f2_sub(&t0, &t0, &v1);
    f2_sub(&t0, &t0, &v2);
    f2_mul_xi(&t0, &t0);
    f2_add(&t0, &t0, &v0); /* c0 = v0 + xi ((a1 + a2)(b1 + b2) - v1 - v2) */
    f2_add(&s0, &a->c0, &a->c1);
    f2_add(&s1, &b->c0, &b->c1);
    f2_mul(&t1, &s0, &s1);
    f2_sub(&t1, &t1, &v0);
    f2_sub(&t1, &t1, &v1);
    f2_mul_xi(&s0, &v2);
    f2_add(&t1, &t1, &s0); /* c1 = (a0 + a1)(b0 + b1) - v0 - v1 + xi v2 */
    f2_add(&s0, &a->c0, &a->c2);
    f2_add(&s1, &b->c0, &b->c2);
    f2_mul(&t2, &s0, &s1);
    f2_sub(&t2, &t2, &v0);
    f2_sub(&t2, &t2, &v2);
    f2_add(&t2, &t2, &v1); /* c2 = (a0 + a2)(b0 + b2) - v0 - v2 + v1 */
    r->c0 = t0;
    r->c1 = t1;
    r->c2 = t2;
}
static void f6_sqr(fq6 *r, const fq6 *a) { f6_mul(r, a, a); }
static void f6_mul_v(fq6 *r, const fq6 *a) { /* v a */
    fq2 t;
    f2_mul_xi(&t, &a->c2);
    r->c2 = a->c1;
    r->c1 = a->c0;
    r->c0 = t;
}
static void f6_inv(fq6 *r, const fq6 *a) {
    fq2 t0, t1, t2, s, d;
    f2_sqr(&t0, &a->c0);
    f2_mul(&s, &a->c1, &a->c2);
    f2_mul_xi(&s, &s);
    f2_sub(&t0, &t0, &s); /* a0^2 - xi a1 a2 */
    f2_sqr(&t1, &a->c2);
    f2_mul_xi(&t1, &t1);
    f2_mul(&s, &a->c0, &a->c1);
    f2_sub(&t1, &t1, &s); /* xi a2^2 - a0 a1 */
    f2_sqr(&t2, &a->c1);
    f2_mul(&s, &a->c0, &a->c2);
    f2_sub(&t2, &t2, &s); /* a1^2 - a0 a2 */
    f2_mul(&d, &a->c2, &t1);
    f2_mul(&s, &a->c1, &t2);
    f2_add(&d, &d, &s);
    f2_mul_xi(&d, &d);
    f2_mul(&s, &a->c0, &t0);
    f2_add(&d, &d, &s);
    f2_inv(&d, &d);
    f2_mul(&r->c0, &t0, &d);
    f2_mul(&r->c1, &t1, &d);
    f2_mul(&r->c2, &t2, &d);
}

/* ------------------------------------------------------------------ Fp12 = Fp6[w] / (w^2 - v) */
static void f12_one(fq12 *r) { memset(r, 0, sizeof *r); r->c0.c0.c0 = Q_ONE; }
static void f12_mul(fq12 *r, const fq12 *a, const fq12 *b) { /* Karatsuba: 3 Fp6 multiplications */
    fq6 t0, t1, s0, s1, t2;
    f6_mul(&t0, &a->c0, &b->c0);
    f6_mul(&t1, &a->c1, &b->c1);
    f6_add(&s0, &a->c0, &a->c1);
    f6_add(&s1, &b->c0, &b->c1);
    f6_mul(&t2, &s0, &s1);
    f6_sub(&t2, &t2, &t0);
    f6_sub(&t2, &t2, &t1);
    f6_mul_v(&s0, &t1);
    f6_add(&r->c0, &t0, &s0);
    r->c1 = t2;
}
static void f12_sqr(fq12 *r, const fq12 *a) { /* complex squaring: 2 Fp6 multiplications */
    fq6 ab, s0, s1, t;
    f6_mul(&ab, &a->c0, &a->c1);
    f6_add(&s0, &a->c0, &a->c1);
    f6_mul_v(&t, &a->c1);
    f6_add(&s1, &a->c0, &t);
    f6_mul(&s0, &s0, &s1); /* (a0 + a1)(a0 + v a1) = a0^2 + v a1^2 + (1 + v) a0 a1 */
    f6_sub(&s0, &s0, &ab);
    f6_mul_v(&t, &ab);
    f6_sub(&r->c0, &s0, &t);
    f6_add(&r->c1, &ab, &ab);
}
static void f12_conj(fq12 *r, const fq12 *a) { r->c0 = a->c0; f6_neg(&r->c1, &a->c1); }
static void f12_inv(fq12 *r, const fq12 *a) {
    fq6 t0, t1;
    f6_sqr(&t0, &a->c0);
    f6_sqr(&t1, &a->c1);
    f6_mul_v(&t1, &t1);
    f6_sub(&t0, &t0, &t1);
    f6_inv(&t0, &t0);
    f6_mul(&r->c0, &a->c0, &t0);
    f6_mul(&t1, &a->c1, &t0);
    f6_neg(&r->c1, &t1);
}
/* (sum a_k w^k)^p = sum conj(a_k) gamma_k w^k, gamma_k = xi^(k (p - 1) / 6); in the tower a = (c0 + c1 w),
 * c_j = b_0 + b_1 v + b_2 v^2:  w^(2m) <-> c0.b_m, w^(2m+1) <-> c1.b_m */
static fq2 GAMMA[6];
static void f12_frob(fq12 *r, const fq12 *a) {
    fq2 t;
    f2_conj(&r->c0.c0, &a->c0.c0);
    f2_conj(&t, &a->c1.c0); f2_mul(&r->c1.c0, &t, &GAMMA[1]);
    f2_conj(&t, &a->c0.c1); f2_mul(&r->c0.c1, &t, &GAMMA[2]);
    f2_conj(&t, &a->c1.c1); f2_mul(&r->c1.c1, &t, &GAMMA[3]);
    f2_conj(&t, &a->c0.c2); f2_mul(&r->c0.c2, &t, &GAMMA[4]);
    f2_conj(&t, &a->c1.c2); f2_mul(&r->c1.c2, &t, &GAMMA[5]);
}
/* Granger-Scott squaring in the cyclotomic subgroup (three Fp4 squarings): with the w-basis coefficients
 * (g0..g5) = (c0.b0, c1.b0, c0.b1, c1.b1, c0.b2, c1.b2) the three Fp4 = Fp2[s]/(s^2 - xi) elements are
 * (g0, g3), (g1, g4), (g2, g5) up to the usual index bookkeeping; result h = 3 A^2-terms -/+ 2 conj.  Written in
 * the form of the published formulas (eprint 2009/565, section 3.2) on z0..z5. */
static void fp4_sqr(fq2 *r0, fq2 *r1, const fq2 *a, const fq2 *b) { /* (a + b s)^2 = (a^2 + xi b^2) + (2ab) s */
    fq2 t0, t1, t2;
    f2_sqr(&t0, a);
    f2_sqr(&t1, b);
    f2_add(&t2, a, b);
    f2_sqr(&t2, &t2);
    f2_sub(&t2, &t2, &t0);
    f2_sub(r1, &t2, &t1);
    f2_mul_xi(&t1, &t1);
    f2_add(r0, &t0, &t1);
}
static void f12_cyclo_sqr(fq12 *r, const fq12 *a) {
    /* z0 = c0.b0, z4 = c0.b1, z3 = c0.b2, z2 = c1.b0, z1 = c1.b1, z5 = c1.b2 */
    const fq2 *z0 = &a->c0.c0, *z4 = &a->c0.c1, *z3 = &a->c0.c2, *z2 = &a->c1.c0, *z1 = &a->c1.c1, *z5 = &a->c1.c2;
    fq2 t0, t1, t2, t3, t4, t5, s;
    fp4_sqr(&t0, &t1, z0, z1);
    fp4_sqr(&t2, &t3, z2, z3);
    fp4_sqr(&t4, &t5, z4, z5);
    fq12 o;
    /* z0' = 3 t0 - 2 z0 ; z1' = 3 t1 + 2 z1 */
    f2_sub(&s, &t0, z0); f2_dbl(&s, &s); f2_add(&o.c0.c0, &s, &t0);
    f2_add(&s, &t1, z1); f2_dbl(&s, &s); f2_add(&o.c1.c1, &s, &t1);
    /* z2' = 3 xi t5 + 2 z2 ; z3' = 3 t4 - 2 z3 */
    fq2 x;
    f2_mul_xi(&x, &t5);
    f2_add(&s, &x, z2); f2_dbl(&s, &s); f2_add(&o.c1.c0, &s, &x);
    f2_sub(&s, &t4, z3); f2_dbl(&s, &s); f2_add(&o.c0.c2, &s, &t4);
    /* z4' = 3 t2 - 2 z4 ; z5' = 3 t3 + 2 z5 */
    f2_sub(&s, &t2, z4); f2_dbl(&s, &s); f2_add(&o.c0.c1, &s, &t2);
    f2_add(&s, &t3, z5); f2_dbl(&s, &s); f2_add(&o.c1.c2, &s, &t3);
    *r = o;
}
static void f12_exp_x_cyclo(fq12 *r, const fq12 *a) { /* conj(a^|x|) = a^x for a in the cyclotomic subgroup (x < 0) */
    fq12 acc = *a;
    for (int i = 62; i >= 0; i--) {
        f12_cyclo_sqr(&acc, &acc);
        if ((X_ABS >> i) & 1) f12_mul(&acc, &acc, a);
    }
    f12_conj(r, &acc);
}

/* ------------------------------------------------------------------ Miller loop
 * T = (X : Y : Z) homogeneous on the twist y^2 = x^3 + 4 xi; P = (xp, yp) in G1.  Every Fp2 multiple of a line value
 * dies in the final exponentiation, so the lines are kept in the scaled forms below (the same forms the GPU's tower
 * machine uses: DESIGN.md section 4a).  A line is l0 + l2 w^2 + l3 w^3 in the w-basis, i.e. c0.b0 = l0, c0.b1 = l2
 * (w^2 = v), c1.b1 = l3 (w^3 = v w). */
typedef struct { fq2 X, Y, Z; } g2p;
/* f *= l0 + l2 w^2 + l3 w^3, i.e. (b0 + b1 w) with b0 = (l0, l2, 0), b1 = (0, l3, 0): 13 Fp2 multiplications
 * instead of the 18 of a full product (Karatsuba over w with sparse Fp6 factors) */
static void f6_mul_01(fq6 *r, const fq6 *a, const fq2 *b0, const fq2 *b1) { /* a * (b0 + b1 v) */
    fq2 v0, v1, t, s0, s1, c0, c1, c2;
    f2_mul(&v0, &a->c0, b0);
    f2_mul(&v1, &a->c1, b1);
    f2_mul(&t, &a->c2, b1);
    f2_mul_xi(&t, &t);
    f2_add(&c0, &v0, &t);           /* a0 b0 + xi a2 b1 */
    f2_add(&s0, &a->c0, &a->c1);
    f2_add(&s1, b0, b1);
    f2_mul(&c1, &s0, &s1);
    f2_sub(&c1, &c1, &v0);
    f2_sub(&c1, &c1, &v1);          /* a0 b1 + a1 b0 */
    f2_mul(&t, &a->c2, b0);
    f2_add(&c2, &v1, &t);           /* a1 b1 + a2 b0 */
    r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
static void f6_mul_1(fq6 *r, const fq6 *a, const fq2 *b1) { /* a * (b1 v) */
    fq2 t0, t1, t2;
    f2_mul(&t0, &a->c2, b1);
    f2_mul_xi(&t0, &t0);
    f2_mul(&t1, &a->c0, b1);
    f2_mul(&t2, &a->c1, b1);
    r->c0 = t0; r->c1 = t1; r->c2 = t2;
}
static void f12_mul_line(fq12 *f, const fq2 *l0, const fq2 *l2, const fq2 *l3) {
    fq6 t0, t1, t2, s;
    fq2 l23;
    f6_mul_01(&t0, &f->c0, l0, l2);       /* a0 b0 */
    f6_mul_1(&t1, &f->c1, l3);            /* a1 b1 */
    f6_add(&s, &f->c0, &f->c1);
    f2_add(&l23, l2, l3);
    f6_mul_01(&t2, &s, l0, &l23);         /* (a0 + a1)(b0 + b1) */
    f6_sub(&t2, &t2, &t0);
    f6_sub(&t2, &t2, &t1);
    f6_mul_v(&s, &t1);
    f6_add(&f->c0, &t0, &s);
    f->c1 = t2;
}
static void miller_dbl(g2p *T, fq12 *f, const fq *xp, const fq *yp) {
    fq2 B, E, XY, YZ, X2, t, u, l0, l2, l3, X3, Y3, Z3;
    f2_sqr(&B, &T->Y);                 /* B = Y^2 */
    f2_sqr(&E, &T->Z);                 /* E = 3 b' Z^2 = 12 xi Z^2 */
    f2_mul_xi(&E, &E);
    f2_dbl(&t, &E); f2_add(&E, &t, &E); f2_dbl(&E, &E); f2_dbl(&E, &E);
    f2_mul(&XY, &T->X, &T->Y);
    f2_mul(&YZ, &T->Y, &T->Z);
    f2_sqr(&X2, &T->X);
    /* tangent at P: (B - E) - 3 X^2 xp w^2 + 2 Y Z yp w^3 */
    f2_sub(&l0, &B, &E);
    f2_dbl(&t, &X2); f2_add(&t, &t, &X2);
    f2_mul_fq(&l2, &t, xp);
    f2_neg(&l2, &l2);
    f2_dbl(&t, &YZ);
    f2_mul_fq(&l3, &t, yp);
    /* X3 = 2 X Y (B - 3E), Y3 = B^2 + 3E (2B - E), Z3 = 8 B Y Z */
    f2_dbl(&t, &E); f2_add(&t, &t, &E);      /* 3E */
    f2_sub(&u, &B, &t);
    f2_mul(&X3, &XY, &u);
    f2_dbl(&X3, &X3);
    f2_dbl(&u, &B); f2_sub(&u, &u, &E);      /* 2B - E */
    f2_mul(&u, &t, &u);
    f2_sqr(&Y3, &B);
    f2_add(&Y3, &Y3, &u);
    f2_mul(&Z3, &B, &YZ);
    f2_dbl(&Z3, &Z3); f2_dbl(&Z3, &Z3); f2_dbl(&Z3, &Z3);
    T->X = X3; T->Y = Y3; T->Z = Z3;
    f12_sqr(f, f);
    f12_mul_line(f, &l0, &l2, &l3);
}
static void miller_add(g2p *T, fq12 *f, const fq2 *xq, const fq2 *yq, const fq *xp, const fq *yp) {
    fq2 th, la, C, D, E, F, G, t, u, l0, l2, l3, X3, Y3, Z3;
    f2_mul(&t, yq, &T->Z); f2_sub(&th, &T->Y, &t);   /* theta = Y - yq Z */
    f2_mul(&t, xq, &T->Z); f2_sub(&la, &T->X, &t);   /* lambda = X - xq Z */
    /* chord at P: (theta xq - lambda yq) - theta xp w^2 + lambda yp w^3 */
    f2_mul(&l0, &th, xq);
    f2_mul(&t, &la, yq);
    f2_sub(&l0, &l0, &t);
    f2_mul_fq(&l2, &th, xp);
    f2_neg(&l2, &l2);
    f2_mul_fq(&l3, &la, yp);
    f2_sqr(&C, &th);
    f2_sqr(&D, &la);
    f2_mul(&E, &la, &D);
    f2_mul(&F, &T->Z, &C);
    f2_mul(&G, &T->X, &D);
    f2_add(&t, &E, &F); f2_sub(&t, &t, &G); f2_sub(&t, &t, &G);   /* E + F - 2G */
    f2_mul(&X3, &la, &t);
    f2_dbl(&u, &G); f2_add(&u, &u, &G); f2_sub(&u, &u, &E); f2_sub(&u, &u, &F);   /* 3G - E - F */
    f2_mul(&Y3, &th, &u);
    f2_mul(&t, &E, &T->Y);
    f2_sub(&Y3, &Y3, &t);
    f2_mul(&Z3, &T->Z, &E);
    T->X = X3; T->Y = Y3; T->Z = Z3;
    f12_mul_line(f, &l0, &l2, &l3);
}
static void miller_loop(fq12 *f, const fq *xp, const fq *yp, const fq2 *xq, const fq2 *yq) {
    g2p T;
    T.X = *xq; T.Y = *yq;
    memset(&T.Z, 0, sizeof T.Z); T.Z.c0 = Q_ONE;
    f12_one(f);
    for (int i = 62; i >= 0; i--) {
        miller_dbl(&T, f, xp, yp);
        if ((X_ABS >> i) & 1) miller_add(&T, f, xq, yq, xp, yp);
    }
    f12_conj(f, f); /* x < 0 */
}
/* f^(3 (p^12 - 1) / r): easy part, then kilic's five-exponentiation chain (oracle/bls12381.py final_exp_kilic_chain) */
static void final_exp(fq12 *r, const fq12 *f) {
    fq12 t[7], a;
    f12_conj(&t[0], f);
    f12_inv(&t[1], f);
    f12_mul(&t[2], &t[0], &t[1]);
    t[1] = t[2];
    f12_frob(&a, &t[2]); f12_frob(&a, &a);
    f12_mul(&t[2], &a, &t[1]);
    f12_cyclo_sqr(&a, &t[2]); f12_conj(&t[1], &a);
    f12_exp_x_cyclo(&t[3], &t[2]);
    f12_cyclo_sqr(&t[4], &t[3]);
    f12_mul(&t[5], &t[1], &t[3]);
    f12_exp_x_cyclo(&t[1], &t[5]);
    f12_exp_x_cyclo(&t[0], &t[1]);
    f12_exp_x_cyclo(&t[6], &t[0]);
    f12_mul(&t[6], &t[6], &t[4]);
    f12_exp_x_cyclo(&t[4], &t[6]);
    f12_conj(&t[5], &t[5]);
    f12_mul(&t[4], &t[4], &t[5]);
    f12_mul(&t[4], &t[4], &t[2]);
    f12_conj(&t[5], &t[2]);
    f12_mul(&t[1], &t[1], &t[2]);
    f12_frob(&t[1], &t[1]); f12_frob(&t[1], &t[1]); f12_frob(&t[1], &t[1]);
    f12_mul(&t[6], &t[6], &t[5]);
    f12_frob(&t[6], &t[6]);
    f12_mul(&t[3], &t[3], &t[0]);
    f12_frob(&t[3], &t[3]); f12_frob(&t[3], &t[3]);
    f12_mul(&t[3], &t[3], &t[1]);
    f12_mul(&t[3], &t[3], &t[6]);
    f12_mul(r, &t[3], &t[4]);
}

static void f2_pow_words(fq2 *r, const fq2 *a, const u64 *e, int nwords) {
    fq2 acc, base = *a;
    memset(&acc, 0, sizeof acc);
    acc.c0 = Q_ONE;
    for (int i = 0; i < 64 * nwords; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) f2_mul(&acc, &acc, &base);
        f2_sqr(&base, &base);
    }
    *r = acc;
}
static void pair_init(void) {
    fq x = {{1, 0, 0, 0, 0, 0}};
    for (int i = 0; i < 384; i++) fq_add(&x, &x, &x);
    Q_ONE = x;
    for (int i = 0; i < 384; i++) fq_add(&x, &x, &x);
    Q_R2 = x;
    /* gamma_1 = xi^((p - 1) / 6), gamma_k = gamma_1^k */
    u64 e[NL], rem = 0;
    memcpy(e, Q, sizeof e);
    e[0] -= 1;
    for (int i = NL - 1; i >= 0; i--) { /* divide by 6 */
        u128 cur = ((u128)rem << 64) | e[i];
        e[i] = (u64)(cur / 6);
        rem = (u64)(cur % 6);
    }
    fq2 xi;
    xi.c0 = Q_ONE;
    xi.c1 = Q_ONE;
    memset(&GAMMA[0], 0, sizeof GAMMA[0]);
    GAMMA[0].c0 = Q_ONE;
    f2_pow_words(&GAMMA[1], &xi, e, NL);
    for (int k = 2; k < 6; k++) f2_mul(&GAMMA[k], &GAMMA[k - 1], &GAMMA[1]);
}

/* ZCash uncompressed points: G1 = x || y (96 bytes), G2 = x.c1 || x.c0 || y.c1 || y.c0 (192 bytes); infinity = 0x40
 * then zeros.  Returns 0 = finite point, 1 = infinity, 2 = bad encoding (range only: the GPU side validated them). */
static int g1_from_unc(fq *x, fq *y, const uint8_t *in) {
    if (in[0] & 0x40) return 1;
    if (in[0] & 0xa0) return 2;
    if (!fq_from_be(x, in, 0) || !fq_from_be(y, in + 48, 0)) return 2;
    return 0;
}
static int g2_from_unc(fq2 *x, fq2 *y, const uint8_t *in) {
    if (in[0] & 0x40) return 1;
    if (in[0] & 0xa0) return 2;
    if (!fq_from_be(&x->c1, in, 0) || !fq_from_be(&x->c0, in + 48, 0) || !fq_from_be(&y->c1, in + 96, 0) ||
        !fq_from_be(&y->c0, in + 144, 0))
        return 2;
    return 0;
}
/* 576 bytes: Fp12.c1 then c0; within Fp6 c2, c1, c0; within Fp2 c1, c0; big-endian (oracle gt_to_bytes) */
static void gt_to_bytes(uint8_t *out, const fq12 *a) {
    const fq6 *h[2] = {&a->c1, &a->c0};
    int k = 0;
    for (int half = 0; half < 2; half++) {
        const fq2 *c[3] = {&h[half]->c2, &h[half]->c1, &h[half]->c0};
        for (int m = 0; m < 3; m++) {
            fq_to_be(out + 96 * k, &c[m]->c1);
            fq_to_be(out + 96 * k + 48, &c[m]->c0);
            k++;
        }
    }
}
static int pair_one(uint8_t *gt, const uint8_t *g1, const uint8_t *g2) {
    fq xp, yp;
    fq2 xq, yq;
    const int s1 = g1_from_unc(&xp, &yp, g1), s2 = g2_from_unc(&xq, &yq, g2);
    fq12 f, e;
    if (s1 == 2 || s2 == 2) {
        memset(gt, 0, 576);
        return 1;
    }
    if (s1 == 1 || s2 == 1) {
        f12_one(&e);
    } else {
        miller_loop(&f, &xp, &yp, &xq, &yq);
        final_exp(&e, &f);
    }
    gt_to_bytes(gt, &e);
    return 0;
}

typedef struct { size_t lo, hi; const uint8_t *g1, *g2; uint8_t *gt, *st; } job;
static void *worker(void *p) {
    job *j = (job *)p;
    for (size_t i = j->lo; i < j->hi; i++) j->st[i] = (uint8_t)pair_one(j->gt + 576 * i, j->g1 + 96 * i, j->g2 + 192 * i);
    return NULL;
}
/* n x Suite.Pair: uncompressed G1 (96 B) and G2 (192 B) in, 576-byte GT out, status 0 / 1 */
void ora_bls12381_pair(size_t n, const uint8_t *g1, const uint8_t *g2, uint8_t *gt, uint8_t *status, int threads) {
    pthread_once(&pair_once, pair_init);
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    job *jobs = (job *)malloc(sizeof(job) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (job){n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads, g1, g2, gt, status};
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

/* ------------------------------------------------------------------------------------------------------------------
 * G1Elt.Mul / G2Elt.Mul (kilic/g1.go:110-116, g2.go:109-115) element-wise, for whole-batch digests at BASELINE.json
 * configs[3]'s size.  The backend's scalar multiplication is external (kilic v0.1.0 MulScalar); what is observable is
 * the canonical ZCash encoding of k P, so this is the textbook MSB-first double-and-add of the in-tree bn256 suite
 * (pairing/bn256/curve.go:69-203: add-2007-bl, dbl-2009-l) on y^2 = x^3 + 4 resp. y^2 = x^3 + 4 (1 + i), between
 * UnmarshalBinary's decompression (kilic/g1.go:127-131: flag rules, x < p, a square root -- WITHOUT the r-torsion test,
 * which oracle/bls12381.py holds; inputs here are multiples of the generator) and MarshalBinary's compression.
 * Held byte for byte against oracle/bls12381.py by tests/test_oracle_bls12381_c.py. */
#define DEFINE_JAC(N, F, ADD, SUB, MUL, SQR, ISZ)                                                                     \
    typedef struct { F x, y, z; } N;                                                                                  \
    static void N##_double(N *c, const N *a) {                                                                        \
        F A, B, C, t, t2, d, e, f;                                                                                    \
        N r;                                                                                                          \
        SQR(&A, &a->x); SQR(&B, &a->y); SQR(&C, &B);                                                                  \
        ADD(&t, &a->x, &B); SQR(&t2, &t); SUB(&t, &t2, &A); SUB(&t2, &t, &C); ADD(&d, &t2, &t2);                      \
        ADD(&t, &A, &A); ADD(&e, &t, &A); SQR(&f, &e);                                                                \
        ADD(&t, &d, &d); SUB(&r.x, &f, &t);                                                                           \
        MUL(&r.z, &a->y, &a->z); ADD(&r.z, &r.z, &r.z);                                                               \
        ADD(&t, &C, &C); ADD(&t2, &t, &t); ADD(&t, &t2, &t2);                                                         \
        SUB(&r.y, &d, &r.x); MUL(&t2, &e, &r.y); SUB(&r.y, &t2, &t);                                                  \
        *c = r;                                                                                                       \
    }                                                                                                                 \
    static void N##_add(N *c, const N *a, const N *b) {                                                               \
        if (ISZ(&a->z)) { *c = *b; return; }                                                                          \
        if (ISZ(&b->z)) { *c = *a; return; }                                                                          \
        F z12, z22, u1, u2, t, s1, s2, h, i, j, r, v, t4, t6;                                                         \
        N o;                                                                                                          \
        SQR(&z12, &a->z); SQR(&z22, &b->z); MUL(&u1, &a->x, &z22); MUL(&u2, &b->x, &z12);                             \
        MUL(&t, &b->z, &z22); MUL(&s1, &a->y, &t); MUL(&t, &a->z, &z12); MUL(&s2, &b->y, &t);                         \
        SUB(&h, &u2, &u1);                                                                                            \
        const int x_equal = ISZ(&h);                                                                                  \
        ADD(&t, &h, &h); SQR(&i, &t); MUL(&j, &h, &i); SUB(&t, &s2, &s1);                                             \
        if (x_equal && ISZ(&t)) { N##_double(c, a); return; }                                                         \
        ADD(&r, &t, &t); MUL(&v, &u1, &i); SQR(&t4, &r); ADD(&t, &v, &v); SUB(&t6, &t4, &j); SUB(&o.x, &t6, &t);      \
        SUB(&t, &v, &o.x); MUL(&t4, &s1, &j); ADD(&t6, &t4, &t4); MUL(&t4, &r, &t); SUB(&o.y, &t4, &t6);              \
        ADD(&t, &a->z, &b->z); SQR(&t4, &t); SUB(&t, &t4, &z12); SUB(&t4, &t, &z22); MUL(&o.z, &t4, &h);              \
        *c = o;                                                                                                       \
    }                                                                                                                 \
    static void N##_mul(N *c, const N *a, const N *inf, const uint8_t *scalar_be) {                                   \
        N sum = *inf, t;                                                                                              \
        int top = -1;                                                                                                 \
        for (int i = 0; i < 256; i++)                                                                                 \
            if ((scalar_be[i >> 3] >> (7 - (i & 7))) & 1) { top = 255 - i; break; }                                   \
        for (int i = top; i >= 0; i--) {                                                                              \
            N##_double(&t, &sum);                                                                                     \
            if ((scalar_be[31 - (i >> 3)] >> (i & 7)) & 1) N##_add(&sum, &t, a);                                      \
            else sum = t;                                                                                             \
        }                                                                                                             \
        *c = sum;                                                                                                     \
    }
DEFINE_JAC(jg1, fq, fq_add, fq_sub, fq_mul, fq_sqr, fq_is_zero)
DEFINE_JAC(jg2, fq2, f2_add, f2_sub, f2_mul, f2_sqr, f2_is_zero)

static u64 Q_HALF[NL];      /* (q - 1) / 2 */
static u64 Q_SQRT_E[NL];    /* (q + 1) / 4 */
static pthread_once_t mul_once = PTHREAD_ONCE_INIT;
static void mul_init(void) {
    pthread_once(&pair_once, pair_init);
    u64 t[NL];
    memcpy(t, Q, sizeof t);
    t[0] -= 1;
    for (int i = 0; i < NL; i++) Q_HALF[i] = (t[i] >> 1) | (i + 1 < NL ? t[i + 1] << 63 : 0);
    memcpy(t, Q, sizeof t);
    t[0] += 1;  /* q = 3 mod 4 and its low word does not overflow */
    for (int i = 0; i < NL; i++) Q_SQRT_E[i] = (t[i] >> 2) | (i + 1 < NL ? t[i + 1] << 62 : 0);
}
static void fq_pow_words(fq *r, const fq *a, const u64 *e) {
    fq acc = Q_ONE, base = *a;
    for (int i = 0; i < 64 * NL; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) fq_mul(&acc, &acc, &base);
        fq_mul(&base, &base, &base);
    }
    *r = acc;
}
static int fq_eq(const fq *a, const fq *b) { return !memcmp(a, b, sizeof(fq)); }
static int fq_sqrt(fq *r, const fq *a) { /* 1 if a is a square: r = a^((q + 1) / 4) */
    fq s, c;
    fq_pow_words(&s, a, Q_SQRT_E);
    fq_sqr(&c, &s);
    *r = s;
    return fq_eq(&c, a);
}
static int fq_larger(const fq *a) { /* the "lexicographically largest" rule of the ZCash encoding: a > (q - 1) / 2 */
    fq one = {{1, 0, 0, 0, 0, 0}}, t;
    fq_mul(&t, a, &one);
    for (int i = NL - 1; i >= 0; i--) {
        if (t.v[i] > Q_HALF[i]) return 1;
        if (t.v[i] < Q_HALF[i]) return 0;
    }
    return 0;
}
static int f2_larger(const fq2 *a) { return fq_is_zero(&a->c1) ? fq_larger(&a->c0) : fq_larger(&a->c1); }
/* square root in Fp2 = Fp[i]/(i^2 + 1) by the complex method: with n = a0^2 + a1^2 = s^2, x0^2 = (a0 + s) / 2 (or
 * (a0 - s) / 2 when that is no square), x1 = a1 / (2 x0) */
static int f2_sqrt(fq2 *r, const fq2 *a) {
    fq n, t, s, two_inv, x0, x1;
    if (fq_is_zero(&a->c1)) {
        if (fq_sqrt(&x0, &a->c0)) { r->c0 = x0; memset(&r->c1, 0, sizeof(fq)); return 1; }
        fq_neg(&t, &a->c0);
        if (!fq_sqrt(&x1, &t)) return 0;
        memset(&r->c0, 0, sizeof(fq));
        r->c1 = x1;
        return 1;
    }
    fq_sqr(&n, &a->c0);
    fq_sqr(&t, &a->c1);
    fq_add(&n, &n, &t);
    if (!fq_sqrt(&s, &n)) return 0;
    fq_add(&t, &Q_ONE, &Q_ONE);
    fq_inv(&two_inv, &t);
    fq_add(&t, &a->c0, &s);
    fq_mul(&t, &t, &two_inv);
    if (!fq_sqrt(&x0, &t)) {
        fq_sub(&t, &a->c0, &s);
        fq_mul(&t, &t, &two_inv);
        if (!fq_sqrt(&x0, &t)) return 0;
    }
    fq_add(&t, &x0, &x0);
    fq_inv(&t, &t);
    fq_mul(&x1, &a->c1, &t);
    r->c0 = x0;
    r->c1 = x1;
    fq2 c;
    f2_sqr(&c, r);
    return fq_eq(&c.c0, &a->c0) && fq_eq(&c.c1, &a->c1);
}
/* the flag rules of the compressed form (bls12381_test.go:74-186's fixtures): bit 7 set, bit 6 = infinity (then bit 5
 * and every other bit clear), bit 5 = the larger y.  Returns 0 finite, 1 infinity, 2 error. */
static int zc_flags(const uint8_t *in, size_t len, int *sign) {
    if (!(in[0] & 0x80)) return 2;
    *sign = (in[0] >> 5) & 1;
    if (in[0] & 0x40) {
        if (in[0] & 0x3f) return 2;
        for (size_t i = 1; i < len; i++) if (in[i]) return 2;
        return 1;
    }
    return 0;
}
static int jg1_decompress(jg1 *p, const uint8_t *in) {
    int sign;
    const int k = zc_flags(in, 48, &sign);
    if (k) return k;
    fq rhs, four;
    if (!fq_from_be(&p->x, in, 1)) return 2;
    fq_sqr(&rhs, &p->x);
    fq_mul(&rhs, &rhs, &p->x);
    fq_add(&four, &Q_ONE, &Q_ONE);
    fq_add(&four, &four, &four);
    fq_add(&rhs, &rhs, &four);
    if (!fq_sqrt(&p->y, &rhs)) return 2;
    if (fq_larger(&p->y) != sign) fq_neg(&p->y, &p->y);
    p->z = Q_ONE;
    return 0;
}
static int jg2_decompress(jg2 *p, const uint8_t *in) {
    int sign;
    const int k = zc_flags(in, 96, &sign);
    if (k) return k;
    fq2 rhs, b;
    if (!fq_from_be(&p->x.c1, in, 1) || !fq_from_be(&p->x.c0, in + 48, 0)) return 2;
    f2_sqr(&rhs, &p->x);
    f2_mul(&rhs, &rhs, &p->x);
    fq_add(&b.c0, &Q_ONE, &Q_ONE);
    fq_add(&b.c0, &b.c0, &b.c0);
    b.c1 = b.c0;  /* 4 (1 + i) */
    f2_add(&rhs, &rhs, &b);
    if (!f2_sqrt(&p->y, &rhs)) return 2;
    if (f2_larger(&p->y) != sign) f2_neg(&p->y, &p->y);
    memset(&p->z, 0, sizeof p->z);
    p->z.c0 = Q_ONE;
    return 0;
}
static void jg1_compress(uint8_t *out, const jg1 *c) {
    if (fq_is_zero(&c->z)) { memset(out, 0, 48); out[0] = 0xc0; return; }
    fq zi, zi2, x, y;
    fq_inv(&zi, &c->z);
    fq_sqr(&zi2, &zi);
    fq_mul(&x, &c->x, &zi2);
    fq_mul(&zi2, &zi2, &zi);
    fq_mul(&y, &c->y, &zi2);
    fq_to_be(out, &x);
    out[0] |= 0x80 | (fq_larger(&y) ? 0x20 : 0);
}
static void jg2_compress(uint8_t *out, const jg2 *c) {
    if (f2_is_zero(&c->z)) { memset(out, 0, 96); out[0] = 0xc0; return; }
    fq2 zi, zi2, x, y;
    f2_inv(&zi, &c->z);
    f2_sqr(&zi2, &zi);
    f2_mul(&x, &c->x, &zi2);
    f2_mul(&zi2, &zi2, &zi);
    f2_mul(&y, &c->y, &zi2);
    fq_to_be(out, &x.c1);
    fq_to_be(out + 48, &x.c0);
    out[0] |= 0x80 | (f2_larger(&y) ? 0x20 : 0);
}
typedef struct { size_t lo, hi; const uint8_t *k, *p; uint8_t *out, *st; int g; } mjob;
static void *mul_worker(void *arg) {
    mjob *j = (mjob *)arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        if (j->g == 1) {
            jg1 p, inf, r;
            memset(&inf, 0, sizeof inf);
            inf.y = Q_ONE;
            const int s = jg1_decompress(&p, j->p + 48 * i);
            if (j->st) j->st[i] = (uint8_t)(s == 2);
            if (s == 2) { memset(j->out + 48 * i, 0, 48); continue; }
            if (s == 1) p = inf;
            jg1_mul(&r, &p, &inf, j->k + 32 * i);
            jg1_compress(j->out + 48 * i, &r);
        } else {
            jg2 p, inf, r;
            memset(&inf, 0, sizeof inf);
            inf.y.c0 = Q_ONE;
            const int s = jg2_decompress(&p, j->p + 96 * i);
            if (j->st) j->st[i] = (uint8_t)(s == 2);
            if (s == 2) { memset(j->out + 96 * i, 0, 96); continue; }
            if (s == 1) p = inf;
            jg2_mul(&r, &p, &inf, j->k + 32 * i);
            jg2_compress(j->out + 96 * i, &r);
        }
    }
    return NULL;
}
static void mul_run(int g, size_t n, const uint8_t *k, const uint8_t *p, uint8_t *out, uint8_t *st, int threads) {
    pthread_once(&mul_once, mul_init);
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    mjob *jobs = (mjob *)malloc(sizeof(mjob) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (mjob){n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads, k, p, out, st, g};
        pthread_create(&th[t], NULL, mul_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}
/* out[i] = k_i P_i: 32-byte big-endian scalars (plain 256-bit integers), 48-byte compressed points in and out;
 * status 1 = the encoding is rejected (flags, x >= p, not on the curve) */
void ora_bls12381_g1_mul(size_t n, const uint8_t *scalars_be, const uint8_t *points, uint8_t *out, uint8_t *status, int threads) {
    mul_run(1, n, scalars_be, points, out, status, threads);
}
/* the same on G2: 96-byte compressed points */
void ora_bls12381_g2_mul(size_t n, const uint8_t *scalars_be, const uint8_t *points, uint8_t *out, uint8_t *status, int threads) {
    mul_run(2, n, scalars_be, points, out, status, threads);
}

/* sum_i k_i P_i over COMPRESSED points, the reference's way -- N x (Point.Mul + Point.Add), share/poly.go:340-348,
 * 449-476, sign/bdn/bdn.go:126-161 -- one partial sum per thread; out = the 48-byte compressed sum.  The same inputs
 * the engine's MSM takes at BASELINE.json configs[2] (80 bytes per point). */
typedef struct { size_t lo, hi; const uint8_t *k, *p; uint8_t *st; jg1 acc; } sjob;
static void *sum_worker(void *arg) {
    sjob *j = (sjob *)arg;
    jg1 inf, p, kp;
    memset(&inf, 0, sizeof inf);
    inf.y = Q_ONE;
    j->acc = inf;
    for (size_t i = j->lo; i < j->hi; i++) {
        const int s = jg1_decompress(&p, j->p + 48 * i);
        if (j->st) j->st[i] = (uint8_t)(s == 2);
        if (s) continue;
        jg1_mul(&kp, &p, &inf, j->k + 32 * i);
        jg1_add(&j->acc, &j->acc, &kp);
    }
    return NULL;
}
void ora_bls12381_g1_mul_sum_compressed(size_t n, const uint8_t *scalars_be, const uint8_t *points, uint8_t *out, uint8_t *status, int threads) {
    pthread_once(&mul_once, mul_init);
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    sjob *jobs = (sjob *)malloc(sizeof(sjob) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t].lo = n * (size_t)t / (size_t)threads;
        jobs[t].hi = n * (size_t)(t + 1) / (size_t)threads;
        jobs[t].k = scalars_be;
        jobs[t].p = points;
        jobs[t].st = status;
        pthread_create(&th[t], NULL, sum_worker, &jobs[t]);
    }
    jg1 acc;
    memset(&acc, 0, sizeof acc);
    acc.y = Q_ONE;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        jg1_add(&acc, &acc, &jobs[t].acc);
    }
    jg1_compress(out, &acc);
    free(th);
    free(jobs);
}

/* n x Suite.Pair over COMPRESSED operands (48 + 96 bytes, what UnmarshalBinary takes): decompression as above, then the
 * pairing of this file; status 1 = a rejected encoding */
static void *cpair_worker(void *arg) {
    job *j = (job *)arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        jg1 p;
        jg2 q;
        const int s1 = jg1_decompress(&p, j->g1 + 48 * i), s2 = jg2_decompress(&q, j->g2 + 96 * i);
        uint8_t *gt = j->gt + 576 * i;
        j->st[i] = (uint8_t)(s1 == 2 || s2 == 2);
        if (j->st[i]) { memset(gt, 0, 576); continue; }
        fq12 f, e;
        if (s1 == 1 || s2 == 1) {
            f12_one(&e);
        } else {
            miller_loop(&f, &p.x, &p.y, &q.x, &q.y);
            final_exp(&e, &f);
        }
        gt_to_bytes(gt, &e);
    }
    return NULL;
}
void ora_bls12381_pair_compressed(size_t n, const uint8_t *g1, const uint8_t *g2, uint8_t *gt, uint8_t *status, int threads) {
    pthread_once(&mul_once, mul_init);
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    job *jobs = (job *)malloc(sizeof(job) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (job){n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads, g1, g2, gt, status};
        pthread_create(&th[t], NULL, cpair_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}
