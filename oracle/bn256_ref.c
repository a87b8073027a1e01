/* CPU ORACLE (test infrastructure, never linked into the product): C restatement of the reference's in-tree bn256
 * pairing and of the N x (Mul + Add) sums its MSM-shaped callers run -- what bench.py's cpu_baseline leg times on the
 * GPU box's host cores beside the GPU figures (the Go reference cannot run there: no Go toolchain), and what
 * tests/test_oracle_bn256_c.py holds byte for byte against the big-integer restatement oracle/bn256.py.
 *
 * Follows, function by function (paths relative to /root/reference/pairing/bn256):
 *   gfp.go:15, gfp_generic.go:158   gfP = 4 x 64-bit Montgomery residue, R = 2^256; gfpMul here is a word-serial CIOS
 *                                   on unsigned __int128 (the reference's generic version works on 16 x 16-bit chunks,
 *                                   its amd64 version is assembler: same function, same residues)
 *   gfp2.go:79-161    gfP2 = x i + y: Mul (schoolbook, 4 products), Square (complex), MulXi (xi = i + 3), Invert
 *   gfp6.go:54-215    gfP6 = x tau^2 + y tau + z: Mul (Karatsuba, 6 products), Square, MulTau, Frobenius*, Invert
 *   gfp12.go:124-230  gfP12 = x omega + y: Mul, Square (complex), Exp (square-and-multiply), Frobenius*, Invert
 *   twist.go / optate.go:5-94    lineFunctionAdd, lineFunctionDouble on (x, y, z, t = z^2) twist points
 *   optate.go:96-115  mulLine (sparse multiplication by (a tau + b) omega + c)
 *   optate.go:126-213 miller: NAF of 6u + 2, then the Q1 / -Q2 Frobenius steps
 *   optate.go:215-264 finalExponentiation (the same chain, Exp by u three times)
 *   optate.go:266-274 optimalAte: one if either input is the point at infinity
 *   curve.go:69-203   curvePoint.Add (add-2007-bl) / Double (dbl-2009-l) / Mul (MSB-first double-and-add from BitLen())
 *   point.go:170-238, 423-499, 630-662   wire formats of G1 / G2 / GT
 *
 *   twist.go:75-205   twistPoint.Add / Double / Mul / MakeAffine (the same formulas over gfP2), point.go:405-452
 *
 * Entry points (ctypes, tests/_oracle_c.py): ora_bn256_pair (n pairings, threaded), ora_bn256_g1_mul_sum
 * (sum_i k_i P_i the reference's way: N x (Mul + Add), threaded partial sums), ora_bn256_g1_mul / ora_bn256_g2_mul
 * (element-wise pointG1 / pointG2.Mul, for whole-batch digests at BASELINE.json configs[4]'s size).  */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef struct { u64 v[4]; } fp;
typedef struct { fp x, y; } fp2;            /* x i + y */
typedef struct { fp2 x, y, z; } fp6;        /* x tau^2 + y tau + z */
typedef struct { fp6 x, y; } fp12;          /* x omega + y */
typedef struct { fp2 x, y, z, t; } twist;   /* Jacobian, t = z^2 */
typedef struct { fp x, y, z; } curve;       /* Jacobian; z = 0: infinity */

/* constants.go:20, gfp_generic.go p2 / np / r2 (np0 = the low word of np) */
static const u64 P[4] = {0x185cac6c5e089667ull, 0xee5b88d120b5b59eull, 0xaa6fecb86184dc21ull, 0x8fb501e34aa387f9ull};
static const u64 NP0 = 0x2387f9007f17daa9ull;
static const fp R2 = {{0x9c21c3ff7e444f56ull, 0x409ed151b2efb0c2ull, 0xc6dc37b80fb1651ull, 0x7c36e0e62c2380b7ull}};
static const u64 U_PARAM = 6518589491078791937ull; /* constants.go:17 */
static fp ONE;   /* R mod p */

static int fp_is_zero(const fp *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static int fp_eq(const fp *a, const fp *b) { return !memcmp(a, b, sizeof(fp)); }
static int geq_p(const u64 *a) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > P[i]) return 1;
        if (a[i] < P[i]) return 0;
    }
    return 1;
}
static void sub_p(u64 *a) {
    u64 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - P[i] - b;
        a[i] = (u64)d;
        b = (u64)(d >> 64) & 1;
    }
}
static void fp_add(fp *r, const fp *a, const fp *b) { /* gfpAdd */
    u64 c = 0, t[4];
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a->v[i] + b->v[i] + c;
        t[i] = (u64)s;
        c = (u64)(s >> 64);
    }
    if (c || geq_p(t)) sub_p(t);
    memcpy(r->v, t, sizeof t);
}
static void fp_sub(fp *r, const fp *a, const fp *b) { /* gfpSub */
    u64 bo = 0, t[4];
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->v[i] - b->v[i] - bo;
        t[i] = (u64)d;
        bo = (u64)(d >> 64) & 1;
    }
    if (bo) {
        u64 c = 0;
        for (int i = 0; i < 4; i++) {
            u128 s = (u128)t[i] + P[i] + c;
            t[i] = (u64)s;
            c = (u64)(s >> 64);
        }
    }
    memcpy(r->v, t, sizeof t);
}
static void fp_neg(fp *r, const fp *a) {
    fp z = {{0, 0, 0, 0}};
    fp_sub(r, &z, a);
}
#ifdef ORA_COUNT  /* gcc -DORA_COUNT: count gfpMul calls (how the "best known formula" multiply-add figures of bench.py were obtained) */
unsigned long long ora_fp_mul_count;
#define ORA_TICK ora_fp_mul_count++
#else
#define ORA_TICK ((void)0)
#endif
static void fp_mul(fp *r, const fp *a, const fp *b) { /* gfpMul: a b R^-1 mod p */
    ORA_TICK;
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->v[j] * b->v[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (u64)c;
        t[5] = (u64)(c >> 64);
        const u64 m = t[0] * NP0;
        c = (u128)m * P[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * P[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (u64)c;
        t[4] = t[5] + (u64)(c >> 64);
    }
    if (t[4] || geq_p(t)) sub_p(t);
    memcpy(r->v, t, 4 * sizeof(u64));
}
static void fp_pow(fp *r, const fp *a, const u64 *e, int nw) {
    fp acc = ONE, base = *a;
    for (int i = 0; i < 64 * nw; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(&acc, &acc, &base);
        fp_mul(&base, &base, &base);
    }
    *r = acc;
}
static void fp_inv(fp *r, const fp *a) { /* gfP.Invert: f^(p-2) */
    u64 e[4] = {P[0] - 2, P[1], P[2], P[3]};
    fp_pow(r, a, e, 4);
}
static void fp_from_be(fp *r, const uint8_t *in) { /* gfP.Unmarshal + montEncode (reduces mod p) */
    fp t;
    for (int i = 0; i < 4; i++) {
        u64 w = 0;
        for (int k = 0; k < 8; k++) w = (w << 8) | in[8 * (3 - i) + k];
        t.v[i] = w;
    }
    fp_mul(r, &t, &R2);
}
static void fp_to_be(uint8_t *out, const fp *a) { /* montDecode + Marshal */
    fp one = {{1, 0, 0, 0}}, t;
    fp_mul(&t, a, &one);
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 8; k++) out[8 * (3 - i) + k] = (uint8_t)(t.v[i] >> (56 - 8 * k));
}
static void fp_set_u64(fp *r, u64 x) {
    fp t = {{x, 0, 0, 0}};
    fp_mul(r, &t, &R2);
}

/* ---- gfp2.go */
static void f2_zero(fp2 *r) { memset(r, 0, sizeof *r); }
static void f2_one(fp2 *r) { memset(&r->x, 0, sizeof(fp)); r->y = ONE; }
static int f2_is_zero(const fp2 *a) { return fp_is_zero(&a->x) && fp_is_zero(&a->y); }
static void f2_add(fp2 *r, const fp2 *a, const fp2 *b) { fp_add(&r->x, &a->x, &b->x); fp_add(&r->y, &a->y, &b->y); }
static void f2_sub(fp2 *r, const fp2 *a, const fp2 *b) { fp_sub(&r->x, &a->x, &b->x); fp_sub(&r->y, &a->y, &b->y); }
static void f2_neg(fp2 *r, const fp2 *a) { fp_neg(&r->x, &a->x); fp_neg(&r->y, &a->y); }
static void f2_conj(fp2 *r, const fp2 *a) { fp_neg(&r->x, &a->x); r->y = a->y; }
static void f2_mul(fp2 *r, const fp2 *a, const fp2 *b) { /* gfp2.go:79-94 */
    fp tx, ty, t;
    fp_mul(&tx, &a->x, &b->y);
    fp_mul(&t, &b->x, &a->y);
    fp_add(&tx, &tx, &t);
    fp_mul(&ty, &a->y, &b->y);
    fp_mul(&t, &a->x, &b->x);
    fp_sub(&ty, &ty, &t);
    r->x = tx;
    r->y = ty;
}
static void f2_mul_scalar(fp2 *r, const fp2 *a, const fp *b) { fp_mul(&r->x, &a->x, b); fp_mul(&r->y, &a->y, b); }
static void f2_mul_xi(fp2 *r, const fp2 *a) { /* (x i + y)(i + 3) = (3x + y) i + (3y - x) */
    fp tx, ty;
    fp_add(&tx, &a->x, &a->x);
    fp_add(&tx, &tx, &a->x);
    fp_add(&tx, &tx, &a->y);
    fp_add(&ty, &a->y, &a->y);
    fp_add(&ty, &ty, &a->y);
    fp_sub(&ty, &ty, &a->x);
    r->x = tx;
    r->y = ty;
}
static void f2_sqr(fp2 *r, const fp2 *a) { /* complex squaring, gfp2.go:121-133 */
    fp tx, ty;
    fp_sub(&tx, &a->y, &a->x);
    fp_add(&ty, &a->x, &a->y);
    fp_mul(&ty, &tx, &ty);
    fp_mul(&tx, &a->x, &a->y);
    fp_add(&tx, &tx, &tx);
    r->x = tx;
    r->y = ty;
}
static void f2_inv(fp2 *r, const fp2 *a) { /* gfp2.go:146-161 */
    fp t1, t2, inv;
    fp_mul(&t1, &a->x, &a->x);
    fp_mul(&t2, &a->y, &a->y);
    fp_add(&t1, &t1, &t2);
    fp_inv(&inv, &t1);
    fp_neg(&t1, &a->x);
    fp_mul(&r->x, &t1, &inv);
    fp_mul(&r->y, &a->y, &inv);
}
static void f2_pow(fp2 *r, const fp2 *a, const u64 *e, int nw) {
    fp2 acc, base = *a;
    f2_one(&acc);
    for (int i = 0; i < 64 * nw; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) f2_mul(&acc, &acc, &base);
        f2_sqr(&base, &base);
    }
    *r = acc;
}

/* constants.go:27-77: powers of xi, computed once */
static fp2 XI_PM1_6, XI_PM1_3, XI_PM1_2, XI_2PM2_3;
static fp XI_P2M1_3, XI_2P2M2_3, XI_P2M1_6;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void div_small(u64 *q, const u64 *a, u64 d) {
    u128 rem = 0;
    for (int i = 3; i >= 0; i--) {
        u128 cur = (rem << 64) | a[i];
        q[i] = (u64)(cur / d);
        rem = cur % d;
    }
}
static void init_consts(void) {
    fp raw_one = {{1, 0, 0, 0}};
    fp_mul(&ONE, &raw_one, &R2);
    fp2 xi;
    fp_set_u64(&xi.x, 1);
    fp_set_u64(&xi.y, 3);
    u64 pm1[4] = {P[0] - 1, P[1], P[2], P[3]}, e[4];
    div_small(e, pm1, 6);
    f2_pow(&XI_PM1_6, &xi, e, 4);
    div_small(e, pm1, 3);
    f2_pow(&XI_PM1_3, &xi, e, 4);
    div_small(e, pm1, 2);
    f2_pow(&XI_PM1_2, &xi, e, 4);
    f2_sqr(&XI_2PM2_3, &XI_PM1_3);
    /* xi^((p^2-1)/k) = (xi^((p-1)/k))^(p+1) = the norm of xi^((p-1)/k): a^p is the conjugate */
    fp t;
    fp_mul(&XI_P2M1_3, &XI_PM1_3.x, &XI_PM1_3.x);
    fp_mul(&t, &XI_PM1_3.y, &XI_PM1_3.y);
    fp_add(&XI_P2M1_3, &XI_P2M1_3, &t);
    fp_mul(&XI_2P2M2_3, &XI_P2M1_3, &XI_P2M1_3);
    fp_mul(&XI_P2M1_6, &XI_PM1_6.x, &XI_PM1_6.x);
    fp_mul(&t, &XI_PM1_6.y, &XI_PM1_6.y);
    fp_add(&XI_P2M1_6, &XI_P2M1_6, &t);
}

/* ---- gfp6.go */
static void f6_add(fp6 *r, const fp6 *a, const fp6 *b) { f2_add(&r->x, &a->x, &b->x); f2_add(&r->y, &a->y, &b->y); f2_add(&r->z, &a->z, &b->z); }
static void f6_sub(fp6 *r, const fp6 *a, const fp6 *b) { f2_sub(&r->x, &a->x, &b->x); f2_sub(&r->y, &a->y, &b->y); f2_sub(&r->z, &a->z, &b->z); }
static void f6_neg(fp6 *r, const fp6 *a) { f2_neg(&r->x, &a->x); f2_neg(&r->y, &a->y); f2_neg(&r->z, &a->z); }
static void f6_mul(fp6 *r, const fp6 *a, const fp6 *b) { /* gfp6.go:96-124, Karatsuba */
    fp2 v0, v1, v2, t0, t1, tx, ty, tz;
    f2_mul(&v0, &a->z, &b->z);
    f2_mul(&v1, &a->y, &b->y);
    f2_mul(&v2, &a->x, &b->x);
    f2_add(&t0, &a->x, &a->y);
    f2_add(&t1, &b->x, &b->y);
    f2_mul(&tz, &t0, &t1);
    f2_sub(&tz, &tz, &v1);
    f2_sub(&tz, &tz, &v2);
    f2_mul_xi(&tz, &tz);
    f2_add(&tz, &tz, &v0);
    f2_add(&t0, &a->y, &a->z);
    f2_add(&t1, &b->y, &b->z);
    f2_mul(&ty, &t0, &t1);
    f2_mul_xi(&t0, &v2);
    f2_sub(&ty, &ty, &v0);
    f2_sub(&ty, &ty, &v1);
    f2_add(&ty, &ty, &t0);
    f2_add(&t0, &a->x, &a->z);
    f2_add(&t1, &b->x, &b->z);
    f2_mul(&tx, &t0, &t1);
    f2_sub(&tx, &tx, &v0);
    f2_add(&tx, &tx, &v1);
    f2_sub(&tx, &tx, &v2);
    r->x = tx;
    r->y = ty;
    r->z = tz;
}
static void f6_mul_scalar(fp6 *r, const fp6 *a, const fp2 *b) { f2_mul(&r->x, &a->x, b); f2_mul(&r->y, &a->y, b); f2_mul(&r->z, &a->z, b); }
static void f6_mul_gfp(fp6 *r, const fp6 *a, const fp *b) { f2_mul_scalar(&r->x, &a->x, b); f2_mul_scalar(&r->y, &a->y, b); f2_mul_scalar(&r->z, &a->z, b); }
static void f6_mul_tau(fp6 *r, const fp6 *a) { /* tau (x tau^2 + y tau + z) = y tau^2 + z tau + x xi */
    fp2 tz, ty;
    f2_mul_xi(&tz, &a->x);
    ty = a->y;
    r->y = a->z;
    r->x = ty;
    r->z = tz;
}
static void f6_sqr(fp6 *r, const fp6 *a) { /* gfp6.go:151-171 */
    fp2 v0, v1, v2, c0, c1, c2, xiv2;
    f2_sqr(&v0, &a->z);
    f2_sqr(&v1, &a->y);
    f2_sqr(&v2, &a->x);
    f2_add(&c0, &a->x, &a->y);
    f2_sqr(&c0, &c0);
    f2_sub(&c0, &c0, &v1);
    f2_sub(&c0, &c0, &v2);
    f2_mul_xi(&c0, &c0);
    f2_add(&c0, &c0, &v0);
    f2_add(&c1, &a->y, &a->z);
    f2_sqr(&c1, &c1);
    f2_sub(&c1, &c1, &v0);
    f2_sub(&c1, &c1, &v1);
    f2_mul_xi(&xiv2, &v2);
    f2_add(&c1, &c1, &xiv2);
    f2_add(&c2, &a->x, &a->z);
    f2_sqr(&c2, &c2);
    f2_sub(&c2, &c2, &v0);
    f2_add(&c2, &c2, &v1);
    f2_sub(&c2, &c2, &v2);
    r->x = c2;
    r->y = c1;
    r->z = c0;
}
static void f6_frob(fp6 *r, const fp6 *a) { /* gfp6.go:54-62 */
    f2_conj(&r->x, &a->x);
    f2_conj(&r->y, &a->y);
    f2_conj(&r->z, &a->z);
    f2_mul(&r->x, &r->x, &XI_2PM2_3);
    f2_mul(&r->y, &r->y, &XI_PM1_3);
}
static void f6_frob2(fp6 *r, const fp6 *a) {
    f2_mul_scalar(&r->x, &a->x, &XI_2P2M2_3);
    f2_mul_scalar(&r->y, &a->y, &XI_P2M1_3);
    r->z = a->z;
}
static void f6_inv(fp6 *r, const fp6 *a) { /* gfp6.go:173-215 */
    fp2 t1, A, B, C, F;
    f2_mul(&t1, &a->x, &a->y);
    f2_mul_xi(&t1, &t1);
    f2_sqr(&A, &a->z);
    f2_sub(&A, &A, &t1);
    f2_sqr(&B, &a->x);
    f2_mul_xi(&B, &B);
    f2_mul(&t1, &a->y, &a->z);
    f2_sub(&B, &B, &t1);
    f2_sqr(&C, &a->y);
    f2_mul(&t1, &a->x, &a->z);
    f2_sub(&C, &C, &t1);
    f2_mul(&F, &C, &a->y);
    f2_mul_xi(&F, &F);
    f2_mul(&t1, &A, &a->z);
    f2_add(&F, &F, &t1);
    f2_mul(&t1, &B, &a->x);
    f2_mul_xi(&t1, &t1);
    f2_add(&F, &F, &t1);
    f2_inv(&F, &F);
    f2_mul(&r->x, &C, &F);
    f2_mul(&r->y, &B, &F);
    f2_mul(&r->z, &A, &F);
}

/* ---- gfp12.go */
static void f12_one(fp12 *r) { memset(r, 0, sizeof *r); r->y.z.y = ONE; }
static void f12_conj(fp12 *r, const fp12 *a) { f6_neg(&r->x, &a->x); r->y = a->y; }
static void f12_mul(fp12 *r, const fp12 *a, const fp12 *b) { /* gfp12.go:170-181 */
    fp6 tx, t, ty;
    f6_mul(&tx, &a->x, &b->y);
    f6_mul(&t, &b->x, &a->y);
    f6_add(&tx, &tx, &t);
    f6_mul(&ty, &a->y, &b->y);
    f6_mul(&t, &a->x, &b->x);
    f6_mul_tau(&t, &t);
    r->x = tx;
    f6_add(&r->y, &ty, &t);
}
static void f12_sqr(fp12 *r, const fp12 *a) { /* gfp12.go:205-219, complex squaring */
    fp6 v0, t, ty;
    f6_mul(&v0, &a->x, &a->y);
    f6_mul_tau(&t, &a->x);
    f6_add(&t, &a->y, &t);
    f6_add(&ty, &a->x, &a->y);
    f6_mul(&ty, &ty, &t);
    f6_sub(&ty, &ty, &v0);
    f6_mul_tau(&t, &v0);
    f6_sub(&ty, &ty, &t);
    f6_add(&r->x, &v0, &v0);
    r->y = ty;
}
static void f12_exp_u64(fp12 *r, const fp12 *a, u64 power) { /* gfp12.go:188-203 */
    fp12 sum, t;
    f12_one(&sum);
    int top = 63;
    while (top > 0 && !((power >> top) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        f12_sqr(&t, &sum);
        if ((power >> i) & 1) f12_mul(&sum, &t, a);
        else sum = t;
    }
    *r = sum;
}
static void f12_frob(fp12 *r, const fp12 *a) {
    f6_frob(&r->x, &a->x);
    f6_frob(&r->y, &a->y);
    f6_mul_scalar(&r->x, &r->x, &XI_PM1_6);
}
static void f12_frob2(fp12 *r, const fp12 *a) {
    f6_frob2(&r->x, &a->x);
    f6_mul_gfp(&r->x, &r->x, &XI_P2M1_6);
    f6_frob2(&r->y, &a->y);
}
static void f12_inv(fp12 *r, const fp12 *a) { /* gfp12.go:221-234 */
    fp6 t1, t2;
    f6_sqr(&t1, &a->x);
    f6_sqr(&t2, &a->y);
    f6_mul_tau(&t1, &t1);
    f6_sub(&t1, &t2, &t1);
    f6_inv(&t2, &t1);
    f6_neg(&r->x, &a->x);
    r->y = a->y;
    f6_mul(&r->x, &r->x, &t2);
    f6_mul(&r->y, &r->y, &t2);
}

/* ---- optate.go */
static void line_add(fp2 *a, fp2 *b, fp2 *c, twist *out, const twist *r, const twist *p, const curve *q, const fp2 *r2) { /* :5-52 */
    fp2 B, D, H, I, E, J, L1, V, t, t2;
    f2_mul(&B, &p->x, &r->t);
    f2_add(&D, &p->y, &r->z);
    f2_sqr(&D, &D);
    f2_sub(&D, &D, r2);
    f2_sub(&D, &D, &r->t);
    f2_mul(&D, &D, &r->t);
    f2_sub(&H, &B, &r->x);
    f2_sqr(&I, &H);
    f2_add(&E, &I, &I);
    f2_add(&E, &E, &E);
    f2_mul(&J, &H, &E);
    f2_sub(&L1, &D, &r->y);
    f2_sub(&L1, &L1, &r->y);
    f2_mul(&V, &r->x, &E);
    f2_sqr(&out->x, &L1);
    f2_sub(&out->x, &out->x, &J);
    f2_sub(&out->x, &out->x, &V);
    f2_sub(&out->x, &out->x, &V);
    f2_add(&out->z, &r->z, &H);
    f2_sqr(&out->z, &out->z);
    f2_sub(&out->z, &out->z, &r->t);
    f2_sub(&out->z, &out->z, &I);
    f2_sub(&t, &V, &out->x);
    f2_mul(&t, &t, &L1);
    f2_mul(&t2, &r->y, &J);
    f2_add(&t2, &t2, &t2);
    f2_sub(&out->y, &t, &t2);
    f2_sqr(&out->t, &out->z);
    f2_add(&t, &p->y, &out->z);
    f2_sqr(&t, &t);
    f2_sub(&t, &t, r2);
    f2_sub(&t, &t, &out->t);
    f2_mul(&t2, &L1, &p->x);
    f2_add(&t2, &t2, &t2);
    f2_sub(a, &t2, &t);
    f2_mul_scalar(c, &out->z, &q->y);
    f2_add(c, c, c);
    f2_neg(b, &L1);
    f2_mul_scalar(b, b, &q->x);
    f2_add(b, b, b);
}
static void line_double(fp2 *a, fp2 *b, fp2 *c, twist *out, const twist *r, const curve *q) { /* :54-94 */
    fp2 A, B, C, D, E, G, t;
    f2_sqr(&A, &r->x);
    f2_sqr(&B, &r->y);
    f2_sqr(&C, &B);
    f2_add(&D, &r->x, &B);
    f2_sqr(&D, &D);
    f2_sub(&D, &D, &A);
    f2_sub(&D, &D, &C);
    f2_add(&D, &D, &D);
    f2_add(&E, &A, &A);
    f2_add(&E, &E, &A);
    f2_sqr(&G, &E);
    f2_sub(&out->x, &G, &D);
    f2_sub(&out->x, &out->x, &D);
    f2_add(&out->z, &r->y, &r->z);
    f2_sqr(&out->z, &out->z);
    f2_sub(&out->z, &out->z, &B);
    f2_sub(&out->z, &out->z, &r->t);
    f2_sub(&out->y, &D, &out->x);
    f2_mul(&out->y, &out->y, &E);
    f2_add(&t, &C, &C);
    f2_add(&t, &t, &t);
    f2_add(&t, &t, &t);
    f2_sub(&out->y, &out->y, &t);
    f2_sqr(&out->t, &out->z);
    f2_mul(&t, &E, &r->t);
    f2_add(&t, &t, &t);
    f2_neg(b, &t);
    f2_mul_scalar(b, b, &q->x);
    f2_add(a, &r->x, &E);
    f2_sqr(a, a);
    f2_sub(a, a, &A);
    f2_sub(a, a, &G);
    f2_add(&t, &B, &B);
    f2_add(&t, &t, &t);
    f2_sub(a, a, &t);
    f2_mul(c, &out->z, &r->t);
    f2_add(c, c, c);
    f2_mul_scalar(c, c, &q->y);
}
static void mul_line(fp12 *ret, const fp2 *a, const fp2 *b, const fp2 *c) { /* :96-115 */
    fp6 a2, t3, t2;
    fp2 t;
    f2_zero(&a2.x);
    a2.y = *a;
    a2.z = *b;
    f6_mul(&a2, &a2, &ret->x);
    f6_mul_scalar(&t3, &ret->y, c);
    f2_add(&t, b, c);
    f2_zero(&t2.x);
    t2.y = *a;
    t2.z = t;
    f6_add(&ret->x, &ret->x, &ret->y);
    ret->y = t3;
    f6_mul(&ret->x, &ret->x, &t2);
    f6_sub(&ret->x, &ret->x, &a2);
    f6_sub(&ret->x, &ret->x, &ret->y);
    f6_mul_tau(&a2, &a2);
    f6_add(&ret->y, &ret->y, &a2);
}
static const int8_t SIXU_PLUS_2_NAF[66] = {0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, -1, 0, 1, 0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, -1, 0, 1,
                                          0, 0, 0, 1, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, -1, 0, 0, 0, 0, 1, 0, 0, 0, 1};
/* q affine twist point (z = t = 1), p affine curve point */
static void miller(fp12 *ret, const twist *q, const curve *p) { /* :126-213 */
    f12_one(ret);
    twist minus_a = *q, r = *q, nr, q1, mq2;
    f2_neg(&minus_a.y, &q->y);
    fp2 r2, a, b, c;
    f2_sqr(&r2, &q->y);
    const int n = 66;
    for (int i = n - 1; i > 0; i--) {
        line_double(&a, &b, &c, &nr, &r, p);
        if (i != n - 1) f12_sqr(ret, ret);
        mul_line(ret, &a, &b, &c);
        r = nr;
        const int d = SIXU_PLUS_2_NAF[i - 1];
        if (d == 1) line_add(&a, &b, &c, &nr, &r, q, p, &r2);
        else if (d == -1) line_add(&a, &b, &c, &nr, &r, &minus_a, p, &r2);
        else continue;
        mul_line(ret, &a, &b, &c);
        r = nr;
    }
    f2_conj(&q1.x, &q->x);
    f2_mul(&q1.x, &q1.x, &XI_PM1_3);
    f2_conj(&q1.y, &q->y);
    f2_mul(&q1.y, &q1.y, &XI_PM1_2);
    f2_one(&q1.z);
    f2_one(&q1.t);
    f2_mul_scalar(&mq2.x, &q->x, &XI_P2M1_3);
    mq2.y = q->y;
    f2_one(&mq2.z);
    f2_one(&mq2.t);
    f2_sqr(&r2, &q1.y);
    line_add(&a, &b, &c, &nr, &r, &q1, p, &r2);
    mul_line(ret, &a, &b, &c);
    r = nr;
    f2_sqr(&r2, &mq2.y);
    line_add(&a, &b, &c, &nr, &r, &mq2, p, &r2);
    mul_line(ret, &a, &b, &c);
}
static void final_exp(fp12 *out, const fp12 *in) { /* :215-264 */
    fp12 t1, inv, t2, fp_, fp2_, fp3, fu, fu2, fu3, y3, fu2p, fu3p, y2, y0, y1, y5, y4, y6, t0;
    f12_conj(&t1, in);
    f12_inv(&inv, in);
    f12_mul(&t1, &t1, &inv);
    f12_frob2(&t2, &t1);
    f12_mul(&t1, &t1, &t2);
    f12_frob(&fp_, &t1);
    f12_frob2(&fp2_, &t1);
    f12_frob(&fp3, &fp2_);
    f12_exp_u64(&fu, &t1, U_PARAM);
    f12_exp_u64(&fu2, &fu, U_PARAM);
    f12_exp_u64(&fu3, &fu2, U_PARAM);
    f12_frob(&y3, &fu);
    f12_frob(&fu2p, &fu2);
    f12_frob(&fu3p, &fu3);
    f12_frob2(&y2, &fu2);
    f12_mul(&y0, &fp_, &fp2_);
    f12_mul(&y0, &y0, &fp3);
    f12_conj(&y1, &t1);
    f12_conj(&y5, &fu2);
    f12_conj(&y3, &y3);
    f12_mul(&y4, &fu, &fu2p);
    f12_conj(&y4, &y4);
    f12_mul(&y6, &fu3, &fu3p);
    f12_conj(&y6, &y6);
    f12_sqr(&t0, &y6);
    f12_mul(&t0, &t0, &y4);
    f12_mul(&t0, &t0, &y5);
    f12_mul(&t1, &y3, &y5);
    f12_mul(&t1, &t1, &t0);
    f12_mul(&t0, &t0, &y2);
    f12_sqr(&t1, &t1);
    f12_mul(&t1, &t1, &t0);
    f12_sqr(&t1, &t1);
    f12_mul(&t0, &t1, &y1);
    f12_mul(&t1, &t1, &y0);
    f12_sqr(&t0, &t0);
    f12_mul(out, &t0, &t1);
}

/* ---- curve.go */
static void curve_set_inf(curve *c) { memset(c, 0, sizeof *c); c->y = ONE; }
static void curve_double(curve *c, const curve *a) { /* :156-187 */
    fp A, B, C, t, t2, d, e, f;
    curve r;
    fp_mul(&A, &a->x, &a->x);
    fp_mul(&B, &a->y, &a->y);
    fp_mul(&C, &B, &B);
    fp_add(&t, &a->x, &B);
    fp_mul(&t2, &t, &t);
    fp_sub(&t, &t2, &A);
    fp_sub(&t2, &t, &C);
    fp_add(&d, &t2, &t2);
    fp_add(&t, &A, &A);
    fp_add(&e, &t, &A);
    fp_mul(&f, &e, &e);
    fp_add(&t, &d, &d);
    fp_sub(&r.x, &f, &t);
    fp_mul(&r.z, &a->y, &a->z);
    fp_add(&r.z, &r.z, &r.z);
    fp_add(&t, &C, &C);
    fp_add(&t2, &t, &t);
    fp_add(&t, &t2, &t2);
    fp_sub(&r.y, &d, &r.x);
    fp_mul(&t2, &e, &r.y);
    fp_sub(&r.y, &t2, &t);
    *c = r;
}
static void curve_add(curve *c, const curve *a, const curve *b) { /* :69-154 */
    if (fp_is_zero(&a->z)) { *c = *b; return; }
    if (fp_is_zero(&b->z)) { *c = *a; return; }
    fp z12, z22, u1, u2, t, s1, s2, h, i, j, r, v, t4, t6;
    curve o;
    fp_mul(&z12, &a->z, &a->z);
    fp_mul(&z22, &b->z, &b->z);
    fp_mul(&u1, &a->x, &z22);
    fp_mul(&u2, &b->x, &z12);
    fp_mul(&t, &b->z, &z22);
    fp_mul(&s1, &a->y, &t);
    fp_mul(&t, &a->z, &z12);
    fp_mul(&s2, &b->y, &t);
    fp_sub(&h, &u2, &u1);
    const int x_equal = fp_is_zero(&h);
    fp_add(&t, &h, &h);
    fp_mul(&i, &t, &t);
    fp_mul(&j, &h, &i);
    fp_sub(&t, &s2, &s1);
    const int y_equal = fp_is_zero(&t);
    if (x_equal && y_equal) { curve_double(c, a); return; }
    fp_add(&r, &t, &t);
    fp_mul(&v, &u1, &i);
    fp_mul(&t4, &r, &r);
    fp_add(&t, &v, &v);
    fp_sub(&t6, &t4, &j);
    fp_sub(&o.x, &t6, &t);
    fp_sub(&t, &v, &o.x);
    fp_mul(&t4, &s1, &j);
    fp_add(&t6, &t4, &t4);
    fp_mul(&t4, &r, &t);
    fp_sub(&o.y, &t4, &t6);
    fp_add(&t, &a->z, &b->z);
    fp_mul(&t4, &t, &t);
    fp_sub(&t, &t4, &z12);
    fp_sub(&t4, &t, &z22);
    fp_mul(&o.z, &t4, &h);
    *c = o;
}
/* curvePoint.Mul (:189-203): from bit BitLen() (one leading doubling of infinity) down to 0 */
static void curve_mul(curve *c, const curve *a, const uint8_t *scalar_be) {
    curve sum, t;
    curve_set_inf(&sum);
    int top = -1;
    for (int i = 0; i < 256; i++)
        if ((scalar_be[i >> 3] >> (7 - (i & 7))) & 1) { top = 255 - i; break; }
    for (int i = top + 1; i >= 0; i--) {
        curve_double(&t, &sum);
        const int bit = i <= 255 ? (scalar_be[31 - (i >> 3)] >> (i & 7)) & 1 : 0;
        if (bit) curve_add(&sum, &t, a);
        else sum = t;
    }
    *c = sum;
}
static void curve_to_bytes(uint8_t *out, const curve *c) { /* MakeAffine + point.go:170-192; infinity = 64 zero bytes */
    if (fp_is_zero(&c->z)) { memset(out, 0, 64); return; }
    fp zi, zi2, x, y;
    fp_inv(&zi, &c->z);
    fp_mul(&zi2, &zi, &zi);
    fp_mul(&x, &c->x, &zi2);
    fp_mul(&zi2, &zi2, &zi);
    fp_mul(&y, &c->y, &zi2);
    fp_to_be(out, &x);
    fp_to_be(out + 32, &y);
}
/* point.go:206-238: (0, 0) is infinity; on-curve check.  Returns 0 ok, 1 malformed. */
static int curve_from_bytes(curve *c, const uint8_t *in) {
    fp_from_be(&c->x, in);
    fp_from_be(&c->y, in + 32);
    if (fp_is_zero(&c->x) && fp_is_zero(&c->y)) { curve_set_inf(c); return 0; }
    c->z = ONE;
    fp y2, x3, three;
    fp_mul(&y2, &c->y, &c->y);
    fp_mul(&x3, &c->x, &c->x);
    fp_mul(&x3, &x3, &c->x);
    fp_set_u64(&three, 3);
    fp_add(&x3, &x3, &three);
    return fp_eq(&y2, &x3) ? 0 : 1;
}
/* point.go:466-499 (on-curve only: twist.go:49-60): x.x, x.y, y.x, y.y with gfP2{x, y} = x i + y */
static int twist_from_bytes(twist *t, int *inf, const uint8_t *in) {
    fp_from_be(&t->x.x, in);
    fp_from_be(&t->x.y, in + 32);
    fp_from_be(&t->y.x, in + 64);
    fp_from_be(&t->y.y, in + 96);
    f2_one(&t->z);
    f2_one(&t->t);
    *inf = f2_is_zero(&t->x) && f2_is_zero(&t->y);
    if (*inf) return 0;
    fp2 y2, x3, b, xi_inv, xi, three;
    f2_sqr(&y2, &t->y);
    f2_sqr(&x3, &t->x);
    f2_mul(&x3, &x3, &t->x);
    fp_set_u64(&xi.x, 1);
    fp_set_u64(&xi.y, 3);
    f2_inv(&xi_inv, &xi);
    f2_zero(&three);
    fp_set_u64(&three.y, 3);
    f2_mul(&b, &three, &xi_inv);  /* twistB = 3 / xi (twist.go:16-19) */
    f2_add(&x3, &x3, &b);
    f2_sub(&y2, &y2, &x3);
    return f2_is_zero(&y2) ? 0 : 1;
}
static void gt_to_bytes(uint8_t *out, const fp12 *a) { /* point.go:630-662: x.x.x, x.x.y, x.y.x ... y.z.y */
    const fp6 *h[2] = {&a->x, &a->y};
    int k = 0;
    for (int i = 0; i < 2; i++) {
        const fp2 *c[3] = {&h[i]->x, &h[i]->y, &h[i]->z};
        for (int j = 0; j < 3; j++) {
            fp_to_be(out + 32 * k++, &c[j]->x);
            fp_to_be(out + 32 * k++, &c[j]->y);
        }
    }
}

/* ---- batch entry points */
typedef struct {
    size_t lo, hi;
    const uint8_t *a, *b;
    uint8_t *out, *status;
} job;

static void *pair_worker(void *arg) {
    job *jb = arg;
    for (size_t i = jb->lo; i < jb->hi; i++) {
        curve p;
        twist q;
        int qinf;
        const int s1 = curve_from_bytes(&p, jb->a + 64 * i), s2 = twist_from_bytes(&q, &qinf, jb->b + 128 * i);
        if (jb->status) jb->status[i] = (uint8_t)(s1 | s2);
        if (s1 | s2) { memset(jb->out + 384 * i, 0, 384); continue; }
        fp12 f, e;
        if (fp_is_zero(&p.z) || qinf) {
            f12_one(&e);  /* optate.go:270-272 */
        } else {
            miller(&f, &q, &p);
            final_exp(&e, &f);
        }
        gt_to_bytes(jb->out + 384 * i, &e);
    }
    return NULL;
}
static void run_jobs(void *(*fn)(void *), job *tmpl, size_t n, int threads) {
    pthread_once(&g_once, init_consts);
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *th = malloc(sizeof(pthread_t) * threads);
    job *jobs = malloc(sizeof(job) * threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = *tmpl;
        jobs[t].lo = n * t / threads;
        jobs[t].hi = n * (t + 1) / threads;
        if (threads == 1) fn(&jobs[t]);
        else pthread_create(&th[t], NULL, fn, &jobs[t]);
    }
    if (threads > 1)
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}
/* gt[i] = Suite.Pair(g1[i], g2[i]) (suite.go:97): 64 + 128 bytes in, 384 bytes out; status 1 = UnmarshalBinary error */
void ora_bn256_pair(size_t n, const uint8_t *g1, const uint8_t *g2, uint8_t *gt, uint8_t *status, int threads) {
    job t = {0, 0, g1, g2, gt, status};
    run_jobs(pair_worker, &t, n, threads);
}

static void *mulsum_worker(void *arg) {
    job *jb = arg;
    curve acc, p, kp;
    curve_set_inf(&acc);
    for (size_t i = jb->lo; i < jb->hi; i++) {
        if (curve_from_bytes(&p, jb->b + 64 * i)) { if (jb->status) jb->status[i] = 1; continue; }
        if (jb->status) jb->status[i] = 0;
        curve_mul(&kp, &p, jb->a + 32 * i);   /* Point.Mul */
        curve_add(&acc, &acc, &kp);           /* Point.Add */
    }
    memcpy(jb->out, &acc, sizeof acc);
    return NULL;
}
/* out = sum_i k_i P_i as the reference's MSM-shaped call sites compute it (share/poly.go:340-348, 449-476;
 * sign/bdn/bdn.go:126-161): N x (Mul + Add), here one partial sum per thread, added up at the end. */
void ora_bn256_g1_mul_sum(size_t n, const uint8_t *scalars_be, const uint8_t *points, uint8_t *out, uint8_t *status, int threads) {
    pthread_once(&g_once, init_consts);
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    curve *parts = malloc(sizeof(curve) * threads);
    pthread_t *th = malloc(sizeof(pthread_t) * threads);
    job *jobs = malloc(sizeof(job) * threads);
    for (int t = 0; t < threads; t++) {
        job j = {n * t / threads, n * (t + 1) / threads, scalars_be, points, (uint8_t *)&parts[t], status};
        jobs[t] = j;
        if (threads == 1) mulsum_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, mulsum_worker, &jobs[t]);
    }
    curve acc;
    curve_set_inf(&acc);
    for (int t = 0; t < threads; t++) {
        if (threads > 1) pthread_join(th[t], NULL);
        curve_add(&acc, &acc, &parts[t]);
    }
    curve_to_bytes(out, &acc);
    free(parts);
    free(th);
    free(jobs);
}
/* out[i] = k_i P_i (pointG1.Mul), for element-wise comparisons */
static void *mul_worker(void *arg) {
    job *jb = arg;
    curve p, kp;
    for (size_t i = jb->lo; i < jb->hi; i++) {
        const int s = curve_from_bytes(&p, jb->b + 64 * i);
        if (jb->status) jb->status[i] = (uint8_t)s;
        if (s) { memset(jb->out + 64 * i, 0, 64); continue; }
        curve_mul(&kp, &p, jb->a + 32 * i);
        curve_to_bytes(jb->out + 64 * i, &kp);
    }
    return NULL;
}
void ora_bn256_g1_mul(size_t n, const uint8_t *scalars_be, const uint8_t *points, uint8_t *out, uint8_t *status, int threads) {
    job t = {0, 0, scalars_be, points, out, status};
    run_jobs(mul_worker, &t, n, threads);
}

/* ---- twist.go: twistPoint.Add (:75-141, add-2007-bl), Double (:143-170, dbl-2009-l), Mul (:172-185, MSB-first
 * double-and-add from BitLen()), MakeAffine (:187-205); pointG2.Mul / MarshalBinary point.go:405-452.  The t
 * coordinate is not maintained by these functions in the reference either (only by the line functions). */
static void twist_set_inf(twist *c) { memset(c, 0, sizeof *c); f2_one(&c->y); }
static void twist_double(twist *c, const twist *a) {
    fp2 A, B, C, t, t2, d, e, f;
    twist r;
    memset(&r, 0, sizeof r);
    f2_sqr(&A, &a->x);
    f2_sqr(&B, &a->y);
    f2_sqr(&C, &B);
    f2_add(&t, &a->x, &B);
    f2_sqr(&t2, &t);
    f2_sub(&t, &t2, &A);
    f2_sub(&t2, &t, &C);
    f2_add(&d, &t2, &t2);
    f2_add(&t, &A, &A);
    f2_add(&e, &t, &A);
    f2_sqr(&f, &e);
    f2_add(&t, &d, &d);
    f2_sub(&r.x, &f, &t);
    f2_mul(&r.z, &a->y, &a->z);
    f2_add(&r.z, &r.z, &r.z);
    f2_add(&t, &C, &C);
    f2_add(&t2, &t, &t);
    f2_add(&t, &t2, &t2);
    f2_sub(&r.y, &d, &r.x);
    f2_mul(&t2, &e, &r.y);
    f2_sub(&r.y, &t2, &t);
    *c = r;
}
static void twist_add(twist *c, const twist *a, const twist *b) {
    if (f2_is_zero(&a->z)) { *c = *b; return; }
    if (f2_is_zero(&b->z)) { *c = *a; return; }
    fp2 z12, z22, u1, u2, t, s1, s2, h, i, j, r, v, t4, t6;
    twist o;
    memset(&o, 0, sizeof o);
    f2_sqr(&z12, &a->z);
    f2_sqr(&z22, &b->z);
    f2_mul(&u1, &a->x, &z22);
    f2_mul(&u2, &b->x, &z12);
    f2_mul(&t, &b->z, &z22);
    f2_mul(&s1, &a->y, &t);
    f2_mul(&t, &a->z, &z12);
    f2_mul(&s2, &b->y, &t);
    f2_sub(&h, &u2, &u1);
    const int x_equal = f2_is_zero(&h);
    f2_add(&t, &h, &h);
    f2_sqr(&i, &t);
    f2_mul(&j, &h, &i);
    f2_sub(&t, &s2, &s1);
    if (x_equal && f2_is_zero(&t)) { twist_double(c, a); return; }
    f2_add(&r, &t, &t);
    f2_mul(&v, &u1, &i);
    f2_sqr(&t4, &r);
    f2_add(&t, &v, &v);
    f2_sub(&t6, &t4, &j);
    f2_sub(&o.x, &t6, &t);
    f2_sub(&t, &v, &o.x);
    f2_mul(&t4, &s1, &j);
    f2_add(&t6, &t4, &t4);
    f2_mul(&t4, &r, &t);
    f2_sub(&o.y, &t4, &t6);
    f2_add(&t, &a->z, &b->z);
    f2_sqr(&t4, &t);
    f2_sub(&t, &t4, &z12);
    f2_sub(&t4, &t, &z22);
    f2_mul(&o.z, &t4, &h);
    *c = o;
}
static void twist_mul(twist *c, const twist *a, const uint8_t *scalar_be) {
    twist sum, t;
    twist_set_inf(&sum);
    int top = -1;
    for (int i = 0; i < 256; i++)
        if ((scalar_be[i >> 3] >> (7 - (i & 7))) & 1) { top = 255 - i; break; }
    for (int i = top + 1; i >= 0; i--) {
        twist_double(&t, &sum);
        const int bit = i <= 255 ? (scalar_be[31 - (i >> 3)] >> (i & 7)) & 1 : 0;
        if (bit) twist_add(&sum, &t, a);
        else sum = t;
    }
    *c = sum;
}
/* MakeAffine + MarshalBinary (point.go:422-452): x.x, x.y, y.x, y.y big-endian; infinity = 128 zero bytes */
static void twist_to_bytes(uint8_t *out, const twist *c) {
    if (f2_is_zero(&c->z)) { memset(out, 0, 128); return; }
    fp2 zi, zi2, x, y, t;
    f2_inv(&zi, &c->z);
    f2_mul(&t, &c->y, &zi);
    f2_sqr(&zi2, &zi);
    f2_mul(&y, &t, &zi2);
    f2_mul(&x, &c->x, &zi2);
    fp_to_be(out, &x.x);
    fp_to_be(out + 32, &x.y);
    fp_to_be(out + 64, &y.x);
    fp_to_be(out + 96, &y.y);
}
/* out[i] = k_i Q_i (pointG2.Mul, point.go:405-420): 32 + 128 bytes in, 128 bytes out; status 1 = UnmarshalBinary's
 * "malformed point" (on-curve only, never the subgroup: twist.go:49-60) */
static void *g2_mul_worker(void *arg) {
    job *jb = arg;
    twist q, kq;
    int inf;
    for (size_t i = jb->lo; i < jb->hi; i++) {
        const int s = twist_from_bytes(&q, &inf, jb->b + 128 * i);
        if (jb->status) jb->status[i] = (uint8_t)s;
        if (s) { memset(jb->out + 128 * i, 0, 128); continue; }
        if (inf) twist_set_inf(&q);
        twist_mul(&kq, &q, jb->a + 32 * i);
        twist_to_bytes(jb->out + 128 * i, &kq);
    }
    return NULL;
}
void ora_bn256_g2_mul(size_t n, const uint8_t *scalars_be, const uint8_t *points, uint8_t *out, uint8_t *status, int threads) {
    job t = {0, 0, scalars_be, points, out, status};
    run_jobs(g2_mul_worker, &t, n, threads);
}
