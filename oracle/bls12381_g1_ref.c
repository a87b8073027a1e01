/* CPU ORACLE (test infrastructure, never linked into the product): sum_i k_i P_i on BLS12-381 G1 the way the reference's
 * MSM-shaped call sites compute it -- N x (Point.Mul + Point.Add), share/poly.go:340-348, 449-476, sign/bdn/bdn.go:126-161
 * over kilic/g1.go:90-116 -- for bench.py's cpu_baseline leg of BASELINE.json configs[2] (MSM at 2^20 points).
 *
 * The group arithmetic of the reference's BLS12-381 backends is NOT in the reference tree (github.com/kilic/bls12-381
 * v0.1.0, go.mod:8); this file is a PORT of the same textbook algorithm the in-tree bn256 suite uses
 * (pairing/bn256/curve.go:69-203: add-2007-bl, dbl-2009-l, MSB-first double-and-add) onto the published curve
 * y^2 = x^3 + 4 over the 381-bit prime, 6 x 64-bit Montgomery limbs.  Points travel in the ZCash uncompressed form
 * (x || y, 48-byte big-endian each; infinity = 0x40 then zeros), which needs no square root.
 * Held against oracle/bls12381.py by tests/test_oracle_bn256_c.py. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;
#define NL 6
typedef struct { u64 v[NL]; } fq;
typedef struct { fq x, y, z; } g1;

static const u64 Q[NL] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const u64 QINV0 = 0x89f3fffcfffcfffdull;  /* -q^-1 mod 2^64 */
static fq Q_R2, Q_ONE;
static pthread_once_t q_once = PTHREAD_ONCE_INIT;

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
static void q_init(void) {
    /* R mod q by 384 doublings of 1, R^2 mod q by 384 more */
    fq x = {{1, 0, 0, 0, 0, 0}};
    for (int i = 0; i < 384; i++) fq_add(&x, &x, &x);
    Q_ONE = x;
    for (int i = 0; i < 384; i++) fq_add(&x, &x, &x);
    Q_R2 = x;
}
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
static int fq_from_be(fq *r, const uint8_t *in, int mask_top) { /* returns 0 if the value is >= q */
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

static void g1_set_inf(g1 *c) { memset(c, 0, sizeof *c); c->y = Q_ONE; }
static void g1_double(g1 *c, const g1 *a) { /* dbl-2009-l (curve.go:156-187) */
    fq A, B, C, t, t2, d, e, f;
    g1 r;
    fq_mul(&A, &a->x, &a->x);
    fq_mul(&B, &a->y, &a->y);
    fq_mul(&C, &B, &B);
    fq_add(&t, &a->x, &B);
    fq_mul(&t2, &t, &t);
    fq_sub(&t, &t2, &A);
    fq_sub(&t2, &t, &C);
    fq_add(&d, &t2, &t2);
    fq_add(&t, &A, &A);
    fq_add(&e, &t, &A);
    fq_mul(&f, &e, &e);
    fq_add(&t, &d, &d);
    fq_sub(&r.x, &f, &t);
    fq_mul(&r.z, &a->y, &a->z);
    fq_add(&r.z, &r.z, &r.z);
    fq_add(&t, &C, &C);
    fq_add(&t2, &t, &t);
    fq_add(&t, &t2, &t2);
    fq_sub(&r.y, &d, &r.x);
    fq_mul(&t2, &e, &r.y);
    fq_sub(&r.y, &t2, &t);
    *c = r;
}
static void g1_add(g1 *c, const g1 *a, const g1 *b) { /* add-2007-bl (curve.go:69-154) */
    if (fq_is_zero(&a->z)) { *c = *b; return; }
    if (fq_is_zero(&b->z)) { *c = *a; return; }
    fq z12, z22, u1, u2, t, s1, s2, h, i, j, r, v, t4, t6;
    g1 o;
    fq_mul(&z12, &a->z, &a->z);
    fq_mul(&z22, &b->z, &b->z);
    fq_mul(&u1, &a->x, &z22);
    fq_mul(&u2, &b->x, &z12);
    fq_mul(&t, &b->z, &z22);
    fq_mul(&s1, &a->y, &t);
    fq_mul(&t, &a->z, &z12);
    fq_mul(&s2, &b->y, &t);
    fq_sub(&h, &u2, &u1);
    const int x_equal = fq_is_zero(&h);
    fq_add(&t, &h, &h);
    fq_mul(&i, &t, &t);
    fq_mul(&j, &h, &i);
    fq_sub(&t, &s2, &s1);
    if (x_equal && fq_is_zero(&t)) { g1_double(c, a); return; }
    fq_add(&r, &t, &t);
    fq_mul(&v, &u1, &i);
    fq_mul(&t4, &r, &r);
    fq_add(&t, &v, &v);
    fq_sub(&t6, &t4, &j);
    fq_sub(&o.x, &t6, &t);
    fq_sub(&t, &v, &o.x);
    fq_mul(&t4, &s1, &j);
    fq_add(&t6, &t4, &t4);
    fq_mul(&t4, &r, &t);
    fq_sub(&o.y, &t4, &t6);
    fq_add(&t, &a->z, &b->z);
    fq_mul(&t4, &t, &t);
    fq_sub(&t, &t4, &z12);
    fq_sub(&t4, &t, &z22);
    fq_mul(&o.z, &t4, &h);
    *c = o;
}
static void g1_mul(g1 *c, const g1 *a, const uint8_t *scalar_be) { /* curve.go:189-203 */
    g1 sum, t;
    g1_set_inf(&sum);
    int top = -1;
    for (int i = 0; i < 256; i++)
        if ((scalar_be[i >> 3] >> (7 - (i & 7))) & 1) { top = 255 - i; break; }
    for (int i = top + 1; i >= 0; i--) {
        g1_double(&t, &sum);
        const int bit = i <= 255 ? (scalar_be[31 - (i >> 3)] >> (i & 7)) & 1 : 0;
        if (bit) g1_add(&sum, &t, a);
        else sum = t;
    }
    *c = sum;
}
/* ZCash uncompressed G1: flags in the top three bits of the first byte (compression 0, infinity, sort 0) */
static int g1_from_unc(g1 *c, const uint8_t *in) {
    const int top = in[0] >> 5;
    if (top & 5) return 1;
    if (top & 2) {
        if (in[0] & 0x1f) return 1;
        for (int i = 1; i < 96; i++) if (in[i]) return 1;
        g1_set_inf(c);
        return 0;
    }
    if (!fq_from_be(&c->x, in, 1) || !fq_from_be(&c->y, in + 48, 0)) return 1;
    c->z = Q_ONE;
    return 0;
}
static void g1_to_unc(uint8_t *out, const g1 *c) {
    if (fq_is_zero(&c->z)) { memset(out, 0, 96); out[0] = 0x40; return; }
    fq zi, zi2, x, y;
    fq_inv(&zi, &c->z);
    fq_mul(&zi2, &zi, &zi);
    fq_mul(&x, &c->x, &zi2);
    fq_mul(&zi2, &zi2, &zi);
    fq_mul(&y, &c->y, &zi2);
    fq_to_be(out, &x);
    fq_to_be(out + 48, &y);
}
typedef struct {
    size_t lo, hi;
    const uint8_t *k, *p;
    g1 acc;
    uint8_t *status;
} qjob;
static void *q_worker(void *arg) {
    qjob *jb = arg;
    g1 p, kp;
    g1_set_inf(&jb->acc);
    for (size_t i = jb->lo; i < jb->hi; i++) {
        const int s = g1_from_unc(&p, jb->p + 96 * i);
        if (jb->status) jb->status[i] = (uint8_t)s;
        if (s) continue;
        g1_mul(&kp, &p, jb->k + 32 * i);
        g1_add(&jb->acc, &jb->acc, &kp);
    }
    return NULL;
}
/* out (96 bytes) = sum_i k_i P_i, N x (Mul + Add); points and result in the uncompressed form */
void ora_bls12381_g1_mul_sum(size_t n, const uint8_t *scalars_be, const uint8_t *points_unc, uint8_t *out, uint8_t *status, int threads) {
    pthread_once(&q_once, q_init);
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *th = malloc(sizeof(pthread_t) * threads);
    qjob *jobs = malloc(sizeof(qjob) * threads);
    for (int t = 0; t < threads; t++) {
        jobs[t].lo = n * t / threads;
        jobs[t].hi = n * (t + 1) / threads;
        jobs[t].k = scalars_be;
        jobs[t].p = points_unc;
        jobs[t].status = status;
        if (threads == 1) q_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, q_worker, &jobs[t]);
    }
    g1 acc;
    g1_set_inf(&acc);
    for (int t = 0; t < threads; t++) {
        if (threads > 1) pthread_join(th[t], NULL);
        g1_add(&acc, &acc, &jobs[t].acc);
    }
    g1_to_unc(out, &acc);
    free(th);
    free(jobs);
}
