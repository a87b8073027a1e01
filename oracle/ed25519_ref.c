/*
 * CPU ORACLE (test infrastructure + cpu_baseline "port"; NOT product code).
 *
 * A plain-C restatement of the reference's Ed25519 hot path, following the
 * algorithm structure of /root/reference/group/edwards25519:
 *   ge.go:373-417  geScalarMultBase  (signed radix-16, 32x8 precomputed table,
 *                                     odd digits, 4 doublings, even digits)
 *   ge.go:443-502  geScalarMult      (signed radix-16, 8-entry cached table,
 *                                     63 x {4 doublings + 1 addition})
 *   ge_mult_vartime.go:11-73         (all-256-bit multiplier semantics)
 *   ge.go:99-150   ToBytes / FromBytes
 * Field arithmetic is radix-2^51 with unsigned __int128 (the reference uses
 * radix-2^25.5 int32 limbs, fe.go:16-20); only canonical encodings are
 * observable so the limb choice is free.  The base table is computed at init
 * instead of being embedded (the reference embeds it, const.go:102).
 *
 * Validated against oracle/ed25519.py (itself pinned by the reference's golden
 * vectors) in tests/test_oracle_ed25519_c.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t fe[5];

#define MASK51 ((1ULL << 51) - 1)

static void fe_0(fe h) { memset(h, 0, sizeof(fe)); }
static void fe_1(fe h) { fe_0(h); h[0] = 1; }
static void fe_copy(fe h, const fe f) { memcpy(h, f, sizeof(fe)); }
static void fe_add(fe h, const fe f, const fe g) { for (int i = 0; i < 5; i++) h[i] = f[i] + g[i]; }
/* h = f - g + 4p (limbs of g must be < 2^53) */
static void fe_sub(fe h, const fe f, const fe g) {
    h[0] = f[0] + 0x1FFFFFFFFFFFB4ULL - g[0];
    for (int i = 1; i < 5; i++) h[i] = f[i] + 0x1FFFFFFFFFFFFCULL - g[i];
}
static void fe_carry(fe h) {
    uint64_t c;
    c = h[0] >> 51; h[0] &= MASK51; h[1] += c;
    c = h[1] >> 51; h[1] &= MASK51; h[2] += c;
    c = h[2] >> 51; h[2] &= MASK51; h[3] += c;
    c = h[3] >> 51; h[3] &= MASK51; h[4] += c;
    c = h[4] >> 51; h[4] &= MASK51; h[0] += 19 * c;
    c = h[0] >> 51; h[0] &= MASK51; h[1] += c;
}
static void fe_neg(fe h, const fe f) { fe z; fe_0(z); fe_sub(h, z, f); fe_carry(h); }

static void fe_mul(fe h, const fe f, const fe g) {
    u128 t0, t1, t2, t3, t4;
    uint64_t f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3], f4 = f[4];
    uint64_t g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4];
    uint64_t g1_19 = 19 * g1, g2_19 = 19 * g2, g3_19 = 19 * g3, g4_19 = 19 * g4;
    t0 = (u128)f0 * g0 + (u128)f1 * g4_19 + (u128)f2 * g3_19 + (u128)f3 * g2_19 + (u128)f4 * g1_19;
    t1 = (u128)f0 * g1 + (u128)f1 * g0 + (u128)f2 * g4_19 + (u128)f3 * g3_19 + (u128)f4 * g2_19;
    t2 = (u128)f0 * g2 + (u128)f1 * g1 + (u128)f2 * g0 + (u128)f3 * g4_19 + (u128)f4 * g3_19;
    t3 = (u128)f0 * g3 + (u128)f1 * g2 + (u128)f2 * g1 + (u128)f3 * g0 + (u128)f4 * g4_19;
    t4 = (u128)f0 * g4 + (u128)f1 * g3 + (u128)f2 * g2 + (u128)f3 * g1 + (u128)f4 * g0;
    uint64_t c;
    t1 += (uint64_t)(t0 >> 51); h[0] = (uint64_t)t0 & MASK51;
    t2 += (uint64_t)(t1 >> 51); h[1] = (uint64_t)t1 & MASK51;
    t3 += (uint64_t)(t2 >> 51); h[2] = (uint64_t)t2 & MASK51;
    t4 += (uint64_t)(t3 >> 51); h[3] = (uint64_t)t3 & MASK51;
    c = (uint64_t)(t4 >> 51);  h[4] = (uint64_t)t4 & MASK51;
    h[0] += c * 19;
    c = h[0] >> 51; h[0] &= MASK51; h[1] += c;
}
static void fe_sq(fe h, const fe f) { fe_mul(h, f, f); }
static void fe_sqn(fe h, const fe f, int n) { fe_sq(h, f); for (int i = 1; i < n; i++) fe_sq(h, h); }

/* z^(2^250 - 1) helper shared by invert and pow22523 (fe.go:906-1010 chain shape) */
static void fe_pow_2_250_1(fe out, fe z11_out, const fe z) {
    fe z2, z9, z11, t, z2_5_0, z2_10_0, z2_20_0, z2_50_0, z2_100_0;
    fe_sq(z2, z);
    fe_sqn(t, z2, 2);
    fe_mul(z9, t, z);
    fe_mul(z11, z9, z2);
    fe_sq(t, z11);
    fe_mul(z2_5_0, t, z9);
    fe_sqn(t, z2_5_0, 5);   fe_mul(z2_10_0, t, z2_5_0);
    fe_sqn(t, z2_10_0, 10); fe_mul(z2_20_0, t, z2_10_0);
    fe_sqn(t, z2_20_0, 20); fe_mul(t, t, z2_20_0);
    fe_sqn(t, t, 10);       fe_mul(z2_50_0, t, z2_10_0);
    fe_sqn(t, z2_50_0, 50); fe_mul(z2_100_0, t, z2_50_0);
    fe_sqn(t, z2_100_0, 100); fe_mul(t, t, z2_100_0);
    fe_sqn(t, t, 50);       fe_mul(out, t, z2_50_0);
    if (z11_out) fe_copy(z11_out, z11);
}
static void fe_invert(fe out, const fe z) { /* z^(p-2) = z^(2^255-21) */
    fe t, z11;
    fe_pow_2_250_1(t, z11, z);
    fe_sqn(t, t, 5);
    fe_mul(out, t, z11);
}
static void fe_pow22523(fe out, const fe z) { /* z^((p-5)/8) = z^(2^252-3) */
    fe t;
    fe_pow_2_250_1(t, NULL, z);
    fe_sqn(t, t, 2);
    fe_mul(out, t, z);
}
static void fe_tobytes(uint8_t s[32], const fe f) {
    fe h; fe_copy(h, f); fe_carry(h); fe_carry(h);
    /* h < 2^255 + small; compute h mod p canonically: q = (h + 19) >> 255 */
    uint64_t q = (h[0] + 19) >> 51;
    q = (h[1] + q) >> 51; q = (h[2] + q) >> 51; q = (h[3] + q) >> 51; q = (h[4] + q) >> 51;
    h[0] += 19 * q;
    uint64_t c;
    c = h[0] >> 51; h[0] &= MASK51; h[1] += c;
    c = h[1] >> 51; h[1] &= MASK51; h[2] += c;
    c = h[2] >> 51; h[2] &= MASK51; h[3] += c;
    c = h[3] >> 51; h[3] &= MASK51; h[4] += c;
    h[4] &= MASK51;
    uint64_t w0 = h[0] | (h[1] << 51);
    uint64_t w1 = (h[1] >> 13) | (h[2] << 38);
    uint64_t w2 = (h[2] >> 26) | (h[3] << 25);
    uint64_t w3 = (h[3] >> 39) | (h[4] << 12);
    memcpy(s, &w0, 8); memcpy(s + 8, &w1, 8); memcpy(s + 16, &w2, 8); memcpy(s + 24, &w3, 8);
}
static void fe_frombytes(fe h, const uint8_t s[32]) { /* bit 255 ignored (fe.go:91) */
    uint64_t w0, w1, w2, w3;
    memcpy(&w0, s, 8); memcpy(&w1, s + 8, 8); memcpy(&w2, s + 16, 8); memcpy(&w3, s + 24, 8);
    h[0] = w0 & MASK51;
    h[1] = ((w0 >> 51) | (w1 << 13)) & MASK51;
    h[2] = ((w1 >> 38) | (w2 << 26)) & MASK51;
    h[3] = ((w2 >> 25) | (w3 << 39)) & MASK51;
    h[4] = (w3 >> 12) & MASK51;
}
static int fe_isnegative(const fe f) { uint8_t s[32]; fe_tobytes(s, f); return s[0] & 1; }
static int fe_isnonzero(const fe f) {
    uint8_t s[32]; fe_tobytes(s, f); uint8_t r = 0;
    for (int i = 0; i < 32; i++) r |= s[i];
    return r != 0;
}
static void fe_cmov(fe f, const fe g, uint64_t b) {
    uint64_t m = (uint64_t)0 - b;
    for (int i = 0; i < 5; i++) f[i] ^= m & (f[i] ^ g[i]);
}

/* constants computed at init */
static fe FE_D, FE_D2, FE_SQRTM1;
typedef struct { fe X, Y, Z, T; } ge_p3;          /* extended (ge.go:22-24) */
typedef struct { fe X, Y, Z; } ge_p2;             /* projective */
typedef struct { fe X, Y, Z, T; } ge_p1p1;        /* completed */
typedef struct { fe YpX, YmX, Z, T2d; } ge_cached;
typedef struct { fe ypx, ymx, xy2d; } ge_precomp;
static ge_precomp BASE_TAB[32][8];
static ge_p3 GE_BASE;

static void p3_0(ge_p3 *h) { fe_0(h->X); fe_1(h->Y); fe_1(h->Z); fe_0(h->T); }
static void p1p1_to_p2(ge_p2 *r, const ge_p1p1 *p) {
    fe_mul(r->X, p->X, p->T); fe_mul(r->Y, p->Y, p->Z); fe_mul(r->Z, p->Z, p->T);
}
static void p1p1_to_p3(ge_p3 *r, const ge_p1p1 *p) {
    fe_mul(r->X, p->X, p->T); fe_mul(r->Y, p->Y, p->Z); fe_mul(r->Z, p->Z, p->T); fe_mul(r->T, p->X, p->Y);
}
static void p3_to_cached(ge_cached *r, const ge_p3 *p) {
    fe_add(r->YpX, p->Y, p->X); fe_sub(r->YmX, p->Y, p->X); fe_carry(r->YmX);
    fe_copy(r->Z, p->Z); fe_mul(r->T2d, p->T, FE_D2);
}
/* ge.go:42-60 */
static void p2_dbl(ge_p1p1 *r, const fe X, const fe Y, const fe Z) {
    fe t0;
    fe_sq(r->X, X); fe_sq(r->Z, Y);
    fe_sq(r->T, Z); fe_add(r->T, r->T, r->T);
    fe_add(r->Y, X, Y); fe_sq(t0, r->Y);
    fe_add(r->Y, r->Z, r->X);
    fe_sub(r->Z, r->Z, r->X); fe_carry(r->Z);
    fe_sub(r->X, t0, r->Y); fe_carry(r->X);
    fe_sub(r->T, r->T, r->Z); fe_carry(r->T);
}
/* ge.go:183-198 (sub = swap YpX/YmX and negate T2d) */
static void ge_add_cached(ge_p1p1 *r, const ge_p3 *p, const ge_cached *q, int sub) {
    fe t0;
    const uint64_t *qa = sub ? q->YmX : q->YpX, *qb = sub ? q->YpX : q->YmX;
    fe_add(r->X, p->Y, p->X);
    fe_sub(r->Y, p->Y, p->X); fe_carry(r->Y);
    fe_mul(r->Z, r->X, qa);
    fe_mul(r->Y, r->Y, qb);
    fe_mul(r->T, q->T2d, p->T);
    fe_mul(r->X, p->Z, q->Z);
    fe_add(t0, r->X, r->X);
    fe_sub(r->X, r->Z, r->Y); fe_carry(r->X);
    fe_add(r->Y, r->Z, r->Y);
    if (!sub) { fe_add(r->Z, t0, r->T); fe_sub(r->T, t0, r->T); }
    else      { fe_sub(r->Z, t0, r->T); fe_add(r->T, t0, r->T); }
    fe_carry(r->Z); fe_carry(r->T);
}
/* ge.go:217-231 mixed add with an affine precomputed entry */
static void ge_madd(ge_p1p1 *r, const ge_p3 *p, const ge_precomp *q) {
    fe t0;
    fe_add(r->X, p->Y, p->X);
    fe_sub(r->Y, p->Y, p->X); fe_carry(r->Y);
    fe_mul(r->Z, r->X, q->ypx);
    fe_mul(r->Y, r->Y, q->ymx);
    fe_mul(r->T, q->xy2d, p->T);
    fe_add(t0, p->Z, p->Z);
    fe_sub(r->X, r->Z, r->Y); fe_carry(r->X);
    fe_add(r->Y, r->Z, r->Y);
    fe_add(r->Z, t0, r->T);
    fe_sub(r->T, t0, r->T); fe_carry(r->T);
}
static void p3_tobytes(uint8_t s[32], const ge_p3 *p) { /* ge.go:99-107 */
    fe recip, x, y;
    fe_invert(recip, p->Z); fe_mul(x, p->X, recip); fe_mul(y, p->Y, recip);
    fe_tobytes(s, y); s[31] ^= (uint8_t)(fe_isnegative(x) << 7);
}
static int p3_frombytes(ge_p3 *p, const uint8_t s[32]) { /* ge.go:110-150 */
    fe u, v, v3, vxx, check;
    fe_frombytes(p->Y, s); fe_1(p->Z);
    fe_sq(u, p->Y); fe_mul(v, u, FE_D);
    fe_sub(u, u, p->Z); fe_carry(u);
    fe_add(v, v, p->Z);
    fe_sq(v3, v); fe_mul(v3, v3, v);
    fe_sq(p->X, v3); fe_mul(p->X, p->X, v); fe_mul(p->X, p->X, u);
    fe_pow22523(p->X, p->X);
    fe_mul(p->X, p->X, v3); fe_mul(p->X, p->X, u);
    fe_sq(vxx, p->X); fe_mul(vxx, vxx, v);
    fe_sub(check, vxx, u); fe_carry(check);
    if (fe_isnonzero(check)) {
        fe_add(check, vxx, u);
        if (fe_isnonzero(check)) return 0;
        fe_mul(p->X, p->X, FE_SQRTM1);
    }
    if (fe_isnegative(p->X) != (s[31] >> 7)) fe_neg(p->X, p->X);
    fe_mul(p->T, p->X, p->Y);
    return 1;
}
/* signed radix-16 recode (ge.go:374-390, 453-467); returns top digit as produced */
static void recode16(int8_t e[64], const uint8_t a[32]) {
    for (int i = 0; i < 32; i++) { e[2 * i] = a[i] & 15; e[2 * i + 1] = (a[i] >> 4) & 15; }
    int8_t carry = 0;
    for (int i = 0; i < 63; i++) {
        e[i] += carry; carry = (int8_t)((e[i] + 8) >> 4); e[i] -= (int8_t)(carry << 4);
    }
    e[63] += carry;
}
static void select_cached(ge_cached *c, const ge_cached tab[8], int b) { /* ge.go:419-435 */
    int neg = b < 0, babs = neg ? -b : b;
    fe_1(c->YpX); fe_1(c->YmX); fe_1(c->Z); fe_0(c->T2d);
    for (int i = 0; i < 8; i++) {
        uint64_t m = (uint64_t)(babs == i + 1);
        fe_cmov(c->YpX, tab[i].YpX, m); fe_cmov(c->YmX, tab[i].YmX, m);
        fe_cmov(c->Z, tab[i].Z, m); fe_cmov(c->T2d, tab[i].T2d, m);
    }
    if (neg) { fe t; fe_copy(t, c->YpX); fe_copy(c->YpX, c->YmX); fe_copy(c->YmX, t); fe_neg(c->T2d, c->T2d); }
}
static void select_precomp(ge_precomp *t, int pos, int b) { /* ge.go:352-365 */
    int neg = b < 0, babs = neg ? -b : b;
    fe_1(t->ypx); fe_1(t->ymx); fe_0(t->xy2d);
    for (int i = 0; i < 8; i++) {
        uint64_t m = (uint64_t)(babs == i + 1);
        fe_cmov(t->ypx, BASE_TAB[pos][i].ypx, m); fe_cmov(t->ymx, BASE_TAB[pos][i].ymx, m);
        fe_cmov(t->xy2d, BASE_TAB[pos][i].xy2d, m);
    }
    if (neg) { fe x; fe_copy(x, t->ypx); fe_copy(t->ypx, t->ymx); fe_copy(t->ymx, x); fe_neg(t->xy2d, t->xy2d); }
}
static void build_cached_table(ge_cached tab[8], const ge_p3 *A) { /* ge.go:470-476 */
    ge_p1p1 t; ge_p3 u;
    p3_to_cached(&tab[0], A);
    for (int i = 0; i < 7; i++) {
        ge_add_cached(&t, A, &tab[i], 0); p1p1_to_p3(&u, &t); p3_to_cached(&tab[i + 1], &u);
    }
}
/* geScalarMult, ge.go:443-502 */
static void scalarmult(ge_p3 *h, const uint8_t a[32], const ge_p3 *A) {
    int8_t e[64]; ge_cached tab[8], c; ge_p1p1 t; ge_p3 u; ge_p2 r;
    recode16(e, a);
    build_cached_table(tab, A);
    p3_0(&u);
    select_cached(&c, tab, e[63]); /* |e[63]| > 8 matches nothing -> identity */
    ge_add_cached(&t, &u, &c, 0);
    for (int i = 62; i >= 0; i--) {
        p1p1_to_p2(&r, &t); p2_dbl(&t, r.X, r.Y, r.Z);
        p1p1_to_p2(&r, &t); p2_dbl(&t, r.X, r.Y, r.Z);
        p1p1_to_p2(&r, &t); p2_dbl(&t, r.X, r.Y, r.Z);
        p1p1_to_p2(&r, &t); p2_dbl(&t, r.X, r.Y, r.Z);
        p1p1_to_p3(&u, &t);
        select_cached(&c, tab, e[i]);
        ge_add_cached(&t, &u, &c, 0);
    }
    p1p1_to_p3(h, &t);
}
/* all-256-bit multiplier (the semantics of geScalarMultVartime): plain MSB-first
 * double-and-add over the integer a; the reference uses a sliding window
 * (ge.go:298-338) which computes the same group element. */
static void scalarmult_full(ge_p3 *h, const uint8_t a[32], const ge_p3 *A) {
    ge_cached cA; ge_p1p1 t; ge_p2 r; ge_p3 u;
    p3_to_cached(&cA, A);
    p3_0(&u);
    for (int i = 255; i >= 0; i--) {
        p2_dbl(&t, u.X, u.Y, u.Z); p1p1_to_p3(&u, &t);
        if ((a[i >> 3] >> (i & 7)) & 1) { ge_add_cached(&t, &u, &cA, 0); p1p1_to_p3(&u, &t); }
    }
    *h = u; (void)r;
}
/* geScalarMultBase, ge.go:373-417 */
static void scalarmult_base(ge_p3 *h, const uint8_t a[32]) {
    int8_t e[64]; ge_precomp t; ge_p1p1 r; ge_p2 s;
    recode16(e, a);
    p3_0(h);
    for (int i = 1; i < 64; i += 2) { select_precomp(&t, i / 2, e[i]); ge_madd(&r, h, &t); p1p1_to_p3(h, &r); }
    p2_dbl(&r, h->X, h->Y, h->Z); p1p1_to_p2(&s, &r);
    p2_dbl(&r, s.X, s.Y, s.Z);    p1p1_to_p2(&s, &r);
    p2_dbl(&r, s.X, s.Y, s.Z);    p1p1_to_p2(&s, &r);
    p2_dbl(&r, s.X, s.Y, s.Z);    p1p1_to_p3(h, &r);
    for (int i = 0; i < 64; i += 2) { select_precomp(&t, i / 2, e[i]); ge_madd(&r, h, &t); p1p1_to_p3(h, &r); }
}

static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_consts(void) {
    /* d = -121665/121666 */
    fe a, b, bi;
    fe_0(a); a[0] = 121665; fe_neg(a, a);
    fe_0(b); b[0] = 121666; fe_invert(bi, b);
    fe_mul(FE_D, a, bi);
    fe_add(FE_D2, FE_D, FE_D); fe_carry(FE_D2);
    /* sqrt(-1) = 2^((p-1)/4); (p-1)/4 = 2^253 - 5 : via pow22523: 2^(2^252-3) -> square, times 2 */
    fe two; fe_0(two); two[0] = 2;
    fe t; fe_pow22523(t, two);      /* 2^(2^252-3) */
    fe_sq(t, t);                    /* 2^(2^253-6) */
    fe_mul(FE_SQRTM1, t, two);      /* 2^(2^253-5) */
    /* base point: y = 4/5, x even */
    uint8_t enc[32]; memset(enc, 0x66, 32); enc[0] = 0x58;
    p3_frombytes(&GE_BASE, enc);
    /* BASE_TAB[i][j] = (j+1) * 256^i * B in (y+x, y-x, 2dxy) affine form (const.go:102) */
    ge_p3 Pi = GE_BASE;
    for (int i = 0; i < 32; i++) {
        ge_cached cPi; p3_to_cached(&cPi, &Pi);
        ge_p3 acc = Pi;
        for (int j = 0; j < 8; j++) {
            fe zi, x, y;
            fe_invert(zi, acc.Z); fe_mul(x, acc.X, zi); fe_mul(y, acc.Y, zi);
            fe_add(BASE_TAB[i][j].ypx, y, x); fe_carry(BASE_TAB[i][j].ypx);
            fe_sub(BASE_TAB[i][j].ymx, y, x); fe_carry(BASE_TAB[i][j].ymx);
            fe_mul(BASE_TAB[i][j].xy2d, x, y); fe_mul(BASE_TAB[i][j].xy2d, BASE_TAB[i][j].xy2d, FE_D2);
            ge_p1p1 t1; ge_add_cached(&t1, &acc, &cPi, 0); p1p1_to_p3(&acc, &t1);
        }
        for (int k = 0; k < 8; k++) { ge_p1p1 t1; p2_dbl(&t1, Pi.X, Pi.Y, Pi.Z); p1p1_to_p3(&Pi, &t1); }
    }
}

typedef struct {
    int kind; size_t lo, hi; const uint8_t *scalars, *points; uint8_t *out, *status; int vartime;
} job_t;
static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        ge_p3 h;
        if (j->kind == 0) {
            scalarmult_base(&h, j->scalars + 32 * i);
        } else {
            ge_p3 A;
            if (!p3_frombytes(&A, j->points + 32 * i)) {
                if (j->status) j->status[i] = 1;
                memset(j->out + 32 * i, 0, 32);
                continue;
            }
            if (j->status) j->status[i] = 0;
            if (j->vartime) scalarmult_full(&h, j->scalars + 32 * i, &A);
            else scalarmult(&h, j->scalars + 32 * i, &A);
        }
        p3_tobytes(j->out + 32 * i, &h);
    }
    return NULL;
}
static void run(job_t proto, size_t n, int nthreads) {
    pthread_once(&g_once, init_consts);
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    pthread_t *th = malloc(sizeof(pthread_t) * nthreads);
    job_t *jobs = malloc(sizeof(job_t) * nthreads);
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = proto; jobs[t].lo = n * t / nthreads; jobs[t].hi = n * (t + 1) / nthreads;
        if (t) pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    worker(&jobs[0]);
    for (int t = 1; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

/* out[i] = scalars[i] * B */
void ora_ed25519_mul_base(size_t n, const uint8_t *scalars, uint8_t *out, int nthreads) {
    job_t j = {0, 0, 0, scalars, NULL, out, NULL, 0};
    run(j, n, nthreads);
}
/* out[i] = scalars[i] * points[i]; status[i] = 1 and out zeroed when points[i] does not decode */
void ora_ed25519_mul(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t *out,
                     uint8_t *status, int vartime, int nthreads) {
    job_t j = {1, 0, 0, scalars, points, out, status, vartime};
    run(j, n, nthreads);
}
/* out = sum_i scalars[i] * points[i] (N x Mul + N x Add, share/poly.go:340-348).
 * returns 0, or 1 + index of the first undecodable point */
long ora_ed25519_msm(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t out[32]) {
    pthread_once(&g_once, init_consts);
    ge_p3 acc; p3_0(&acc);
    for (size_t i = 0; i < n; i++) {
        ge_p3 A, h; ge_cached c; ge_p1p1 t;
        if (!p3_frombytes(&A, points + 32 * i)) return (long)i + 1;
        scalarmult(&h, scalars + 32 * i, &A);
        p3_to_cached(&c, &h); ge_add_cached(&t, &acc, &c, 0); p1p1_to_p3(&acc, &t);
    }
    p3_tobytes(out, &acc);
    return 0;
}
