// bn256 (dclxvi parameters) G1 / G2 / GT device library: the reference's wire formats and
// acceptance rules, scalar multiplication, optimal ate pairing.
//
// Replaces pairing/bn256 (all in-tree): pointG1/G2.Mul + (Un)MarshalBinary (point.go:154-238,
// 405-499), curvePoint/twistPoint.Mul (curve.go:189, twist.go:162), miller / lineFunctionAdd /
// lineFunctionDouble / mulLine / finalExponentiation / optimalAte (optate.go), pointGT.MarshalBinary
// (point.go:630-662), Suite.Pair / ValidatePairing (suite.go:97-107).
// The Miller loop and the final exponentiation follow optate.go's formulas and addition chain so
// that GT bytes are those of the reference (oracle/bn256.py restates the same code).
#pragma once
#include "bn256_params.h"
#include "curve.cuh"
#include "sha256.cuh"

namespace kyb {
namespace bn {

using FC = Bn256Fp;
using TC = Bn256Tower;
using CC = Bn256Curve;
using fp = Fp<FC>;
using fp2 = Fp2<TC>;
using fp6 = Fp6<TC>;
using fp12 = Fp12<TC>;
using g1_aff = Aff<fp>;
using g2_aff = Aff<fp2>;
using g1_jac = Jac<fp>;
using g2_jac = Jac<fp2>;

constexpr int ST_OK = 0, ST_BAD_POINT = 1;

KYB_HD void fp_const(fp& r, const uint32_t (&c)[FC::NWORDS]) {
#pragma unroll
    for (int l = 0; l < FC::NWORDS; l++) r.v[l] = c[l];
}

// ------------------------------------------------------------------ decoding
// 32-byte big-endian coordinate, reduced mod p like montEncode does (point.go:218-221): values
// >= p are accepted.
KYB_HD void fp_decode(fp& r, const uint8_t* in) {
    uint32_t w[8];
    words_from_be<8>(w, in);
    fp_from_words<FC>(r, w);
}
KYB_HD void fp_encode(uint8_t* out, const fp& a) {
    uint32_t w[8];
    fp_to_words<FC>(w, a);
    words_to_be<8>(out, w);
}
// pointG1.UnmarshalBinary (point.go:206-238): (0, 0) is infinity; otherwise y^2 = x^3 + 3.
// bn256's wire format is already uncompressed affine and its UnmarshalBinary has no subgroup check
// (point.go:206-238, 466-499), so the flags of the shared ABI change nothing here.
KYB_HD size_t g1_wire_size(uint32_t) { return 64; }
KYB_HD size_t g2_wire_size(uint32_t) { return 128; }
KYB_HD size_t g1_out_size(uint32_t) { return 64; }
KYB_HD size_t g2_out_size(uint32_t) { return 128; }
KYB_HD_NOINLINE int g1_decode(g1_aff& a, const uint8_t* in) {
    fp_decode(a.x, in);
    fp_decode(a.y, in + 32);
    a.inf = fp_is_zero(a.x) & fp_is_zero(a.y);
    fp y2, x3, b;
    fp_const(b, CC::B1);
    fp_sqr(y2, a.y);
    fp_sqr(x3, a.x);
    fp_mul(x3, x3, a.x);
    fp_add(x3, x3, b);
    return (a.inf || fp_eq(y2, x3)) ? ST_OK : ST_BAD_POINT;
}
// pointG2.UnmarshalBinary (point.go:466-499): x.x, x.y, y.x, y.y with gfP2{x, y} = x i + y;
// on-curve check only, never a subgroup check.
KYB_HD_NOINLINE int g2_decode(g2_aff& a, const uint8_t* in) {
    fp_decode(a.x.c1, in);
    fp_decode(a.x.c0, in + 32);
    fp_decode(a.y.c1, in + 64);
    fp_decode(a.y.c0, in + 96);
    a.inf = fp2_is_zero(a.x) & fp2_is_zero(a.y);
    fp2 y2, x3, b;
    fp2_load_const<TC>(b, CC::B2);
    fp2_sqr_c(y2, a.y);
    fp2_sqr_c(x3, a.x);
    fp2_mul_c(x3, x3, a.x);
    fp2_add(x3, x3, b);
    return (a.inf || fp2_eq(y2, x3)) ? ST_OK : ST_BAD_POINT;
}
// MarshalBinary: infinity is all zero bytes (point.go:170-192, 423-452)
KYB_HD_NOINLINE void g1_encode(uint8_t* out, const g1_aff& a) {
    fp x = a.x, y = a.y;
    if (a.inf) {
        fp_zero(x);
        fp_zero(y);
    }
    fp_encode(out, x);
    fp_encode(out + 32, y);
}
KYB_HD_NOINLINE void g2_encode(uint8_t* out, const g2_aff& a) {
    fp2 x = a.x, y = a.y;
    if (a.inf) {
        fp2_zero(x);
        fp2_zero(y);
    }
    fp_encode(out, x.c1);
    fp_encode(out + 32, x.c0);
    fp_encode(out + 64, y.c1);
    fp_encode(out + 96, y.c0);
}
// pointGT.MarshalBinary (point.go:630-662): x.x.x, x.x.y, ..., y.z.y  ==  omega-coefficient
// first, tau^2 / tau / 1 order inside, imaginary part before real part.
KYB_HD_NOINLINE void gt_encode(uint8_t* out, const fp12& f) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const fp6& s = h == 0 ? f.c1 : f.c0;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const fp2& c = m == 0 ? s.c2 : (m == 1 ? s.c1 : s.c0);
            fp_encode(out + (h * 3 + m) * 64, c.c1);
            fp_encode(out + (h * 3 + m) * 64 + 32, c.c0);
        }
    }
}

// pointGT.UnmarshalBinary (point.go:664-716): twelve coefficients reduced mod p, no membership check
KYB_HD_NOINLINE void gt_decode(fp12& f, const uint8_t* in) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        fp6& s = h == 0 ? f.c1 : f.c0;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            fp2& c = m == 0 ? s.c2 : (m == 1 ? s.c1 : s.c0);
            fp_decode(c.c1, in + (h * 3 + m) * 64);
            fp_decode(c.c0, in + (h * 3 + m) * 64 + 32);
        }
    }
}
// gfP12.Exp (gfp12.go:177-192) for a plain 256-bit exponent; general Fp12 squarings because
// UnmarshalBinary admits elements outside the cyclotomic subgroup
KYB_HD_NOINLINE void gt_pow_u256(fp12& r, const fp12& a, const uint32_t (&k)[8]) {
    fp12 acc;
    fp12_one(acc);
#pragma unroll 1
    for (int i = 255; i >= 0; i--) {
        fp12_sqr(acc, acc);
        if ((k[i >> 5] >> (i & 31)) & 1) fp12_mul(acc, acc, a);
    }
    r = acc;
}

// ------------------------------------------------- GLV scalar multiplication on G1
// G1 = E(Fp) has prime order n (cofactor 1), so phi(x, y) = (beta x, y) acts as [lambda] on every accepted point and
// k P = k1 P + k2 phi(P) for the Babai-rounded split k = k1 + k2 lambda (mod n), |k1|, |k2| < 2^130 (constants and
// their derivation: gen_consts.py bn256()).  34 windows of (4 doublings + 2 additions) instead of 65 x (4 + 1).
// G2 keeps the plain ladder: its UnmarshalBinary accepts twist points outside the order-n subgroup, where the
// Frobenius eigenvalue relation does not hold and the reference's double-and-add result must be reproduced.
//
// r (NR low words) = a (NA words) * b (NB words)
template <int NR, int NA, int NB>
KYB_HD void mul_words(uint32_t (&r)[NR], const uint32_t* a, const uint32_t* b) {
    uint64_t acc = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        // column k: sum a[i] b[k - i]; 96-bit accumulation in (hi : acc)
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int j = k - i;
            if (j < 0 || j >= NB) continue;
            const uint64_t pr = (uint64_t)a[i] * b[j];
            acc += pr;
            hi += acc < pr ? 1u : 0u;
        }
        r[k] = (uint32_t)acc;
        acc = (acc >> 32) | (hi << 32);
        hi = 0;
    }
}
KYB_HD bool abs_words5(uint32_t (&x)[5]) {  // 160-bit two's complement -> magnitude, returns the sign
    const bool neg = (x[4] >> 31) != 0;
    uint32_t b = 0, t[5];
#pragma unroll
    for (int i = 0; i < 5; i++) t[i] = sbb32(0u, x[i], b);
#pragma unroll
    for (int i = 0; i < 5; i++) x[i] = neg ? t[i] : x[i];
    return neg;
}
// k -> (|k1|, sign1, |k2|, sign2)
KYB_HD void glv_split(uint32_t (&m1)[5], bool& n1, uint32_t (&m2)[5], bool& n2, const uint32_t (&k)[8]) {
    uint32_t g1[3], g2[5], a1[2], a2[4], b1n[4], b2[2];
#pragma unroll
    for (int i = 0; i < 3; i++) g1[i] = CC::GLV_G1[i];
#pragma unroll
    for (int i = 0; i < 5; i++) g2[i] = CC::GLV_G2[i];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        a1[i] = CC::GLV_A1[i];
        b2[i] = CC::GLV_B2[i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a2[i] = CC::GLV_A2[i];
        b1n[i] = CC::GLV_B1N[i];
    }
    uint32_t p1[11], p2[13];
    mul_words<11, 8, 3>(p1, k, g1);
    mul_words<13, 8, 5>(p2, k, g2);
    uint32_t c1[3] = {p1[8], p1[9], p1[10]};
    uint32_t c2[5] = {p2[8], p2[9], p2[10], p2[11], p2[12]};
    uint32_t t1[5], t2[5], t3[5], t4[5];
    mul_words<5, 3, 2>(t1, c1, a1);   // c1 A1
    mul_words<5, 5, 4>(t2, c2, a2);   // c2 A2
    mul_words<5, 3, 4>(t3, c1, b1n);  // c1 |B1|
    mul_words<5, 5, 2>(t4, c2, b2);   // c2 B2
    uint32_t b = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) m1[i] = sbb32(k[i], t1[i], b);
    b = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) m1[i] = sbb32(m1[i], t2[i], b);  // k1 = k - c1 A1 - c2 A2   (mod 2^160, |k1| < 2^130)
    b = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) m2[i] = sbb32(t3[i], t4[i], b);  // k2 = c1 |B1| - c2 B2
    n1 = abs_words5(m1);
    n2 = abs_words5(m2);
}
KYB_HD_NOINLINE void g1_mul_glv(g1_jac& r, const g1_jac& p, const uint32_t (&k)[8]) {
    uint32_t m1[5], m2[5];
    bool n1, n2;
    glv_split(m1, n1, m2, n2, k);
    uint32_t w1[8], w2[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        w1[i] = i < 5 ? m1[i] : 0u;
        w2[i] = i < 5 ? m2[i] : 0u;
    }
    int8_t e1[65], e2[65];
    recode16_u256(e1, w1);
    recode16_u256(e2, w2);
    g1_jac tab[8];  // (j + 1) * P
    tab[0] = p;
    jac_dbl(tab[1], p);
#pragma unroll 1
    for (int j = 2; j < 8; j++) jac_add(tab[j], tab[j - 1], p);
    fp beta;
    fp_const(beta, CC::BETA);
    // (fusing the window into one out-of-line step, as BLS12-381 G1 does, cost 13 % here: left as separate calls)
    g1_jac acc, t, s;
    jac_set_inf(acc);
#pragma unroll 1
    for (int i = 33; i >= 0; i--) {
        if (i != 33) jac_dbl_n(acc, acc, 4);
        jac_select8(t, tab, n1 ? -e1[i] : e1[i]);
        jac_add(s, acc, t);
        jac_cmov(acc, s, e1[i] != 0);
        jac_select8(t, tab, n2 ? -e2[i] : e2[i]);
        fp_mul(t.X, t.X, beta);
        jac_add(s, acc, t);
        jac_cmov(acc, s, e2[i] != 0);
    }
    r = acc;
}

// ------------------------------------------------- per-element wire-level operations
KYB_HD void zero_bytes(uint8_t* out, int n) {
    uint32_t* q = reinterpret_cast<uint32_t*>(out);
    for (int k = 0; k < n / 4; k++) q[k] = 0;
}
KYB_HD int g1_mul_wire(uint8_t* out, const uint8_t* scalar_be, const uint8_t* pt, uint32_t = 0) {
    g1_aff a;
    const int st = g1_decode(a, pt);
    if (st != ST_OK) {
        zero_bytes(out, 64);
        return st;
    }
    uint32_t k[8];
    words_from_be<8>(k, scalar_be);
    g1_jac p, r;
    jac_from_aff(p, a);
    g1_mul_glv(r, p, k);
    jac_to_aff(a, r);
    g1_encode(out, a);
    return ST_OK;
}
KYB_HD int g2_mul_wire(uint8_t* out, const uint8_t* scalar_be, const uint8_t* pt, uint32_t = 0) {
    g2_aff a;
    const int st = g2_decode(a, pt);
    if (st != ST_OK) {
        zero_bytes(out, 128);
        return st;
    }
    uint32_t k[8];
    words_from_be<8>(k, scalar_be);
    g2_jac p, r;
    jac_from_aff(p, a);
    jac_mul_u256(r, p, k);
    jac_to_aff(a, r);
    g2_encode(out, a);
    return ST_OK;
}
// out = Marshal(Unmarshal(in)): pointG1/pointG2.UnmarshalBinary (pairing/bn256/point.go:206-238, 466-499): coordinates
// < p, on the curve (G2: on the twist, no subgroup check), all-zero bytes = infinity.
KYB_HD int g1_unmarshal_wire(uint8_t* out, const uint8_t* pt, uint32_t = 0) {
    g1_aff a;
    const int st = g1_decode(a, pt);
    if (st != ST_OK) {
        zero_bytes(out, 64);
        return st;
    }
    g1_encode(out, a);
    return ST_OK;
}
KYB_HD int g2_unmarshal_wire(uint8_t* out, const uint8_t* pt, uint32_t = 0) {
    g2_aff a;
    const int st = g2_decode(a, pt);
    if (st != ST_OK) {
        zero_bytes(out, 128);
        return st;
    }
    g2_encode(out, a);
    return ST_OK;
}
// out = a + b   (Point.Add: kilic/g1.go:90-96, pairing/bn256/point.go:130-140 -> curve.go:69)
KYB_HD int g1_add_wire(uint8_t* out, const uint8_t* pa, const uint8_t* pb) {
    g1_aff a, b;
    int st = g1_decode(a, pa);
    const int st2 = g1_decode(b, pb);
    if (st == ST_OK) st = st2;
    if (st != ST_OK) {
        zero_bytes(out, 64);
        return st;
    }
    g1_jac p, q, r;
    jac_from_aff(p, a);
    jac_from_aff(q, b);
    jac_add(r, p, q);
    jac_to_aff(a, r);
    g1_encode(out, a);
    return ST_OK;
}
KYB_HD int g2_add_wire(uint8_t* out, const uint8_t* pa, const uint8_t* pb) {
    g2_aff a, b;
    int st = g2_decode(a, pa);
    const int st2 = g2_decode(b, pb);
    if (st == ST_OK) st = st2;
    if (st != ST_OK) {
        zero_bytes(out, 128);
        return st;
    }
    g2_jac p, q, r;
    jac_from_aff(p, a);
    jac_from_aff(q, b);
    jac_add(r, p, q);
    jac_to_aff(a, r);
    g2_encode(out, a);
    return ST_OK;
}
// pointG1.Hash -> hashToPoint (point.go:261-313): x = SHA-256(m) mod p; while x^3 + 3 has no square root
// x += 1; y = (x^3 + 3)^((p+1)/4) (big.Int.ModSqrt for p = 3 mod 4).  Output: 64-byte G1 encoding.
KYB_HD int hash_g1_wire(uint8_t* out, const uint8_t* msg, size_t len) {
    uint32_t h[8], w[8];
    sha256(h, msg, len);
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = h[7 - k];  // digest as a big-endian integer -> little-endian words
    fp x, y, t, b, one;
    fp_from_words<FC>(x, w);  // reduces mod p
    fp_const(b, CC::B1);
    fp_one(one);
    bool found = false;
#pragma unroll 1
    for (int iter = 0; iter < 256 && !found; iter++) {
        fp_sqr(t, x);
        fp_mul(t, t, x);
        fp_add(t, t, b);
        fp_pow_words<FC>(y, t, FC::SQRT_EXP, FC::SQRT_BITS);
        fp y2;
        fp_sqr(y2, y);
        found = fp_eq(y2, t);
        if (!found) fp_add(x, x, one);
    }
    fp_encode(out, x);
    fp_encode(out + 32, y);
    return found ? ST_OK : ST_BAD_POINT;
}
// out = gt^k   (pointGT.Mul, point.go:613-628 -> gfP12.Exp)
KYB_HD int gt_mul_wire(uint8_t* out, const uint8_t* scalar_be, const uint8_t* gt) {
    fp12 f;
    gt_decode(f, gt);
    uint32_t k[8];
    words_from_be<8>(k, scalar_be);
    gt_pow_u256(f, f, k);
    gt_encode(out, f);
    return ST_OK;
}
}  // namespace bn
}  // namespace kyb
