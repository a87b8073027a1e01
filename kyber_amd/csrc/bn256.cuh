// pairing/bn256 (dclxvi parameters) instance of the BN device library (bn_suite.inc) + its two hashes to G1
// (pointG1.Hash point.go:261-313; HashG1 hash.go:10-110).
#pragma once
#include "bn256_params.h"
#include "sha256.cuh"
#define KYB_BN_NS bn
#define KYB_BN_PARAMS Bn256
#include "bn_suite.inc"
#undef KYB_BN_NS
#undef KYB_BN_PARAMS

namespace kyb {
namespace bn {
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

// ---- HashG1 (pairing/bn256/hash.go:10-110): HKDF-SHA-256 to the base field (hashToBase, gfp.go:46-68), then the
// Shallue-van de Woestijne map in the reference's arrangement.
// HMAC-SHA-256 (RFC 2104) over up to three message pieces; key = at most 64 bytes (zero-padded by the caller's length)
KYB_HD_NOINLINE void hmac_sha256(uint8_t (&mac)[32], const uint8_t* key, size_t klen, const uint8_t* m1, size_t l1,
                                 const uint8_t* m2, size_t l2, const uint8_t* m3, size_t l3) {
    Sha256 c;
    c.init();
    for (size_t i = 0; i < 64; i++) c.put((uint8_t)((i < klen ? key[i] : 0) ^ 0x36));
    c.update(m1, l1);
    c.update(m2, l2);
    c.update(m3, l3);
    c.finish();
    uint32_t inner[8];
    for (int i = 0; i < 8; i++) inner[i] = c.h[i];
    c.init();
    for (size_t i = 0; i < 64; i++) c.put((uint8_t)((i < klen ? key[i] : 0) ^ 0x5c));
    c.update_words_be(inner, 8);
    c.finish();
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 4; k++) mac[4 * i + k] = (uint8_t)(c.h[i] >> (24 - 8 * k));
}
// OS2IP(48 big-endian bytes) mod p
KYB_HD void fp_from_be48(fp& r, const uint8_t* in) {
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint8_t* b = in + 44 - 4 * k;
        lo[k] = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint8_t* b = in + 12 - 4 * k;
        hi[k] = k < 4 ? (((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]) : 0u;
    }
    fp a, b, c;
    fp_from_words<FC>(a, lo);
    fp_from_words<FC>(b, hi);
    fp_const(c, CC::TWO256);
    fp_mul(b, b, c);
    fp_add(r, a, b);
}
// hashToBase (gfp.go:46-68): t = the first 48 bytes of HKDF(secret = msg, salt = dst, info = "H2C" 0 1) mod p.  An empty
// salt is a key of zeros (RFC 5869), which HMAC's zero padding gives anyway; a salt longer than a block is hashed first.
KYB_HD_NOINLINE void hash_to_base(fp& t, const uint8_t* msg, size_t len, const DstArg& dst) {
    uint8_t key[32], prk[32], t1[32], t2[32], okm[48];
    const uint8_t info1[6] = {'H', '2', 'C', 0, 1, 1}, info2[6] = {'H', '2', 'C', 0, 1, 2};
    if (dst.len > 64) {
        uint32_t h[8];
        sha256(h, dst.b, dst.len);
        for (int i = 0; i < 8; i++)
            for (int k = 0; k < 4; k++) key[4 * i + k] = (uint8_t)(h[i] >> (24 - 8 * k));
        hmac_sha256(prk, key, 32, msg, len, nullptr, 0, nullptr, 0);
    } else {
        hmac_sha256(prk, dst.b, dst.len, msg, len, nullptr, 0, nullptr, 0);
    }
    hmac_sha256(t1, prk, 32, info1, 6, nullptr, 0, nullptr, 0);
    hmac_sha256(t2, prk, 32, t1, 32, info2, 6, nullptr, 0);
    for (int i = 0; i < 32; i++) okm[i] = t1[i];
    for (int i = 0; i < 16; i++) okm[32 + i] = t2[i];
    fp_from_be48(t, okm);
}
// sign0 (gfp.go:137-148): +1 when the residue is >= (p - 1) / 2 (the comparison falls through to 1 on equality), else -1
KYB_HD bool fp_sign0_pos(const fp& a) {
    uint32_t w[8];
    fp_to_words<FC>(w, a);
    for (int k = 7; k >= 0; k--) {
        if (w[k] > FC::HALF[k]) return true;
        if (w[k] < FC::HALF[k]) return false;
    }
    return true;
}
KYB_HD bool fp_legendre_is_one(const fp& a) {  // legendre(a) == 1 (gfp.go:150-164): a^((p-1)/2) == 1
    fp t, one;
    fp_one(one);
    fp_pow_words<FC>(t, a, FC::HALF, FC::PBITS);
    return fp_eq(t, one);
}
KYB_HD void curve_g(fp& r, const fp& x) {  // x^3 + 3
    fp b;
    fp_const(b, CC::B1);
    fp_sqr(r, x);
    fp_mul(r, r, x);
    fp_add(r, r, b);
}
// mapToCurve (hash.go:14-110): w = (s t)^2 / (s t (1 + B + t^2)); x1 = (s - 1) / 2 - t w, x2 = -1 - x1,
// x3 = 1 + (1 + B + t^2)^4 / (s t (1 + B + t^2))^2; the first with legendre(x^3 + 3) == 1 (x3 unconditionally), y the
// power (p + 1) / 4 with the sign of sign0(t)
KYB_HD_NOINLINE void map_to_curve_svdw(fp& x, fp& y, const fp& t) {
    fp one, b, s, a, st, w0, w, x1, x2, x3, gx, tmp;
    fp_one(one);
    fp_const(b, CC::B1);
    fp_const(s, CC::SVDW_S);
    fp_sqr(a, t);
    fp_add(a, a, b);
    fp_add(a, a, one);
    fp_mul(st, s, t);
    fp_mul(w0, st, a);
    fp_inv(w0, w0);  // gfP.Invert is the power p - 2: 0 -> 0
    fp_sqr(w, st);
    fp_mul(w, w, w0);
    fp_const(x1, CC::SVDW_SM1D2);
    fp_mul(tmp, t, w);
    fp_sub(x1, x1, tmp);
    fp_neg(x2, one);
    fp_sub(x2, x2, x1);
    fp_sqr(x3, a);
    fp_sqr(x3, x3);
    fp_mul(x3, x3, w0);
    fp_mul(x3, x3, w0);
    fp_add(x3, x3, one);
    x = x3;
    curve_g(gx, x2);
    if (fp_legendre_is_one(gx)) x = x2;
    curve_g(gx, x1);
    if (fp_legendre_is_one(gx)) x = x1;
    curve_g(gx, x);
    fp_pow_words<FC>(y, gx, FC::SQRT_EXP, FC::SQRT_BITS);
    if (fp_sign0_pos(t) != fp_sign0_pos(y)) fp_neg(y, y);
}
// HashG1(msg, dst) -> the 64-byte G1 encoding (pointG1.MarshalBinary; the map never yields infinity)
KYB_HD int hash_g1_svdw_wire(uint8_t* out, const uint8_t* msg, size_t len, const DstArg& dst) {
    fp t, x, y;
    hash_to_base(t, msg, len, dst);
    map_to_curve_svdw(x, y, t);
    fp_encode(out, x);
    fp_encode(out + 32, y);
    return ST_OK;
}
}  // namespace bn
}  // namespace kyb
