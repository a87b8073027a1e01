// pairing/bn254 (Ethereum's alt_bn128) instance of the BN device library (bn_suite.inc: strict coordinate decoding and
// the G2 subgroup check of bn254/twist.go:47-66 are its CC::STRICT branches) + pointG1.Hash of that suite.
#pragma once
#include "bn254_params.h"
#include "keccak256.cuh"
#define KYB_BN_NS bn4
#define KYB_BN_PARAMS Bn254
#include "bn_suite.inc"
#undef KYB_BN_NS
#undef KYB_BN_PARAMS

namespace kyb {
namespace bn4 {

// uniform_bytes = expand_message_xmd(msg, DST, 96) with H = legacy Keccak-256 (bn254/point.go:289-340): b_1 || b_2 || b_3
KYB_HD_NOINLINE void expand_message_keccak96(uint8_t (&out)[96], const uint8_t* msg, size_t msg_len, const DstArg& dst) {
    Keccak256 c;
    uint8_t b0[32], bi[32];
    c.init();
    for (int i = 0; i < 136; i++) c.put(0);  // Z_pad: one rate of zeros
    c.update(msg, msg_len);
    c.put(0);
    c.put(96);
    c.put(0);
    c.update(dst.b, dst.len);
    c.put((uint8_t)dst.len);
    c.finish(b0);
#pragma unroll 1
    for (int blk = 1; blk <= 3; blk++) {
        c.init();
        for (int i = 0; i < 32; i++) c.put(blk == 1 ? b0[i] : (uint8_t)(b0[i] ^ bi[i]));
        c.put((uint8_t)blk);
        c.update(dst.b, dst.len);
        c.put((uint8_t)dst.len);
        c.finish(bi);
        for (int i = 0; i < 32; i++) out[32 * (blk - 1) + i] = bi[i];
    }
}
// OS2IP(48 big-endian bytes) mod p (hashToField, point.go:223-235)
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
KYB_HD bool fp_sgn0(const fp& a) {
    uint32_t w[8];
    fp_to_words<FC>(w, a);
    return w[0] & 1;
}
KYB_HD void curve_g(fp& r, const fp& x) {  // x^3 + 3
    fp b;
    fp_const(b, CC::B1);
    fp_sqr(r, x);
    fp_mul(r, r, x);
    fp_add(r, r, b);
}
KYB_HD bool fp_legendre_is_one(const fp& a) {  // legendre(a) == 1 (gfp.go:128-143): a^((p-1)/2) == 1
    fp t, one;
    fp_one(one);
    fp_pow_words<FC>(t, a, FC::HALF, FC::PBITS);
    return fp_eq(t, one);
}
// mapToPoint (point.go:239-285): Shallue-van de Woestijne, x1 / x2 / x3 in this order with legendre == 1, y = sqrt by
// exponentiation, sign of y made equal to sgn0(u)
KYB_HD_NOINLINE void map_to_point(g1_jac& r, const fp& u) {
    fp one, c1, c2, c3, c4, tv1, tv2, tv3, tv5, x1, x2, x3, t, gx;
    fp_one(one);
    fp_const(c1, CC::SVDW_C1);
    fp_const(c2, CC::SVDW_C2);
    fp_const(c3, CC::SVDW_C3);
    fp_const(c4, CC::SVDW_C4);
    fp_sqr(tv1, u);
    fp_mul(tv1, tv1, c1);
    fp_add(tv2, one, tv1);
    fp_sub(tv1, one, tv1);
    fp_mul(tv3, tv1, tv2);
    fp_inv(tv3, tv3);  // inv0: 0 -> 0
    fp_mul(tv5, u, tv1);
    fp_mul(tv5, tv5, tv3);
    fp_mul(tv5, tv5, c3);
    fp_sub(x1, c2, tv5);
    fp_add(x2, c2, tv5);
    fp_sqr(t, tv2);
    fp_mul(t, t, tv3);
    fp_sqr(x3, t);
    fp_mul(x3, x3, c4);
    fp_add(x3, x3, one);
    fp x = x3;
    curve_g(gx, x2);
    if (fp_legendre_is_one(gx)) x = x2;
    curve_g(gx, x1);
    if (fp_legendre_is_one(gx)) x = x1;
    curve_g(gx, x);
    fp y;
    fp_pow_words<FC>(y, gx, FC::SQRT_EXP, FC::SQRT_BITS);
    if (fp_sgn0(u) != fp_sgn0(y)) fp_neg(y, y);
    g1_aff a;
    a.x = x;
    a.y = y;
    a.inf = false;
    jac_from_aff(r, a);
}
// pointG1.Hash -> hashToPoint (point.go:211-221): hash_to_field, two maps, one addition; G1 has cofactor 1.
KYB_HD int hash_g1_wire(uint8_t* out, const uint8_t* msg, size_t len, const DstArg& dst) {
    uint8_t ub[96];
    expand_message_keccak96(ub, msg, len, dst);
    fp e0, e1;
    fp_from_be48(e0, ub);
    fp_from_be48(e1, ub + 48);
    g1_jac p0, p1, s;
    map_to_point(p0, e0);
    map_to_point(p1, e1);
    jac_add(s, p0, p1);
    g1_aff a;
    jac_to_aff(a, s);
    g1_encode(out, a);
    return ST_OK;
}

}  // namespace bn4
}  // namespace kyb
