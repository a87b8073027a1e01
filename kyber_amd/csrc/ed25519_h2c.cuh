// Ed25519 hash-to-curve, suite edwards25519_XMD:SHA-512_ELL2_RO_ (RFC 9380 sections 5.3.1, 6.8.2, 8.5), one
// message per lane.  Replaces (*point).Hash (group/edwards25519/point.go:325-334): hashToField (:336-360,
// expandMessageXMD :362-430), mapToCurveElligator2Ed25519, Add, Mul by the cofactor 8.
#pragma once
#include "ge25519.cuh"
#include "sha512.cuh"

namespace kyb {

struct EdDstArg {  // domain separation tag, at most 255 bytes (longer tags are hashed down by the caller)
    uint8_t b[256];
    uint32_t len;
};

// 96 uniform bytes (two 48-byte field-element seeds) as twelve big-endian 64-bit words
KYB_DEV void ed_expand_message_xmd_96(uint64_t (&out)[12], const uint8_t* msg, size_t msg_len, const EdDstArg& dst) {
    Sha512 c;
    c.init();
    for (int i = 0; i < 128; i++) c.put(0);  // Z_pad: one zero block
    c.update(msg, msg_len);
    c.put(0);
    c.put(96);  // l_i_b_str = I2OSP(96, 2)
    c.put(0);
    c.update(dst.b, dst.len);
    c.put((uint8_t)dst.len);
    c.finish();
    uint64_t b0[8], b1[8];
    for (int i = 0; i < 8; i++) b0[i] = c.h[i];
    c.init();
    c.update_words_be(b0, 8);
    c.put(1);
    c.update(dst.b, dst.len);
    c.put((uint8_t)dst.len);
    c.finish();
    for (int i = 0; i < 8; i++) b1[i] = c.h[i];
    uint64_t x[8];
    for (int i = 0; i < 8; i++) x[i] = b0[i] ^ b1[i];
    c.init();
    c.update_words_be(x, 8);
    c.put(2);
    c.update(dst.b, dst.len);
    c.put((uint8_t)dst.len);
    c.finish();
    for (int i = 0; i < 8; i++) out[i] = b1[i];
    for (int i = 0; i < 4; i++) out[8 + i] = c.h[i];
}
// OS2IP(48 bytes) mod p, the bytes given as six big-endian 64-bit words (most significant first)
KYB_DEV void fe_from_be384(fe& r, const uint64_t* be) {
    // little-endian 32-bit words of the 384-bit integer
    uint32_t w[12];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        w[2 * k] = (uint32_t)be[5 - k];
        w[2 * k + 1] = (uint32_t)(be[5 - k] >> 32);
    }
    // value = lo (bits 0..254) + 2^255 * hi (129 bits);  2^255 = 19 mod p
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int k = 0; k < 8; k++) lo[k] = w[k];
    lo[7] &= 0x7fffffffu;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int src = 7 + k;  // bits 255.. : shift right by 31 within word 7
        const uint32_t a = src < 12 ? w[src] >> 31 : 0;
        const uint32_t b = src + 1 < 12 ? w[src + 1] << 1 : 0;
        hi[k] = a | b;
    }
    fe flo, fhi, c19;
    fe_fromwords(flo, lo);
    fe_fromwords(fhi, hi);
    fe_0(c19);
    c19.v[0] = 19;
    KYB_FE_MAG_SET(c19, 19.0 / (1 << 25));
    fe_mul(fhi, fhi, c19);
    fe_add(r, flo, fhi);
    fe one;
    fe_1(one);
    fe_mul(r, r, one);  // carry the sum back into the multiplier's input range
}
KYB_DEV bool fe_eq(const fe& a, const fe& b) {
    fe d;
    fe_sub(d, a, b);
    return !fe_isnonzero(d);
}
// square root in GF(2^255-19) (p = 5 mod 8); false when a is not a square
KYB_DEV bool fe_sqrt(fe& r, const fe& a) {
    fe c, c2, na;
    fe_pow22523(c, a);  // a^((p-5)/8)
    fe_mul(c, c, a);    // a^((p+3)/8)
    fe_sq(c2, c);
    fe_neg(na, a);
    const bool ok1 = fe_eq(c2, a), ok2 = fe_eq(c2, na);
    fe ci;
    fe_mul(ci, c, fe_sqrtm1());
    fe_cmov(c, ci, ok2);
    r = c;
    return ok1 | ok2;
}
// map_to_curve_elligator2_edwards25519 (RFC 9380 section 6.8.2; behaviour of mapToCurveElligator2Ed25519): Elligator 2 on
// curve25519 (J = 486662, Z = 2) followed by the birational map to the Edwards curve -- in the straight-line form of the
// RFC's appendix G.2.1 / G.2.2: ONE power (gx1 gxd^7)^((p-5)/8) yields the root for x1 and, times u 2^((p+3)/8), the root
// for x2 = 2 u^2 x1; everything stays a fraction (xn / xd, y) and the fractions go straight into extended coordinates
// (X : Y : Z : T) = (xn yd : yn xd : xd yd : xn yn) -- no inversion.  Round 1's version inverted the denominator, took a
// square root for x1, a second one for x2 where the first failed (half the lanes of any wave) and inverted again for
// the Edwards map: two powers and two inversions of 254 squarings each per map where this takes one.
KYB_DEV void ed_map_to_curve(ge_p3& q, const fe& u) {
    fe J, nJ, one, tv1, tv2, tv3, xd, gxd, gx1, gx2, y11, y12, y21, y22, y1, y2, x2n, xn, y, t;
    fe_0(J);
    J.v[0] = 486662;
    KYB_FE_MAG_SET(J, 486662.0 / (1 << 25));
    fe_1(one);
    fe_neg(nJ, J);            // x1n = -J
    fe_sq2(tv1, u);           // 2 u^2
    fe_add(xd, tv1, one);     // 1 + 2 u^2: never zero (-1/2 is no square)
    fe_sq(tv2, xd);
    fe_mul(gxd, tv2, xd);     // xd^3
    fe_mul(gx1, J, tv1);
    fe_mul(gx1, gx1, nJ);
    fe_add(gx1, gx1, tv2);
    fe_mul(gx1, gx1, nJ);     // x1n^3 + J x1n^2 xd + x1n xd^2 = g(x1) xd^3
    fe_sq(tv3, gxd);
    fe_sq(tv2, tv3);          // gxd^4
    fe_mul(tv3, tv3, gxd);    // gxd^3
    fe_mul(tv3, tv3, gx1);    // gx1 gxd^3
    fe_mul(tv2, tv2, tv3);    // gx1 gxd^7
    fe_pow22523(y11, tv2);    // ^((p-5)/8)
    fe_mul(y11, y11, tv3);
    fe_mul(y12, y11, fe_sqrtm1());
    fe_sq(tv2, y11);
    fe_mul(tv2, tv2, gxd);
    const bool e1 = fe_eq(tv2, gx1);
    y1 = y12;
    fe_cmov(y1, y11, e1);     // the root of g(x1) if there is one
    fe_mul(x2n, nJ, tv1);     // x2 = 2 u^2 x1
    fe_mul(y21, y11, u);
    fe_mul(y21, y21, fe_elligator_c2());
    fe_mul(y22, y21, fe_sqrtm1());
    fe_mul(gx2, gx1, tv1);    // g(x2) xd^3 = 2 u^2 g(x1) xd^3
    fe_sq(tv2, y21);
    fe_mul(tv2, tv2, gxd);
    const bool e2 = fe_eq(tv2, gx2);
    y2 = y22;
    fe_cmov(y2, y21, e2);
    fe_sq(tv2, y1);
    fe_mul(tv2, tv2, gxd);
    const bool e3 = fe_eq(tv2, gx1);  // g(x1) is a square: x = x1, else x = x2
    xn = x2n;
    fe_cmov(xn, nJ, e3);
    y = y2;
    fe_cmov(y, y1, e3);
    fe ny;
    fe_neg(ny, y);
    fe_cmov(y, ny, e3 != fe_isnegative(y));  // sgn0(y) = 1 on the first branch, 0 on the second
    // Montgomery (xn / xd, y) -> Edwards: x = c1 (xn / xd) / y, y = (xn - xd) / (xn + xd); a zero denominator -> (0, 1)
    fe xne, xde, yne, yde, z;
    fe_mul(xne, xn, fe_elligator_c1());
    fe_mul(xde, xd, y);
    fe_sub(yne, xn, xd);
    fe_add(yde, xn, xd);
    fe_mul(z, xde, yde);
    const bool exc = !fe_isnonzero(z);
    fe_mul(q.X, xne, yde);
    fe_mul(q.Y, yne, xde);
    fe_mul(q.T, xne, yne);
    q.Z = z;
    fe_0(t);
    fe_cmov(q.X, t, exc);
    fe_cmov(q.T, t, exc);
    fe_cmov(q.Y, one, exc);
    fe_cmov(q.Z, one, exc);
}
// (*point).Hash: 32-byte encoding of 8 * (map(u0) + map(u1))
KYB_DEV void ed_hash_wire(uint8_t* out, const uint8_t* msg, size_t msg_len, const EdDstArg& dst) {
    uint64_t ub[12];
    ed_expand_message_xmd_96(ub, msg, msg_len, dst);
    fe u0, u1;
    fe_from_be384(u0, ub);
    fe_from_be384(u1, ub + 6);
    ge_p3 q0, q1, r;
    ed_map_to_curve(q0, u0);
    ed_map_to_curve(q1, u1);
    ge_cached c;
    ge_p3_to_cached(c, q1);
    ge_p1p1 t;
    ge_add(t, q0, c);
    ge_p1p1_to_p3(r, t);
#pragma unroll 1
    for (int k = 0; k < 3; k++) {
        ge_dbl(t, r.X, r.Y, r.Z);
        ge_p1p1_to_p3(r, t);
    }
    uint32_t w[8];
    ge_p3_towords(w, r);
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = w[k];
}

}  // namespace kyb
