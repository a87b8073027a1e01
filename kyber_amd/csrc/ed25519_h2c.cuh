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
// map_to_curve_elligator2_edwards25519 (RFC 9380 section 6.8.2; behaviour of mapToCurveElligator2Ed25519):
// Elligator 2 on curve25519 (J = 486662, Z = 2) followed by the birational map to the Edwards curve.
KYB_DEV void ed_map_to_curve(ge_p3& q, const fe& u) {
    fe J, one, t, den, x1, gx, y, xm, nJ;
    fe_0(J);
    J.v[0] = 486662;
    KYB_FE_MAG_SET(J, 486662.0 / (1 << 25));
    fe_1(one);
    fe_neg(nJ, J);
    fe_sq2(t, u);  // 2 u^2
    fe_add(den, t, one);
    fe_mul(den, den, one);
    const bool den_zero = !fe_isnonzero(den);
    fe_invert(x1, den);
    fe_mul(x1, x1, nJ);  // -J / (1 + 2 u^2)
    fe_cmov(x1, nJ, den_zero);
    // g(x) = x^3 + J x^2 + x
    fe_sq(t, x1);
    fe_add(gx, x1, J);
    fe_mul(gx, gx, t);
    fe_add(gx, gx, x1);
    const bool sq1 = fe_sqrt(y, gx);
    xm = x1;
    bool want_odd = true;  // sgn0(y) = 1 on the first branch, 0 on the second
    if (!sq1) {
        fe_sub(xm, nJ, x1);  // -x1 - J
        fe_mul(xm, xm, one);
        fe_sq(t, xm);
        fe_add(gx, xm, J);
        fe_mul(gx, gx, t);
        fe_add(gx, gx, xm);
        fe_sqrt(y, gx);
        want_odd = false;
    }
    fe ny;
    fe_neg(ny, y);
    fe_cmov(y, ny, fe_isnegative(y) != want_odd);
    // Montgomery (xm, y) -> Edwards: xe = c1 xm / y, ye = (xm - 1) / (xm + 1); exceptional cases -> (0, 1)
    fe xp1, xm1, d, di, xe, ye;
    fe_add(xp1, xm, one);
    fe_sub(xm1, xm, one);
    fe_mul(d, y, xp1);
    const bool exc = !fe_isnonzero(d);
    fe_invert(di, d);  // 1 / (y (xm + 1))
    fe_mul(xe, xm, fe_elligator_c1());
    fe_mul(xe, xe, xp1);
    fe_mul(xe, xe, di);  // c1 xm / y
    fe_mul(ye, xm1, y);
    fe_mul(ye, ye, di);  // (xm - 1) / (xm + 1)
    fe z;
    fe_0(z);
    fe_cmov(xe, z, exc);
    fe_cmov(ye, one, exc);
    q.X = xe;
    q.Y = ye;
    fe_1(q.Z);
    fe_mul(q.T, xe, ye);
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
