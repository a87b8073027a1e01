// BLS12-381 hash-to-curve (RFC 9380 section 8.8: BLS12381G1_XMD:SHA-256_SSWU_RO_ / BLS12381G2_...), one message
// per lane: expand_message_xmd (SHA-256) -> hash_to_field -> simplified SWU on the isogenous curve -> isogeny map
// -> add the two points -> clear the cofactor.
//
// Replaces G1Elt.Hash / G2Elt.Hash of the adapters (kilic/g1.go:161-170, g2.go: HashToCurve(msg, dst) in the
// external backend) -- the step before the pairing check in sign/bls Verify (bls.go:87-88).  The isogeny
// constants are derived by tools/derive_bls12381_isogenies.py and pinned by the reference's drand fixtures.
#pragma once
#include "bls12381.cuh"
#include "bls12381_h2c_params.h"
#include "sha256.cuh"

namespace kyb {
namespace bls {

using HC = Bls12381H2c;
using kyb::DstArg;  // hd.h

// uniform_bytes = expand_message_xmd(msg, DST, 32 * ELL) as big-endian words out[8 * ELL]
template <int ELL>
KYB_HD_NOINLINE void expand_message_xmd(uint32_t (&out)[8 * ELL], const uint8_t* msg, size_t msg_len, const DstArg& dst) {
    Sha256 c;
    c.init();
    for (int i = 0; i < 64; i++) c.put(0);  // Z_pad: one zero block
    c.update(msg, msg_len);
    const uint32_t nbytes = 32 * ELL;
    c.put((uint8_t)(nbytes >> 8));
    c.put((uint8_t)nbytes);
    c.put(0);
    c.update(dst.b, dst.len);
    c.put((uint8_t)dst.len);
    c.finish();
    uint32_t b0[8], prev[8];
    for (int i = 0; i < 8; i++) b0[i] = c.h[i];
    for (int i = 0; i < 8; i++) prev[i] = 0;
#pragma unroll 1
    for (int blk = 1; blk <= ELL; blk++) {
        uint32_t x[8];
        for (int i = 0; i < 8; i++) x[i] = blk == 1 ? b0[i] : (b0[i] ^ prev[i]);
        c.init();
        c.update_words_be(x, 8);
        c.put((uint8_t)blk);
        c.update(dst.b, dst.len);
        c.put((uint8_t)dst.len);
        c.finish();
        for (int i = 0; i < 8; i++) {
            prev[i] = c.h[i];
            out[8 * (blk - 1) + i] = c.h[i];
        }
    }
}
// OS2IP(64 bytes) mod p: be[0..16) are the 16 big-endian words of the 512-bit integer
KYB_HD void fp_from_be512(fp& r, const uint32_t* be) {
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int k = 0; k < 12; k++) lo[k] = be[15 - k];  // low 384 bits, little-endian words
#pragma unroll
    for (int k = 0; k < 12; k++) hi[k] = k < 4 ? be[3 - k] : 0;  // top 128 bits
    fp a, b, c;
    fp_from_words<FC>(a, lo);
    fp_from_words<FC>(b, hi);
    fp_const(c, HC::TWO384);
    fp_mul(b, b, c);
    fp_add(r, a, b);
}
KYB_HD bool fp_sgn0(const fp& a) {
    uint32_t w[12];
    fp_to_words<FC>(w, a);
    return w[0] & 1;
}
KYB_HD bool fp2_sgn0(const fp2& a) {
    uint32_t w0[12], w1[12];
    fp_to_words<FC>(w0, a.c0);
    fp_to_words<FC>(w1, a.c1);
    uint32_t z = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) z |= w0[k];
    return (w0[0] & 1) | ((z == 0) & (w1[0] & 1));
}

// ------------------------------------------------------------------ G1
KYB_HD void g1_curve_rhs(fp& g, const fp& x) {  // x^3 + A x + B on the isogenous curve
    fp a, b, t;
    fp_const(a, HC::G1_A);
    fp_const(b, HC::G1_B);
    fp_sqr(t, x);
    fp_add(t, t, a);
    fp_mul(t, t, x);
    fp_add(g, t, b);
}
// map_to_curve_simple_swu on E1' in the straight-line form of RFC 9380 appendix F.2 with sqrt_ratio for p = 3 mod 4
// (F.2.1.2): ONE exponentiation decides whether gx1 is a square AND yields the root in either case -- y1 =
// u v (u v^3)^((p-3)/4) is sqrt(u / v) when that is a square and y1 sqrt(-Z) is sqrt(Z u / v) when it is not.  Round 1's
// version tried x1, and took a SECOND 379-bit power for x2 where gx1 was no square: half the lanes of a wave need it, so
// every wave paid both (4 powers per hash_to_curve, 2 now: bls12381_hash_g1_kernel 3.3 -> 2.3 ms per 2^16).  Same map,
// same bytes (the RFC's vectors: tests/test_host_harness_bls12381.py, tests/test_gpu_bls12381.py).
KYB_HD_NOINLINE void g1_sswu(fp& x, fp& y, const fp& u) {
    fp A, B, Z, one, tv1, tv2, tv3, tv4, tv5, tv6, y1, y2, t, uv;
    fp_const(A, HC::G1_A);
    fp_const(B, HC::G1_B);
    fp_const(Z, HC::G1_Z);
    fp_one(one);
    fp_sqr(tv1, u);
    fp_mul(tv1, Z, tv1);      // Z u^2
    fp_sqr(tv2, tv1);
    fp_add(tv2, tv2, tv1);    // Z^2 u^4 + Z u^2
    fp_add(tv3, tv2, one);
    fp_mul(tv3, B, tv3);      // B (tv2 + 1): numerator of x1
    fp_neg(tv4, tv2);
    fp_cmov(tv4, Z, fp_is_zero(tv2));
    fp_mul(tv4, A, tv4);      // denominator of x1: -A tv2 (A Z in the exceptional case)
    fp_sqr(tv2, tv3);
    fp_sqr(tv6, tv4);
    fp_mul(tv5, A, tv6);
    fp_add(tv2, tv2, tv5);
    fp_mul(tv2, tv2, tv3);    // tv3^3 + A tv3 tv4^2
    fp_mul(tv6, tv6, tv4);    // tv4^3
    fp_mul(tv5, B, tv6);
    fp_add(tv2, tv2, tv5);    // gx1 = tv2 / tv6
    fp_mul(x, tv1, tv3);      // numerator of x2 = Z u^2 x1
    // sqrt_ratio(tv2, tv6)
    fp_sqr(t, tv6);
    fp_mul(uv, tv2, tv6);
    fp_mul(t, t, uv);         // u v^3
    fp_pow_words<FC>(y1, t, FC::PM3D4, FC::SQRT_BITS);
    fp_mul(y1, y1, uv);
    fp_const(t, HC::G1_SQRT_NEG_Z);
    fp_mul(y2, y1, t);
    fp_sqr(t, y1);
    fp_mul(t, t, tv6);
    const bool is_sq = fp_eq(t, tv2);
    fp_cmov(y1, y2, !is_sq);
    fp_mul(y, tv1, u);
    fp_mul(y, y, y1);         // the root for x2: Z u^3 sqrt(Z gx1)... = sqrt(gx2)
    fp_cmov(x, tv3, is_sq);
    fp_cmov(y, y1, is_sq);
    fp ny;
    fp_neg(ny, y);
    fp_cmov(y, ny, fp_sgn0(u) != fp_sgn0(y));
    fp_inv(t, tv4);
    fp_mul(x, x, t);
}
template <int LEN>
KYB_HD void fp_horner(fp& r, const uint32_t (&c)[LEN][FC::NWORDS], const fp& x) {
    fp acc, k;
    fp_const(acc, c[LEN - 1]);
#pragma unroll 1
    for (int i = LEN - 2; i >= 0; i--) {
        fp_mul(acc, acc, x);
        fp_const(k, c[i]);
        fp_add(acc, acc, k);
    }
    r = acc;
}
// 11-isogeny E1' -> E as a Jacobian point (no inversion): X/Z^2 = xn/xd, Y/Z^3 = y yn/yd with Z = xd yd
KYB_HD_NOINLINE void g1_iso_map(g1_jac& q, const fp& x, const fp& y) {
    fp xn, xd, yn, yd, t;
    fp_horner<12>(xn, HC::G1_XNUM, x);
    fp_horner<11>(xd, HC::G1_XDEN, x);
    fp_horner<16>(yn, HC::G1_YNUM, x);
    fp_horner<16>(yd, HC::G1_YDEN, x);
    fp_mul(q.Z, xd, yd);
    fp_mul(t, yd, q.Z);
    fp_mul(q.X, xn, t);  // xn xd yd^2
    fp_sqr(t, xd);
    fp_mul(t, t, q.Z);   // xd^3 yd
    fp_mul(t, t, yd);    // xd^3 yd^2
    fp_mul(t, t, yn);
    fp_mul(q.Y, t, y);
}
// G1Elt.Hash: out = hash_to_curve(msg, dst) as a 48-byte compressed point
KYB_HD_NOINLINE void hash_g1_point(g1_jac& r, const uint8_t* msg, size_t msg_len, const DstArg& dst);
KYB_HD int hash_g1_wire(uint8_t* out, const uint8_t* msg, size_t msg_len, const DstArg& dst) {
    g1_jac r;
    hash_g1_point(r, msg, msg_len, dst);
    g1_aff a;
    jac_to_aff(a, r);
    g1_encode(out, a);
    return ST_OK;
}

// hash_to_curve(msg, dst) as a Jacobian point (no encoding): shared by hash_g1_wire and verify_g1_wire
KYB_HD_NOINLINE void hash_g1_point(g1_jac& r, const uint8_t* msg, size_t msg_len, const DstArg& dst) {
    uint32_t ub[32];
    expand_message_xmd<4>(ub, msg, msg_len, dst);
    g1_jac q0, q1;
    fp u, x, y;
    fp_from_be512(u, ub);
    g1_sswu(x, y, u);
    g1_iso_map(q0, x, y);
    fp_from_be512(u, ub + 16);
    g1_sswu(x, y, u);
    g1_iso_map(q1, x, y);
    jac_add(r, q0, q1);
#ifdef KYB_BLS_PACKED_LADDER
    jac_mul_u64(r, r, 0xd201000000010001ull);
#else
    // clear_cofactor = [h_eff] on lazy limbs (jac_lazy.cuh: doublings and mixed additions, so the sum goes through its
    // affine form first -- one division-step inversion against the 63 doublings it feeds)
    {
        g1_aff a;
        jac_to_aff(a, r);
        jaclz_mul_u64_aff_t<LzFpN<FC>>(r, a, 0xd201000000010001ull);
    }
#endif
}

// ------------------------------------------------------------------ G2
KYB_HD void g2_curve_rhs(fp2& g, const fp2& x) {
    fp2 a, b, t;
    fp2_load_const<TC>(a, HC::G2_A);
    fp2_load_const<TC>(b, HC::G2_B);
    fp2_sqr_c(t, x);
    fp2_add(t, t, a);
    fp2_mul_c(t, t, x);
    fp2_add(g, t, b);
}
KYB_HD_NOINLINE void g2_sswu(fp2& x, fp2& y, const fp2& u) {
    fp2 z, u2, tv1, t, x1, gx, one;
    fp2_load_const<TC>(z, HC::G2_Z);
    fp2_one(one);
    fp2_sqr_c(u2, u);
    fp2_mul_c(t, z, u2);
    fp2_sqr_c(tv1, t);
    fp2_add(tv1, tv1, t);
    if (fp2_is_zero(tv1)) {
        fp2_load_const<TC>(x1, HC::G2_B_OVER_ZA);
    } else {
        fp2 k;
        fp2_load_const<TC>(k, HC::G2_NEG_B_OVER_A);
        fp2_inv(x1, tv1);
        fp2_add(x1, x1, one);
        fp2_mul_c(x1, x1, k);
    }
    g2_curve_rhs(gx, x1);
    // g(x1) is a square in Fp2 exactly when its norm n1 is one in Fp: s1 = n1^((p+1)/4) decides it (s1^2 = n1) -- and when it
    // is not (s1^2 = -n1), the norm root of g(x2) = (Z u^2)^3 g(x1) is s1 sqrt(-N(Z)^3) N(u)^3 without another power.
    // Either way ONE more power finishes the root (fp2_sqrt_from_norm_root).  Round 1 ran the whole two-power square
    // root on g(x1) and again on g(x2): half the lanes of a wave need the second, so every wave paid four.
    {
        fp n1, s1, tt, nu, k;
        fp_sqr(n1, gx.c0);
        fp_sqr(tt, gx.c1);
        fp_add(n1, n1, tt);
        fp_pow_words<FC>(s1, n1, FC::SQRT_EXP, FC::SQRT_BITS);
        fp_sqr(tt, s1);
        const bool sq = fp_eq(tt, n1);
        fp2 x2, gx2;
        fp2_mul_c(x2, t, x1);
        g2_curve_rhs(gx2, x2);
        fp_sqr(nu, u.c0);
        fp_sqr(tt, u.c1);
        fp_add(nu, nu, tt);   // N(u)
        fp_sqr(tt, nu);
        fp_mul(nu, tt, nu);   // N(u)^3
        fp_const(k, HC::G2_SQRT_NEG_NZ3);
        fp_mul(k, k, nu);
        fp_mul(k, k, s1);     // the norm root of g(x2) when g(x1) is no square
        fp_cmov(s1, k, !sq);
        fp2_cmov(gx, gx2, !sq);
        fp2_cmov(x1, x2, !sq);
        fp2_sqrt_from_norm_root(y, gx, s1);
    }
    fp2 ny;
    fp2_neg(ny, y);
    fp2_cmov(y, ny, fp2_sgn0(u) != fp2_sgn0(y));
    x = x1;
}
template <int LEN>
KYB_HD void fp2_horner(fp2& r, const uint32_t (&c)[LEN][2][FC::NWORDS], const fp2& x) {
    fp2 acc, k;
    fp2_load_const<TC>(acc, c[LEN - 1]);
#pragma unroll 1
    for (int i = LEN - 2; i >= 0; i--) {
        fp2_mul_c(acc, acc, x);
        fp2_load_const<TC>(k, c[i]);
        fp2_add(acc, acc, k);
    }
    r = acc;
}
KYB_HD_NOINLINE void g2_iso_map(g2_jac& q, const fp2& x, const fp2& y) {
    fp2 xn, xd, yn, yd, t;
    fp2_horner<4>(xn, HC::G2_XNUM, x);
    fp2_horner<3>(xd, HC::G2_XDEN, x);
    fp2_horner<4>(yn, HC::G2_YNUM, x);
    fp2_horner<4>(yd, HC::G2_YDEN, x);
    fp2_mul_c(q.Z, xd, yd);
    fp2_mul_c(t, yd, q.Z);
    fp2_mul_c(q.X, xn, t);
    fp2_sqr_c(t, xd);
    fp2_mul_c(t, t, q.Z);
    fp2_mul_c(t, t, yd);
    fp2_mul_c(t, t, yn);
    fp2_mul_c(q.Y, t, y);
}
// psi = twist o Frobenius o untwist on a Jacobian point: (conj X cx, conj Y cy, conj Z)
KYB_HD void g2_psi(g2_jac& r, const g2_jac& p) {
    fp2 cx, cy;
    fp2_load_const<TC>(cx, CC::PSI_CX);
    fp2_load_const<TC>(cy, CC::PSI_CY);
    fp2_conj(r.X, p.X);
    fp2_mul_c(r.X, r.X, cx);
    fp2_conj(r.Y, p.Y);
    fp2_mul_c(r.Y, r.Y, cy);
    fp2_conj(r.Z, p.Z);
}
// [x] P for the (negative) curve parameter x = -X_ABS
KYB_HD void g2_mul_x(g2_jac& r, const g2_jac& p) {
    jac_mul_u64(r, p, CC::X_ABS);
    jac_neg(r, r);
}
// clear_cofactor_bls12381_g2 (RFC 9380 appendix G.3): [x^2 - x - 1] P + [x - 1] psi(P) + psi^2(2P)
KYB_HD_NOINLINE void g2_clear_cofactor(g2_jac& q, const g2_jac& p) {
    g2_jac t1, t2, t3, n;
    g2_mul_x(t1, p);
    g2_psi(t2, p);
    jac_dbl(t3, p);
    g2_psi(t3, t3);
    g2_psi(t3, t3);
    jac_neg(n, t2);
    jac_add(t3, t3, n);  // psi^2(2P) - psi(P)
    jac_add(t2, t1, t2);
    g2_mul_x(t2, t2);    // x (xP + psi P)
    jac_add(t3, t3, t2);
    jac_neg(n, t1);
    jac_add(t3, t3, n);
    jac_neg(n, p);
    jac_add(q, t3, n);
}
// hash_to_curve(msg, dst) on G2 as a Jacobian point: shared by hash_g2_wire and verify_g2_wire
KYB_HD_NOINLINE void hash_g2_point(g2_jac& r, const uint8_t* msg, size_t msg_len, const DstArg& dst) {
    uint32_t ub[64];
    expand_message_xmd<8>(ub, msg, msg_len, dst);
    g2_jac q0, q1;
    fp2 u, x, y;
    fp_from_be512(u.c0, ub);
    fp_from_be512(u.c1, ub + 16);
    g2_sswu(x, y, u);
    g2_iso_map(q0, x, y);
    fp_from_be512(u.c0, ub + 32);
    fp_from_be512(u.c1, ub + 48);
    g2_sswu(x, y, u);
    g2_iso_map(q1, x, y);
    jac_add(r, q0, q1);
    g2_clear_cofactor(r, r);
}
// The operand side of a whole sign/bls Verify for the batch engine: unmarshal key and signature with the adapter's
// checks, hash the message, and hand the two pairs of the product check
//   signatures on G1:  e(H(m), X) e(-sig, G2.Base()) == 1        signatures on G2:  e(G1.Base(), sig) e(-X, H(m)) == 1
// to the cooperative tower machine (bls12381_pair.hip CHECK program) as affine points.  Returns the status; on
// rejection the points are left at infinity.
KYB_HD int verify_g1_operands(g1_aff& a1, g2_aff& a2, g1_aff& b1, g2_aff& b2, const uint8_t* pk96, const uint8_t* msg,
                              size_t msg_len, const DstArg& dst, const uint8_t* sig48, uint32_t flags) {
    int st = g2_decode_f(a2, pk96, flags, 0);
    const int st2 = g1_decode_f(b1, sig48, flags, 1);
    if (st == ST_OK) st = st2;
    fp_zero(a1.x);
    fp_zero(a1.y);
    a1.inf = true;
    fp2_zero(b2.x);
    fp2_zero(b2.y);
    b2.inf = true;
    if (st != ST_OK) return st;
    g1_jac hj;
    hash_g1_point(hj, msg, msg_len, dst);
    jac_to_aff(a1, hj);
    fp2_load_const<TC>(b2.x, CC::G2X);
    fp2_load_const<TC>(b2.y, CC::G2Y);
    b2.inf = false;
    fp_neg(b1.y, b1.y);
    return ST_OK;
}
KYB_HD int verify_g2_operands(g1_aff& a1, g2_aff& a2, g1_aff& b1, g2_aff& b2, const uint8_t* pk48, const uint8_t* msg,
                              size_t msg_len, const DstArg& dst, const uint8_t* sig96, uint32_t flags) {
    int st = g1_decode_f(b1, pk48, flags, 0);
    const int st2 = g2_decode_f(a2, sig96, flags, 1);
    if (st == ST_OK) st = st2;
    fp_zero(a1.x);
    fp_zero(a1.y);
    a1.inf = true;
    fp2_zero(b2.x);
    fp2_zero(b2.y);
    b2.inf = true;
    if (st != ST_OK) return st;
    g2_jac hj;
    hash_g2_point(hj, msg, msg_len, dst);
    jac_to_aff(b2, hj);
    fp_const(a1.x, CC::G1X);
    fp_const(a1.y, CC::G1Y);
    a1.inf = false;
    fp_neg(b1.y, b1.y);
    return ST_OK;
}
KYB_HD int hash_g2_wire(uint8_t* out, const uint8_t* msg, size_t msg_len, const DstArg& dst) {
    g2_jac r;
    hash_g2_point(r, msg, msg_len, dst);
    g2_aff a;
    jac_to_aff(a, r);
    g2_encode(out, a);
    return ST_OK;
}

}  // namespace bls
}  // namespace kyb
