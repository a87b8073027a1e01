// Ed25519 group operations on the device (one point per lane).
//
// Replaces the reference's group/edwards25519/ge.go point types and formulas:
// extended (X:Y:Z:T) ge.go:22, projective ge.go:16, completed ge.go:26, cached
// ge.go:32, precomputed ge.go:28; Double ge.go:42, Add/Sub ge.go:183/200,
// MixedAdd ge.go:217, ToBytes ge.go:99, FromBytes ge.go:110.
// The formulas are the a=-1 twisted-Edwards ones of Hisil-Wong-Carter-Dawson
// (add-2008-hwcd-3 with a cached second operand, dbl-2008-hwcd).
#pragma once
#include "fe25519.cuh"

namespace kyb {

struct ge_p2 { fe X, Y, Z; };
struct ge_p3 { fe X, Y, Z, T; };
struct ge_p1p1 { fe X, Y, Z, T; };
struct ge_cached { fe YpX, YmX, Z, T2d; };
struct ge_precomp { fe ypx, ymx, xy2d; };

// Curve constants as radix-2^25.5 limbs.  They are derived on the host from
// d = -121665/121666 by kyber_amd/csrc/gen_consts.py (not copied from
// const.go:34-44) and checked by tests/test_constants.py.
#include "ed25519_consts.inc"

KYB_DEV void ge_p3_0(ge_p3& h) {
    fe_0(h.X);
    fe_1(h.Y);
    fe_1(h.Z);
    fe_0(h.T);
}
KYB_DEV void ge_cached_0(ge_cached& c) {
    fe_1(c.YpX);
    fe_1(c.YmX);
    fe_1(c.Z);
    fe_0(c.T2d);
}
KYB_DEV void ge_p1p1_to_p2(ge_p2& r, const ge_p1p1& p) {
    fe_mul(r.X, p.X, p.T);
    fe_mul(r.Y, p.Y, p.Z);
    fe_mul(r.Z, p.Z, p.T);
}
KYB_DEV void ge_p1p1_to_p3(ge_p3& r, const ge_p1p1& p) {
    fe_mul(r.X, p.X, p.T);
    fe_mul(r.Y, p.Y, p.Z);
    fe_mul(r.Z, p.Z, p.T);
    fe_mul(r.T, p.X, p.Y);
}
KYB_DEV void ge_p3_to_cached(ge_cached& r, const ge_p3& p) {
    fe_add(r.YpX, p.Y, p.X);
    fe_sub(r.YmX, p.Y, p.X);
    r.Z = p.Z;
    fe_mul(r.T2d, p.T, fe_d2());
}
// r = 2 * (X:Y:Z)   (4 squarings)
KYB_DEV void ge_dbl(ge_p1p1& r, const fe& X, const fe& Y, const fe& Z) {
    fe t0;
    fe_sq(r.X, X);
    fe_sq(r.Z, Y);
    fe_sq2(r.T, Z);
    fe_add(r.Y, X, Y);
    fe_sq(t0, r.Y);
    fe_add(r.Y, r.Z, r.X);
    fe_sub(r.Z, r.Z, r.X);
    fe_sub(r.X, t0, r.Y);
    fe_sub(r.T, r.T, r.Z);
}
// r = p + q  (q cached)   4 multiplications
KYB_DEV void ge_add(ge_p1p1& r, const ge_p3& p, const ge_cached& q) {
    fe t0;
    fe_add(r.X, p.Y, p.X);
    fe_sub(r.Y, p.Y, p.X);
    fe_mul(r.Z, r.X, q.YpX);
    fe_mul(r.Y, r.Y, q.YmX);
    fe_mul(r.T, q.T2d, p.T);
    fe_mul(r.X, p.Z, q.Z);
    fe_add(t0, r.X, r.X);
    fe_sub(r.X, r.Z, r.Y);
    fe_add(r.Y, r.Z, r.Y);
    fe_add(r.Z, t0, r.T);
    fe_sub(r.T, t0, r.T);
}
// r = p + q  (q affine precomputed)   3 multiplications
KYB_DEV void ge_madd(ge_p1p1& r, const ge_p3& p, const ge_precomp& q) {
    fe t0;
    fe_add(r.X, p.Y, p.X);
    fe_sub(r.Y, p.Y, p.X);
    fe_mul(r.Z, r.X, q.ypx);
    fe_mul(r.Y, r.Y, q.ymx);
    fe_mul(r.T, q.xy2d, p.T);
    fe_add(t0, p.Z, p.Z);
    fe_sub(r.X, r.Z, r.Y);
    fe_add(r.Y, r.Z, r.Y);
    fe_add(r.Z, t0, r.T);
    fe_sub(r.T, t0, r.T);
}
// -q for a cached operand: swap (Y+X, Y-X), negate 2dT
KYB_DEV void ge_cached_cneg(ge_cached& c, bool neg) {
    fe_cswap(c.YpX, c.YmX, neg);
    fe nt;
    fe_neg(nt, c.T2d);
    fe_cmov(c.T2d, nt, neg);
}
KYB_DEV void ge_precomp_cneg(ge_precomp& c, bool neg) {
    fe_cswap(c.ypx, c.ymx, neg);
    fe nt;
    fe_neg(nt, c.xy2d);
    fe_cmov(c.xy2d, nt, neg);
}

// 32-byte canonical encoding (as 8 words) of X/Z, Y/Z given zinv = 1/Z
KYB_DEV void ge_encode_with_zinv(uint32_t w[8], const fe& X, const fe& Y, const fe& zinv) {
    fe x, y;
    fe_mul(x, X, zinv);
    fe_mul(y, Y, zinv);
    fe_towords(w, y);
    w[7] ^= (uint32_t)fe_isnegative(x) << 31;
}
KYB_DEV void ge_p3_towords(uint32_t w[8], const ge_p3& p) {
    fe zi;
    fe_invert(zi, p.Z);
    ge_encode_with_zinv(w, p.X, p.Y, zi);
}
// Decode with the reference's exact acceptance rules (ge.go:110-150): y taken
// mod 2^255 (non-canonical accepted), failure only when no square root exists,
// x = 0 with the sign bit set is accepted.
KYB_DEV bool ge_p3_fromwords(ge_p3& p, const uint32_t w[8]) {
    fe u, v, v3, vxx, check;
    fe_fromwords(p.Y, w);
    fe_1(p.Z);
    fe_sq(u, p.Y);
    fe_mul(v, u, fe_d());
    fe_sub(u, u, p.Z);  // u = y^2 - 1
    fe_add(v, v, p.Z);  // v = d y^2 + 1
    fe_sq(v3, v);
    fe_mul(v3, v3, v);  // v^3
    fe_sq(p.X, v3);
    fe_mul(p.X, p.X, v);
    fe_mul(p.X, p.X, u);  // u v^7
    fe_pow22523(p.X, p.X);
    fe_mul(p.X, p.X, v3);
    fe_mul(p.X, p.X, u);  // u v^3 (u v^7)^((p-5)/8)
    fe_sq(vxx, p.X);
    fe_mul(vxx, vxx, v);
    fe_sub(check, vxx, u);
    bool ok = true;
    if (fe_isnonzero(check)) {
        fe_add(check, vxx, u);
        ok = !fe_isnonzero(check);
        fe xs;
        fe_mul(xs, p.X, fe_sqrtm1());
        p.X = xs;
    }
    const bool sign = (w[7] >> 31) != 0;
    fe nx;
    fe_neg(nx, p.X);
    fe_cmov(p.X, nx, fe_isnegative(p.X) != sign);
    fe_mul(p.T, p.X, p.Y);
    return ok;
}

// Signed radix-16 digits of a 256-bit little-endian scalar, the recoding of
// ge.go:374-390 / 453-467: e[0..62] in [-8,8]; e[63] in [0,16].
// `full256` selects the all-bits semantics of geScalarMultVartime
// (ge_mult_vartime.go:11): e[63] is carried once more into e[64] in {0,1}.
// Otherwise a top digit above 8 is dropped, reproducing selectCached's
// "no table entry matches" behaviour for scalars >= 2^255 (ge.go:424-427).
KYB_DEV void recode16(int8_t e[65], const uint32_t a[8], bool full256) {
    int carry = 0;
#pragma unroll
    for (int i = 0; i < 64; i++) {
        int d = (int)((a[i >> 3] >> ((i & 7) * 4)) & 15) + carry;
        if (i < 63) {
            carry = (d + 8) >> 4;
            d -= carry << 4;
        }
        e[i] = (int8_t)d;
    }
    int top = e[63];
    int c2 = 0;
    if (full256) {
        c2 = (top + 8) >> 4;
        top -= c2 << 4;
    } else if (top > 8) {
        top = 0;
    }
    e[63] = (int8_t)top;
    e[64] = (int8_t)c2;
}

// The integer the recoding above makes geScalarMult / geScalarMultBase multiply by, as a 256-bit magnitude and a sign
// (for the callers that cut their own digits: the MSM).  With e[63] = (a + 0x0888..8) >> 252 the top digit of the
// signed radix-16 recoding, a scalar with e[63] > 8 loses that digit: a - e[63] 2^252, which is a mod 2^252 when the
// recoding carried nothing into the top digit and (a mod 2^252) - 2^252 -- negative -- when it did.  Returns true
// when the result is negative (k then holds its magnitude).  Same function as oracle effective_scalar_consttime.
KYB_DEV bool ed_effective_scalar(uint32_t (&k)[8]) {
    uint32_t c = 0, t7 = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint64_t s = (uint64_t)k[j] + (j < 7 ? 0x88888888u : 0x08888888u) + c;
        t7 = (uint32_t)s;
        c = (uint32_t)(s >> 32);
    }
    const uint32_t e63 = (c << 4) | (t7 >> 28);
    if (e63 <= 8) return false;
    const uint32_t carried = e63 - (k[7] >> 28);  // 0 or 1
    k[7] &= 0x0fffffffu;
    if (!carried) return false;
    uint32_t b = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint64_t d = (uint64_t)(j == 7 ? 0x10000000u : 0u) - k[j] - b;
        k[j] = (uint32_t)d;
        b = (uint32_t)(d >> 63);
    }
    return true;
}

}  // namespace kyb
