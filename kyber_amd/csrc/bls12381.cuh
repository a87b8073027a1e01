// BLS12-381 G1 / G2 / GT device library: ZCash (de)serialisation with the reference's acceptance
// rules, subgroup checks, scalar multiplication, optimal ate pairing.
//
// Replaces the external arithmetic behind pairing/bls12381/kilic (adapter call sites:
// kilic/g1.go:110-131 Mul / MarshalBinary / UnmarshalBinary, kilic/g2.go, kilic/gt.go:85-117,
// kilic/suite.go:57-75 ValidatePairing / Pair) -- the arithmetic itself is
// github.com/kilic/bls12-381 v0.1.0 (go.mod:8), not in the reference tree; what is implemented
// here is the published curve / encoding / pairing definition, validated
// against oracle/bls12381.py.
#pragma once
#include "bls12381_params.h"
#include "curve.cuh"
#include "jac_lazy.cuh"

namespace kyb {
namespace bls {

using FC = Bls12381Fp;
using TC = Bls12381Tower;
using CC = Bls12381Curve;
using fp = Fp<FC>;
using fp2 = Fp2<TC>;
using fp6 = Fp6<TC>;
using fp12 = Fp12<TC>;
using g1_aff = Aff<fp>;
using g2_aff = Aff<fp2>;
using g1_jac = Jac<fp>;
using g2_jac = Jac<fp2>;

constexpr int ST_OK = 0, ST_BAD_POINT = 1, ST_NOT_IN_SUBGROUP = 2;

KYB_HD void fp_const(fp& r, const uint32_t (&c)[FC::NWORDS]) {
#pragma unroll
    for (int l = 0; l < FC::NWORDS; l++) r.v[l] = c[l];
}

// y > (p - 1) / 2 on the canonical value
KYB_HD bool fp_is_larger(const fp& y) {
    uint32_t w[FC::NWORDS];
    fp_to_words<FC>(w, y);
    return words_gt<FC::NWORDS>(w, FC::HALF);
}
KYB_HD bool fp2_is_larger(const fp2& y) {
    const bool c1z = fp_is_zero(y.c1);
    return c1z ? fp_is_larger(y.c0) : fp_is_larger(y.c1);
}

// ------------------------------------------------------------- subgroup checks
// G1: P has order r  <=>  phi(P) = [-x^2] P, phi(x, y) = (beta x, y)   (Scott, eprint 2021/1130)
KYB_HD_NOINLINE bool g1_in_subgroup(const g1_aff& a) {
    g1_jac p, q;
#ifdef KYB_BLS_PACKED_LADDER
    jac_from_aff(p, a);
    jac_mul_u64(q, p, CC::X_ABS);
    jac_mul_u64(q, q, CC::X_ABS);  // x^2 P
#else
    // both 64-bit multiplications on lazy limbs (jac_lazy.cuh: doublings and MIXED additions only, so [x] P returns to
    // affine form in between -- one division-step inversion, a twelfth of the chain it feeds)
    {
        using LF = LzFpN<FC>;
        g1_aff a1;
        jaclz_mul_u64_aff_t<LF>(p, a, CC::X_ABS);
        jac_to_aff(a1, p);
        jaclz_mul_u64_aff_t<LF>(q, a1, CC::X_ABS);  // x^2 P (infinity when [x] P was: Z = 0 fails the comparison below)
    }
#endif
    fp beta, bx, ny, z2, z3, l, r;
    fp_const(beta, CC::BETA);
    fp_mul(bx, a.x, beta);
    fp_neg(ny, a.y);
    // -q == (beta x, y)  <=>  q.X = bx Z^2, q.Y = -y Z^3, Z != 0
    fp_sqr(z2, q.Z);
    fp_mul(z3, z2, q.Z);
    fp_mul(l, bx, z2);
    fp_mul(r, ny, z3);
    const bool ok = fp_eq(l, q.X) & fp_eq(r, q.Y) & !fp_is_zero(q.Z);
    return a.inf | ok;
}
// G2: psi(Q) = [x] Q, psi = twist o Frobenius o untwist
KYB_HD_NOINLINE bool g2_in_subgroup(const g2_aff& a) {
    g2_jac p, q;
#ifdef KYB_BLS_PACKED_LADDER
    jac_from_aff(p, a);
    jac_mul_u64(q, p, CC::X_ABS);  // |x| Q ; need psi(Q) = -q
#else
    // on lazy limbs (jac_lazy.cuh, fourteen 30-bit limbs per coefficient: the Fp2 formulas need R' / p > 2^14).  The doubling is
    // ~75 KB of straight line, more than the instruction cache -- and still 17 % ahead of the packed code's calls
    // (G2 UnmarshalBinary 2^20: 2.20 -> 2.57e7/s, profiles/r05_bls12381_g2_member_lazy_ab.jsonl)
    jaclz_mul_u64_aff<LzFp2<Limb30<FC>, TC>>(q, a, CC::X_ABS);
#endif
    fp2 cx, cy, px, py, z2, z3, l, r;
    fp2_load_const<TC>(cx, CC::PSI_CX);
    fp2_load_const<TC>(cy, CC::PSI_CY);
    fp2_conj(px, a.x);
    fp2_mul_c(px, px, cx);
    fp2_conj(py, a.y);
    fp2_mul_c(py, py, cy);
    fp2_neg(py, py);
    fp2_sqr_c(z2, q.Z);
    fp2_mul_c(z3, z2, q.Z);
    fp2_mul_c(l, px, z2);
    fp2_mul_c(r, py, z3);
    const bool ok = fp2_eq(l, q.X) & fp2_eq(r, q.Y) & !fp2_is_zero(q.Z);
    return a.inf | ok;
}

// ------------------------------------------------------------------ decoding
// 48-byte ZCash compressed G1 (kilic/g1.go:127-131 FromCompressed + subgroup check).
KYB_HD_NOINLINE int g1_decode(g1_aff& a, const uint8_t* in, bool check_subgroup) {
    uint32_t w[12];
    words_from_be<12>(w, in);
    const uint32_t top = w[11] >> 29;
    const bool c = top & 4, inf = top & 2, s = top & 1;
    w[11] &= 0x1fffffffu;
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) any |= w[k];
    fp_zero(a.x);
    fp_zero(a.y);
    a.inf = true;
    if (!c) return ST_BAD_POINT;
    if (inf) return (s || any) ? ST_BAD_POINT : ST_OK;
    if (!fp_words_lt_p<FC>(w)) return ST_BAD_POINT;
    fp x, y, rhs, b, t;
    fp_from_words<FC>(x, w);
    fp_const(b, CC::B1);
    fp_sqr(rhs, x);
    fp_mul(rhs, rhs, x);
    fp_add(rhs, rhs, b);
    fp_pow_words<FC>(y, rhs, FC::SQRT_EXP, FC::SQRT_BITS);
    fp_sqr(t, y);
    if (!fp_eq(t, rhs)) return ST_BAD_POINT;
    fp_neg(t, y);
    fp_cmov(y, t, fp_is_larger(y) != s);
    a.x = x;
    a.y = y;
    a.inf = false;
    if (check_subgroup && !g1_in_subgroup(a)) return ST_NOT_IN_SUBGROUP;
    return ST_OK;
}

// Square root in Fp2 (p = 3 mod 4) with two base-field exponentiations; false if none exists.
// the second half of fp2_sqrt: given s with s^2 = norm(a) (either root), the root of a -- one 379-bit power
KYB_HD_NOINLINE bool fp2_sqrt_from_norm_root(fp2& r, const fp2& a, const fp& s) {
    fp t, u, c, c2, h, inv2;
    fp_const(inv2, FC::INV2);
    fp_add(t, a.c0, s);
    fp_mul(t, t, inv2);
    fp_cmov(t, a.c0, fp_is_zero(a.c1));
    fp_pow_words<FC>(u, t, FC::PM3D4, FC::SQRT_BITS);  // t^((p-3)/4)
    fp_mul(c, u, t);                                   // t^((p+1)/4)
    fp_sqr(c2, c);
    const bool qr = fp_eq(c2, t);  // chi(t) = +1 ; otherwise c^2 = -t and 1/c = -u
    fp_mul(h, a.c1, inv2);
    fp_mul(h, h, u);  // a1 / (2c) up to the sign chi
    fp2 x;
    if (qr) {
        x.c0 = c;
        x.c1 = h;
    } else {
        fp_neg(x.c0, h);
        x.c1 = c;
    }
    fp2 chk;
    fp2_sqr(chk, x);
    r = x;
    return fp2_eq(chk, a);
}
KYB_HD_NOINLINE bool fp2_sqrt(fp2& r, const fp2& a) {
    fp n, s, t;
    fp_sqr(n, a.c0);
    fp_sqr(t, a.c1);
    fp_add(n, n, t);
    fp_pow_words<FC>(s, n, FC::SQRT_EXP, FC::SQRT_BITS);  // sqrt of the norm (if it is a square)
    return fp2_sqrt_from_norm_root(r, a, s);
}

// 96-byte ZCash compressed G2: x.c1 || x.c0 big-endian, flags in the first byte.
KYB_HD_NOINLINE int g2_decode(g2_aff& a, const uint8_t* in, bool check_subgroup) {
    uint32_t w1[12], w0[12];
    words_from_be<12>(w1, in);
    words_from_be<12>(w0, in + 48);
    const uint32_t top = w1[11] >> 29;
    const bool c = top & 4, inf = top & 2, s = top & 1;
    w1[11] &= 0x1fffffffu;
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) any |= w1[k] | w0[k];
    fp2_zero(a.x);
    fp2_zero(a.y);
    a.inf = true;
    if (!c) return ST_BAD_POINT;
    if (inf) return (s || any) ? ST_BAD_POINT : ST_OK;
    if (!fp_words_lt_p<FC>(w1) || !fp_words_lt_p<FC>(w0)) return ST_BAD_POINT;
    fp2 x, y, rhs, b, t;
    fp_from_words<FC>(x.c0, w0);
    fp_from_words<FC>(x.c1, w1);
    fp2_load_const<TC>(b, CC::B2);
    fp2_sqr_c(rhs, x);
    fp2_mul_c(rhs, rhs, x);
    fp2_add(rhs, rhs, b);
    if (!fp2_sqrt(y, rhs)) return ST_BAD_POINT;
    fp2_neg(t, y);
    fp2_cmov(y, t, fp2_is_larger(y) != s);
    a.x = x;
    a.y = y;
    a.inf = false;
    if (check_subgroup && !g2_in_subgroup(a)) return ST_NOT_IN_SUBGROUP;
    return ST_OK;
}

// ZCash uncompressed forms (x || y, 96 B for G1; x.c1 || x.c0 || y.c1 || y.c0, 192 B for G2; top bits of the first
// byte: compression 0, infinity, sort 0) -- the `_aff` inputs of SURVEY.md section 8(b): no square root.  With
// validate = false (KYB_F_TRUSTED_*) the curve-equation and subgroup checks are skipped as well.
constexpr int G1_WIRE_UNC = 96, G2_WIRE_UNC = 192;
KYB_HD int g1_decode_unc_inl(g1_aff& a, const uint8_t* in, bool validate, bool subgroup) {
    uint32_t wx[12], wy[12];
    words_from_be<12>(wx, in);
    words_from_be<12>(wy, in + 48);
    const uint32_t top = wx[11] >> 29;
    const bool c = top & 4, inf = top & 2, s = top & 1;
    wx[11] &= 0x1fffffffu;
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) any |= wx[k] | wy[k];
    fp_zero(a.x);
    fp_zero(a.y);
    a.inf = true;
    if (c || s) return ST_BAD_POINT;
    if (inf) return any ? ST_BAD_POINT : ST_OK;
    if (!fp_words_lt_p<FC>(wx) || !fp_words_lt_p<FC>(wy)) return ST_BAD_POINT;
    fp x, y;
    fp_from_words<FC>(x, wx);
    fp_from_words<FC>(y, wy);
    if (validate) {
        fp rhs, b, t;
        fp_const(b, CC::B1);
        fp_sqr(rhs, x);
        fp_mul(rhs, rhs, x);
        fp_add(rhs, rhs, b);
        fp_sqr(t, y);
        if (!fp_eq(t, rhs)) return ST_BAD_POINT;
    }
    a.x = x;
    a.y = y;
    a.inf = false;
    if (validate && subgroup && !g1_in_subgroup(a)) return ST_NOT_IN_SUBGROUP;
    return ST_OK;
}
KYB_HD_NOINLINE int g1_decode_unc(g1_aff& a, const uint8_t* in, bool validate, bool subgroup = true) {
    return g1_decode_unc_inl(a, in, validate, subgroup);
}
// the vouched-for uncompressed form with the body in the caller (msm.cuh's light decode kernel: no curve equation, no
// subgroup rule, nothing that would need the out-of-line routines' register budget)
KYB_HD int g1_decode_unc_trusted(g1_aff& a, const uint8_t* in) { return g1_decode_unc_inl(a, in, false, false); }
KYB_HD_NOINLINE int g2_decode_unc(g2_aff& a, const uint8_t* in, bool validate, bool subgroup = true) {
    uint32_t w[4][12];
    uint32_t any = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) words_from_be<12>(w[j], in + 48 * j);
    const uint32_t top = w[0][11] >> 29;
    const bool c = top & 4, inf = top & 2, s = top & 1;
    w[0][11] &= 0x1fffffffu;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int k = 0; k < 12; k++) any |= w[j][k];
    fp2_zero(a.x);
    fp2_zero(a.y);
    a.inf = true;
    if (c || s) return ST_BAD_POINT;
    if (inf) return any ? ST_BAD_POINT : ST_OK;
    bool lt = true;
#pragma unroll
    for (int j = 0; j < 4; j++) lt &= fp_words_lt_p<FC>(w[j]);
    if (!lt) return ST_BAD_POINT;
    fp2 x, y;
    fp_from_words<FC>(x.c1, w[0]);
    fp_from_words<FC>(x.c0, w[1]);
    fp_from_words<FC>(y.c1, w[2]);
    fp_from_words<FC>(y.c0, w[3]);
    if (validate) {
        fp2 rhs, b, t;
        fp2_load_const<TC>(b, CC::B2);
        fp2_sqr_c(rhs, x);
        fp2_mul_c(rhs, rhs, x);
        fp2_add(rhs, rhs, b);
        fp2_sqr_c(t, y);
        if (!fp2_eq(t, rhs)) return ST_BAD_POINT;
    }
    a.x = x;
    a.y = y;
    a.inf = false;
    if (validate && subgroup && !g2_in_subgroup(a)) return ST_NOT_IN_SUBGROUP;
    return ST_OK;
}
// Input decoding as selected by the call's flags for point argument `arg`.
KYB_HD int g1_decode_f(g1_aff& a, const uint8_t* in, uint32_t flags, int arg) {
    const bool validate = !flag_trusted(flags, arg);
    return (flags & FLAG_UNCOMPRESSED) ? g1_decode_unc(a, in, validate) : g1_decode(a, in, validate);
}
KYB_HD int g2_decode_f(g2_aff& a, const uint8_t* in, uint32_t flags, int arg) {
    const bool validate = !flag_trusted(flags, arg);
    return (flags & FLAG_UNCOMPRESSED) ? g2_decode_unc(a, in, validate) : g2_decode(a, in, validate);
}
// UnmarshalBinary of an unvouched-for G2 point minus its r-torsion test (flag rules, range, on the curve): for callers
// that decide the membership themselves (the pairing programs: bls12381_prep.hip)
KYB_HD int g2_decode_on_curve(g2_aff& a, const uint8_t* in, uint32_t flags) {
    return (flags & FLAG_UNCOMPRESSED) ? g2_decode_unc(a, in, true, false) : g2_decode(a, in, false);
}
KYB_HD size_t g1_wire_size(uint32_t flags) { return (flags & FLAG_UNCOMPRESSED) ? G1_WIRE_UNC : 48; }
KYB_HD size_t g2_wire_size(uint32_t flags) { return (flags & FLAG_UNCOMPRESSED) ? G2_WIRE_UNC : 96; }
// UnmarshalBinary proves r-torsion (kilic/g2.go: ZCash rules), so "decoded once" and "vouched for" are the same claim
constexpr bool g2_decode_proves_subgroup() { return true; }

// ------------------------------------------------------------------ encoding
KYB_HD_NOINLINE void g1_encode(uint8_t* out, const g1_aff& a) {
    uint32_t w[12];
    fp_to_words<FC>(w, a.x);
    uint32_t flags = 0x80000000u | (fp_is_larger(a.y) ? 0x20000000u : 0u);
    if (a.inf) {
#pragma unroll
        for (int k = 0; k < 12; k++) w[k] = 0;
        flags = 0xC0000000u;
    }
    w[11] |= flags;
    words_to_be<12>(out, w);
}
KYB_HD_NOINLINE void g2_encode(uint8_t* out, const g2_aff& a) {
    uint32_t w1[12], w0[12];
    fp_to_words<FC>(w1, a.x.c1);
    fp_to_words<FC>(w0, a.x.c0);
    uint32_t flags = 0x80000000u | (fp2_is_larger(a.y) ? 0x20000000u : 0u);
    if (a.inf) {
#pragma unroll
        for (int k = 0; k < 12; k++) w1[k] = w0[k] = 0;
        flags = 0xC0000000u;
    }
    w1[11] |= flags;
    words_to_be<12>(out, w1);
    words_to_be<12>(out + 48, w0);
}
KYB_HD_NOINLINE void g1_encode_unc(uint8_t* out, const g1_aff& a) {
    uint32_t wx[12], wy[12];
    fp_to_words<FC>(wx, a.x);
    fp_to_words<FC>(wy, a.y);
    if (a.inf) {
#pragma unroll
        for (int k = 0; k < 12; k++) wx[k] = wy[k] = 0;
        wx[11] = 0x40000000u;
    }
    words_to_be<12>(out, wx);
    words_to_be<12>(out + 48, wy);
}
KYB_HD_NOINLINE void g2_encode_unc(uint8_t* out, const g2_aff& a) {
    uint32_t w[4][12];
    fp_to_words<FC>(w[0], a.x.c1);
    fp_to_words<FC>(w[1], a.x.c0);
    fp_to_words<FC>(w[2], a.y.c1);
    fp_to_words<FC>(w[3], a.y.c0);
    if (a.inf) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 12; k++) w[j][k] = 0;
        w[0][11] = 0x40000000u;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) words_to_be<12>(out + 48 * j, w[j]);
}
KYB_HD size_t g1_out_size(uint32_t flags) { return (flags & FLAG_UNCOMPRESSED_OUT) ? G1_WIRE_UNC : 48; }
KYB_HD size_t g2_out_size(uint32_t flags) { return (flags & FLAG_UNCOMPRESSED_OUT) ? G2_WIRE_UNC : 96; }
KYB_HD void g1_encode_f(uint8_t* out, const g1_aff& a, uint32_t flags) {
    if (flags & FLAG_UNCOMPRESSED_OUT) g1_encode_unc(out, a);
    else g1_encode(out, a);
}
KYB_HD void g2_encode_f(uint8_t* out, const g2_aff& a, uint32_t flags) {
    if (flags & FLAG_UNCOMPRESSED_OUT) g2_encode_unc(out, a);
    else g2_encode(out, a);
}
// 576 bytes: Fp12.c1 then c0; within Fp6 c2, c1, c0; within Fp2 c1, c0; big-endian (oracle gt_to_bytes)
KYB_HD_NOINLINE void gt_encode(uint8_t* out, const fp12& f) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const fp6& s = h == 0 ? f.c1 : f.c0;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const fp2& c = m == 0 ? s.c2 : (m == 1 ? s.c1 : s.c0);
            uint32_t w[12];
            fp_to_words<FC>(w, c.c1);
            words_to_be<12>(out + (h * 3 + m) * 96, w);
            fp_to_words<FC>(w, c.c0);
            words_to_be<12>(out + (h * 3 + m) * 96 + 48, w);
        }
    }
}
// The range half of GT.FromBytes of the kilic backend (kilic/gt.go:100-104): 576 bytes, every coefficient < p; the
// membership of the order-r subgroup is decided by the tower machine's GTMUL program (bls12381_pair.hip).  Layout as
// gt_encode.
KYB_HD_NOINLINE int gt_decode(fp12& f, const uint8_t* in) {
    bool ok = true;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        fp6& s = h == 0 ? f.c1 : f.c0;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            fp2& c = m == 0 ? s.c2 : (m == 1 ? s.c1 : s.c0);
            uint32_t w[12];
            words_from_be<12>(w, in + (h * 3 + m) * 96);
            ok = ok & fp_words_lt_p<FC>(w);
            fp_from_words<FC>(c.c1, w);
            words_from_be<12>(w, in + (h * 3 + m) * 96 + 48);
            ok = ok & fp_words_lt_p<FC>(w);
            fp_from_words<FC>(c.c0, w);
        }
    }
    return ok ? ST_OK : ST_BAD_POINT;
}
// 32-byte big-endian scalar (mod.Int wire format, group/mod/int.go:334-350) -> little-endian words
KYB_HD void scalar_from_be(uint32_t (&k)[8], const uint8_t* in) { words_from_be<8>(k, in); }

// ------------------------------------------------- endomorphism-accelerated scalar multiplication
// Points that passed (or are vouched for by KYB_F_TRUSTED) the subgroup check satisfy phi(P) = [-z^2] P on G1 and
// psi(Q) = [z] Q on G2 (z = -X_ABS) -- the relations the checks themselves test.  Splitting the scalar in base z^2
// (G1: two ~128-bit halves, GLV) or base |z| (G2: four ~64-bit quarters, GLS) divides the doublings by 2 / 4; the
// quotients are plain non-negative integers obtained by long division, so there is no lattice rounding to get wrong,
// and scalars >= r need no special case (k P = sum a_i |z|^i P holds over the integers).
//
// floor(k / d) and k mod d for the two divisors of the endomorphism splits -- DW = 4: d = z^2 (G1, GLV), DW = 2:
// d = |z| (G2, GLS) -- by the divisor's reciprocal m = floor(2^(256 + 32 DW) / d) = 2^256 + MR (Barrett): with
// k < 2^256 the estimate floor(k m / 2^(256 + 32 DW)) is the quotient or one below it, so one conditional correction
// is exact (two are executed).  ~250 instructions; the bitwise restoring division this replaces took ~25 per bit --
// 6 400 per scalar, more than a point addition of the MSM whose decode kernel splits 2^20 scalars.
template <int DW>
struct ZDiv;
template <>
struct ZDiv<4> {  // z^2 = 0xac45a4010001a402 00000001 00000000
    static constexpr uint32_t D[4] = {0x00000000u, 0x00000001u, 0x0001a402u, 0xac45a401u};
    static constexpr uint32_t MR[8] = {0x818be409u, 0xa1a872d6u, 0x27adc027u, 0x034eb4b9u, 0xf6cfee2eu, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u};
};
template <>
struct ZDiv<2> {  // |z| = 0xd201000000010000
    static constexpr uint32_t D[2] = {0x00010000u, 0xd2010000u};
    static constexpr uint32_t MR[8] = {0x2942e444u, 0xf77cf78au, 0x8573b29cu, 0x92078a5eu, 0x3e76ec28u, 0x33cfcc0du, 0x56cd56b5u, 0x381204cau};
};
template <int DW>
KYB_HD void divmod_z(uint32_t (&q)[8], uint32_t (&rem)[DW], const uint32_t (&k)[8]) {
    uint32_t d[DW], mr[8];
#pragma unroll
    for (int i = 0; i < DW; i++) d[i] = ZDiv<DW>::D[i];
#pragma unroll
    for (int i = 0; i < 8; i++) mr[i] = ZDiv<DW>::MR[i];
    uint32_t acc[17];  // k * m = k * MR + k * 2^256
#pragma unroll
    for (int i = 0; i < 17; i++) acc[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint64_t t = (uint64_t)k[i] * mr[j] + acc[i + j] + carry;  // < 2^64
            acc[i + j] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
        acc[i + 8] = carry;
    }
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc[8 + i] = adc32(acc[8 + i], k[i], c);
    acc[16] = c;
    constexpr int QW = 9 - DW;  // words of the quotient: k / d < 2^(257 - 32 DW)
    uint32_t qh[QW];
#pragma unroll
    for (int i = 0; i < QW; i++) qh[i] = acc[8 + DW + i];
    // r = k - qh * d, known to lie in [0, 2 d): DW + 1 words suffice
    uint32_t pr[DW + 1];
#pragma unroll
    for (int i = 0; i <= DW; i++) pr[i] = 0;
#pragma unroll
    for (int i = 0; i <= DW && i < QW; i++) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < DW && i + j <= DW; j++) {
            const uint64_t t = (uint64_t)qh[i] * d[j] + pr[i + j] + carry;
            pr[i + j] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
        if (i + DW <= DW) pr[i + DW] += carry;  // only i = 0 reaches the top word this way
    }
    uint32_t r[DW + 1];
    uint32_t b = 0;
#pragma unroll
    for (int i = 0; i <= DW; i++) r[i] = sbb32(k[i], pr[i], b);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        uint32_t t[DW + 1];
        b = 0;
#pragma unroll
        for (int i = 0; i <= DW; i++) t[i] = sbb32(r[i], i < DW ? d[i] : 0u, b);
        const uint32_t ge = b - 1u;  // all ones when r >= d
#pragma unroll
        for (int i = 0; i <= DW; i++) r[i] = sel32(ge, t[i], r[i]);
        c = ge & 1u;
#pragma unroll
        for (int i = 0; i < QW; i++) {
            const uint32_t v = qh[i] + c;
            c = v < c ? 1u : 0u;
            qh[i] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = i < QW ? qh[i] : 0u;
#pragma unroll
    for (int i = 0; i < DW; i++) rem[i] = r[i];
}
KYB_HD void glv_digits(int8_t (&e)[65], const uint32_t* w, int nwords) {
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = i < nwords ? w[i] : 0u;
    recode16_u256(e, k);
}
// One GLV window on G1 in one out-of-line call: acc = 16 acc (when dbl) + d0 P + d1 z^2 P, z^2 P = (beta x, -y).
KYB_HD_NOINLINE void g1_glv_step(g1_jac& acc, const g1_jac (&tab)[8], int d0, int d1, const fp& beta, bool dbl) {
    g1_jac x = acc, t, s;
    if (dbl) {  // uniform across the grid
#pragma unroll 1
        for (int i = 0; i < 4; i++) jac_dbl_inl(x, x);
    }
    jac_select8(t, tab, d0);
    jac_madd_inl(s, x, t.X, t.Y, jac_is_inf(t));  // the table is affine (jac_table8_to_affine)
    jac_cmov(x, s, d0 != 0);
    jac_select8(t, tab, -d1);  // -y
    fp_mul(t.X, t.X, beta);
    jac_madd_inl(s, x, t.X, t.Y, jac_is_inf(t));
    jac_cmov(x, s, d1 != 0);
    acc = x;
}
// r = k * P for P in G1: k = k1 z^2 + k0, z^2 P = -phi(P) = (beta x, -y); 34 windows of (4 doublings + 2 additions).
KYB_HD_NOINLINE void g1_mul_glv(g1_jac& r, const g1_jac& p, const uint32_t (&k)[8]) {
    uint32_t q[8], rem[4];
    divmod_z<4>(q, rem, k);
    int8_t e0[65], e1[65];
    glv_digits(e0, rem, 4);
    glv_digits(e1, q, 5);  // k < 2^256 and z^2 > 2^127: the quotient has at most 129 bits
    g1_jac tab[8];  // (j + 1) * P
    tab[0] = p;
    jac_dbl(tab[1], p);
#pragma unroll 1
    for (int j = 2; j < 8; j++) jac_add(tab[j], tab[j - 1], p);
    jac_table8_to_affine(tab);
    fp beta;
    fp_const(beta, CC::BETA);
    g1_jac acc;
    jac_set_inf(acc);
#pragma unroll 1
    for (int i = 33; i >= 0; i--) g1_glv_step(acc, tab, e0[i], e1[i], beta, i != 33);
    r = acc;
}
// The same walk on lazy limbs (jac_lazy.cuh) -- the field's own thirteen 30-bit limbs, R / p = 630: the formulas written for
// that headroom (jaclz_dbl_t 3M + 4S, jaclz_madd_t 8M + 3S; a first version on fourteen limbs, R' / p = 2^39, paid 392
// multiply-adds per product for the generic formulas).  The window table is built by the packed code, its entries and their
// beta x images are unpacked once, and the 136 doublings + up to 68 mixed additions run inlined with no reduction
// between products and nothing but the table and the digits in private memory.  (Inlined into its caller for the reason given at bn_suite.inc g1_mul_glv_lz.)
// TAB (round 6): the 16 table entries the walk reads by the lane's own digits -- (j + 1) P and its image (beta x, -y) --
// live in a lane-contiguous slab of global memory (`tab`: this lane's G1_TAB_WORDS words; an entry is seven 16-byte loads)
// instead of private scratch, whose per-dword interleaving over the wave turns an indexed entry read into up to eight
// rows per dword: 2.1 GB per 2^16 multiplications against 8 MB of operands (252 x; "bound by its scratch
// traffic", bls12381_g1split.hip).  The BN G2 ladders did the same (bn_suite.inc g2_mul_gls_lz<TAB>).  TAB = false: the
// arrays (host harness; -DKYB_BLS_G1_TAB_SCRATCH for A/B builds).
constexpr size_t G1_TAB_ENTRY_WORDS = 28;                       // x (13 limbs), y (13 limbs), 2 words of padding: 7 x 16 bytes
constexpr size_t G1_TAB_WORDS = 2 * 8 * G1_TAB_ENTRY_WORDS;     // per lane: the two images of eight multiples
static_assert(2 * FC::N <= G1_TAB_ENTRY_WORDS, "entry layout");
KYB_HD void g1_tab_put(uint32_t* __restrict__ tab, int h, int i, const LzFpN<FC>::E& x, const LzFpN<FC>::E& y) {
    uint32_t w[G1_TAB_ENTRY_WORDS];
#pragma unroll
    for (int l = 0; l < FC::N; l++) {
        w[l] = x.l[l];
        w[FC::N + l] = y.l[l];
    }
    w[26] = w[27] = 0;
    uint32_t* q = tab + (size_t)(h * 8 + i) * G1_TAB_ENTRY_WORDS;
#if defined(__HIPCC__)
#pragma unroll
    for (int v = 0; v < (int)G1_TAB_ENTRY_WORDS / 4; v++)
        reinterpret_cast<uint4*>(q)[v] = make_uint4(w[4 * v], w[4 * v + 1], w[4 * v + 2], w[4 * v + 3]);
#else
    for (int v = 0; v < (int)G1_TAB_ENTRY_WORDS; v++) q[v] = w[v];
#endif
}
KYB_HD void g1_tab_get(LzFpN<FC>::E& x, LzFpN<FC>::E& y, const uint32_t* __restrict__ tab, int h, int i) {
    uint32_t w[G1_TAB_ENTRY_WORDS];
    const uint32_t* q = tab + (size_t)(h * 8 + i) * G1_TAB_ENTRY_WORDS;
#if defined(__HIPCC__)
#pragma unroll
    for (int v = 0; v < (int)G1_TAB_ENTRY_WORDS / 4; v++) {
        const uint4 t = reinterpret_cast<const uint4*>(q)[v];
        w[4 * v] = t.x;
        w[4 * v + 1] = t.y;
        w[4 * v + 2] = t.z;
        w[4 * v + 3] = t.w;
    }
#else
    for (int v = 0; v < (int)G1_TAB_ENTRY_WORDS; v++) w[v] = q[v];
#endif
#pragma unroll
    for (int l = 0; l < FC::N; l++) {
        x.l[l] = w[l];
        y.l[l] = w[FC::N + l];
    }
    KYB_LZ_K(x.k = y.k = 2.0;)
}
template <bool TAB>
KYB_HD void g1_mul_glv_lz(g1_jac& r, const g1_jac& p, const uint32_t (&k)[8], uint32_t* __restrict__ tab = nullptr) {
    using LF = LzFpN<FC>;
    uint32_t q[8], rem[4];
    divmod_z<4>(q, rem, k);
    int8_t e0[65], e1[65];
    glv_digits(e0, rem, 4);
    glv_digits(e1, q, 5);
    if (jac_is_inf(p)) {
        jac_set_inf(r);
        return;
    }
    constexpr int NT = TAB ? 1 : 2;
    typename LF::E tx[NT][8], ty[8], beta;  // (j + 1) P and its image (beta x, -y): affine, in limb form
    uint32_t infmask;
    jaclz_table8<LF, true>(tx[0], ty, infmask, p.X, p.Y);  // p comes from jac_from_aff: Z = 1
    {
        fp b;
        fp_const(b, CC::BETA);
        LF::enter(beta, b);
    }
#pragma unroll 1
    for (int j = 0; j < 8; j++) {
        typename LF::E bx;
        LF::mul(bx, tx[0][j], beta);
        if constexpr (TAB) {
            g1_tab_put(tab, 0, j, tx[0][j], ty[j]);
            g1_tab_put(tab, 1, j, bx, ty[j]);
        } else {
            tx[NT - 1][j] = bx;
        }
    }
    JacLz<LF> acc;
    jaclz_set_inf(acc);
#pragma unroll 1
    for (int i = 33; i >= 0; i--) {
        if (i != 33) {
#pragma unroll 1
            for (int d = 0; d < 4; d++) jaclz_dbl_t(acc);
        }
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const int d = h == 0 ? e0[i] : e1[i];
            if (d == 0) continue;
            const int idx = (d < 0 ? -d : d) - 1;
            if ((infmask >> idx) & 1u) continue;
            if constexpr (TAB) {
                typename LF::E ex, ey;
                g1_tab_get(ex, ey, tab, h, idx);
                jaclz_madd_t(acc, ex, ey, (d < 0) != (h == 1));  // z^2 P = (beta x, -y)
            } else {
                jaclz_madd_t(acc, tx[h < NT ? h : 0][idx], ty[idx], (d < 0) != (h == 1));
            }
        }
    }
    jaclz_leave(r, acc);
}
// r = k * Q for Q in G2: k = a0 + a1 |z| + a2 |z|^2 + a3 |z|^3 and |z| Q = -psi(Q), so
// k Q = a0 Q - a1 psi(Q) + a2 psi^2(Q) - a3 psi^3(Q); 18 windows of (4 doublings + 4 additions).
// psi(x, y) = (cx conj x, cy conj y) on the affine table entries; psi^2 scales x, y by the norms of cx, cy.
KYB_HD_NOINLINE void g2_mul_gls(g2_jac& r, const g2_jac& p, const uint32_t (&k)[8]) {
    uint32_t q1[8], q2[8], q3[8], a0[2], a1[2], a2[2];
    divmod_z<2>(q1, a0, k);
    divmod_z<2>(q2, a1, q1);
    divmod_z<2>(q3, a2, q2);  // q3 = a3 < 2^66
    int8_t e[4][65];
    glv_digits(e[0], a0, 2);
    glv_digits(e[1], a1, 2);
    glv_digits(e[2], a2, 2);
    glv_digits(e[3], q3, 3);
    g2_jac tab[8];
    tab[0] = p;
    jac_dbl(tab[1], p);
#pragma unroll 1
    for (int j = 2; j < 8; j++) jac_add(tab[j], tab[j - 1], p);
    jac_table8_to_affine(tab);  // psi^j acts on the affine entry; the 72 table additions are mixed additions
    fp2 cx, cy, cx3, cy3, t2;
    fp nx, ny, u;
    fp2_load_const<TC>(cx, CC::PSI_CX);
    fp2_load_const<TC>(cy, CC::PSI_CY);
    fp_sqr(nx, cx.c0);
    fp_sqr(u, cx.c1);
    fp_add(nx, nx, u);  // cx conj(cx)
    fp_sqr(ny, cy.c0);
    fp_sqr(u, cy.c1);
    fp_add(ny, ny, u);
    fp2_mul_fp(cx3, cx, nx);
    fp2_mul_fp(cy3, cy, ny);
    g2_jac acc, t, s;
    jac_set_inf(acc);
#pragma unroll 1
    for (int i = 17; i >= 0; i--) {
        if (i != 17) jac_dbl_n(acc, acc, 4);
        // a0 Q
        jac_select8(t, tab, e[0][i]);
        jac_madd(s, acc, t.X, t.Y, jac_is_inf(t));
        jac_cmov(acc, s, e[0][i] != 0);
        // -a1 psi(Q)
        jac_select8(t, tab, -e[1][i]);
        fp2_conj(t2, t.X);
        fp2_mul_c(t.X, t2, cx);
        fp2_conj(t2, t.Y);
        fp2_mul_c(t.Y, t2, cy);
        jac_madd(s, acc, t.X, t.Y, jac_is_inf(t));
        jac_cmov(acc, s, e[1][i] != 0);
        // a2 psi^2(Q)
        jac_select8(t, tab, e[2][i]);
        fp2_mul_fp(t.X, t.X, nx);
        fp2_mul_fp(t.Y, t.Y, ny);
        jac_madd(s, acc, t.X, t.Y, jac_is_inf(t));
        jac_cmov(acc, s, e[2][i] != 0);
        // -a3 psi^3(Q)
        jac_select8(t, tab, -e[3][i]);
        fp2_conj(t2, t.X);
        fp2_mul_c(t.X, t2, cx3);
        fp2_conj(t2, t.Y);
        fp2_mul_c(t.Y, t2, cy3);
        jac_madd(s, acc, t.X, t.Y, jac_is_inf(t));
        jac_cmov(acc, s, e[3][i] != 0);
    }
    r = acc;
}

// ------------------------------------------------- per-element wire-level operations
// (what one lane of a batch kernel does; tests/host_harness.cpp runs the same functions on the CPU)
KYB_HD void zero_bytes(uint8_t* out, int n) {
    uint32_t* q = reinterpret_cast<uint32_t*>(out);
    for (int k = 0; k < n / 4; k++) q[k] = 0;
}
// out = k * P   (G1Elt.UnmarshalBinary + Mul + MarshalBinary, kilic/g1.go:110-131)
// `tab`: this lane's table slab (G1_TAB_WORDS words of global memory; device code always gets one) or nullptr on the host
KYB_HD int g1_mul_wire(uint8_t* out, const uint8_t* scalar_be, const uint8_t* pt, uint32_t flags = 0, uint32_t* tab = nullptr) {
    g1_aff a;
    const int st = g1_decode_f(a, pt, flags, 0);
    if (st != ST_OK) {
        zero_bytes(out, (int)g1_out_size(flags));
        return st;
    }
    uint32_t k[8];
    scalar_from_be(k, scalar_be);
    g1_jac p, r;
    jac_from_aff(p, a);
#ifdef KYB_BLS_PACKED_LADDER
    g1_mul_glv(r, p, k);
#else
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KYB_BLS_G1_TAB_SCRATCH)
    g1_mul_glv_lz<true>(r, p, k, tab);
#else
    (void)tab;
    g1_mul_glv_lz<false>(r, p, k);
#endif
#endif
    jac_to_aff(a, r);
    g1_encode_f(out, a, flags);
    return ST_OK;
}
constexpr size_t G2_TAB_WORDS = 0;  // (pairing_abi.cuh: no per-lane table slab in global memory for this suite's G2 kernel)
KYB_HD int g2_mul_wire(uint8_t* out, const uint8_t* scalar_be, const uint8_t* pt, uint32_t flags = 0, uint32_t* = nullptr) {
    g2_aff a;
    const int st = g2_decode_f(a, pt, flags, 0);
    if (st != ST_OK) {
        zero_bytes(out, (int)g2_out_size(flags));
        return st;
    }
    uint32_t k[8];
    scalar_from_be(k, scalar_be);
    g2_jac p, r;
    jac_from_aff(p, a);
    g2_mul_gls(r, p, k);
    jac_to_aff(a, r);
    g2_encode_f(out, a, flags);
    return ST_OK;
}
// out = Marshal(Unmarshal(in)): the validation step of G1Elt/G2Elt.UnmarshalBinary (kilic/g1.go:127-131, g2.go) as a
// batch operation -- ZCash flag rules, on-curve, r-torsion -- followed by the re-encoding the flags ask for, so that
// a caller validates once and afterwards passes KYB_F_UNCOMPRESSED | KYB_F_TRUSTED(i).
KYB_HD int g1_unmarshal_wire(uint8_t* out, const uint8_t* pt, uint32_t flags = 0) {
    g1_aff a;
    const int st = g1_decode_f(a, pt, flags, 0);
    if (st != ST_OK) {
        zero_bytes(out, (int)g1_out_size(flags));
        return st;
    }
    g1_encode_f(out, a, flags);
    return ST_OK;
}
KYB_HD int g2_unmarshal_wire(uint8_t* out, const uint8_t* pt, uint32_t flags = 0) {
    g2_aff a;
    const int st = g2_decode_f(a, pt, flags, 0);
    if (st != ST_OK) {
        zero_bytes(out, (int)g2_out_size(flags));
        return st;
    }
    g2_encode_f(out, a, flags);
    return ST_OK;
}
// out = a + b   (Point.Add: kilic/g1.go:90-96, pairing/bn256/point.go:130-140 -> curve.go:69)
KYB_HD int g1_add_wire(uint8_t* out, const uint8_t* pa, const uint8_t* pb) {
    g1_aff a, b;
    int st = g1_decode(a, pa, true);
    const int st2 = g1_decode(b, pb, true);
    if (st == ST_OK) st = st2;
    if (st != ST_OK) {
        zero_bytes(out, 48);
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
    int st = g2_decode(a, pa, true);
    const int st2 = g2_decode(b, pb, true);
    if (st == ST_OK) st = st2;
    if (st != ST_OK) {
        zero_bytes(out, 96);
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
}  // namespace bls
}  // namespace kyb
