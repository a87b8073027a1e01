// BLS12-381: the fixed-base table's policies (fixed_base.cuh) -- how a 256-bit scalar is cut into sub-scalars over the
// endomorphism images of a table entry, what UnmarshalBinary checks before the table exists, and how the finished
// table answers the one rule left (subgroup membership: kilic/g1.go:127-131, g2.go -> InCorrectSubgroup).
#pragma once
#include "bls12381.cuh"
#include "fixed_base.cuh"

namespace kyb {
namespace bls {

// G1: k = q z^2 + rem (divmod_z<4>: rem < z^2 < 2^128, q < 2^129), images (x, y) and z^2 P = (beta x, -y) -- the relation
// g1_in_subgroup tests, i.e. what UnmarshalBinary guarantees.  13 windows of 10 bits hold 130 bits.
struct fb_g1_policy {
    static constexpr int NI = 2, NW = 13;
    KYB_HD static void split(uint32_t (&sub)[2][8], const uint32_t (&k)[8]) {
        uint32_t q[8], rem[4];
        divmod_z<4>(q, rem, k);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            sub[0][i] = i < 4 ? rem[i] : 0u;
            sub[1][i] = i < 5 ? q[i] : 0u;
        }
    }
    KYB_HD static void images(fb::Entry<fp, 2>& e, const fp& x, const fp& y) {
        fp beta;
        fp_const(beta, CC::BETA);
        e.x[0] = x;
        e.y[0] = y;
        fp_mul(e.x[1], x, beta);
        fp_neg(e.y[1], y);
    }
    // every rule of UnmarshalBinary but the subgroup's (flag bits, range, the curve equation -- a vouched-for
    // uncompressed point skips the equation too, as everywhere)
    KYB_HD static int decode_on_curve(g1_aff& a, const uint8_t* in, uint32_t flags) {
        const bool trusted = flag_trusted(flags, 0);
        return (flags & FLAG_UNCOMPRESSED) ? g1_decode_unc(a, in, !trusted, false) : g1_decode(a, in, false);
    }
    // g1_decode(a, in, false) in two halves around its one 379-bit power, which fixed_base.cuh chain_rows_kernel runs on a
    // row (rowfp::pow_words: 0.25 ms instead of a lane's 0.5) -- restated rather than shared, so that the batch kernels'
    // copy keeps its shape.  decode_head: flag bits, range, x and x^3 + b; go = there is a root to take.
    static constexpr int ROW_SQRT = 1;
    KYB_HD static int decode_head(fp& x, fp& rhs, int& sflag, int& inf, const uint8_t* in) {
        uint32_t w[12];
        words_from_be<12>(w, in);
        const uint32_t top = w[11] >> 29;
        const bool c = top & 4, i = top & 2, s = top & 1;
        w[11] &= 0x1fffffffu;
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) any |= w[k];
        inf = 1;
        sflag = s ? 1 : 0;
        if (!c) return ST_BAD_POINT;
        if (i) return (s || any) ? ST_BAD_POINT : ST_OK;
        if (!fp_words_lt_p<FC>(w)) return ST_BAD_POINT;
        fp b;
        fp_from_words<FC>(x, w);
        fp_const(b, CC::B1);
        fp_sqr(rhs, x);
        fp_mul(rhs, rhs, x);
        fp_add(rhs, rhs, b);
        inf = 0;
        return ST_OK;
    }
    // y0 = rhs^((p + 1) / 4), fully reduced: a root iff its square is rhs; the sort flag picks the sign
    KYB_HD static int decode_tail(g1_aff& a, const fp& x, const fp& rhs, const fp& y0, int sflag) {
        fp t, y = y0;
        fp_zero(a.x);
        fp_zero(a.y);
        a.inf = true;
        fp_sqr(t, y);
        if (!fp_eq(t, rhs)) return ST_BAD_POINT;
        fp_neg(t, y);
        fp_cmov(y, t, fp_is_larger(y) != (sflag != 0));
        a.x = x;
        a.y = y;
        a.inf = false;
        return ST_OK;
    }
    KYB_HD static bool needs_member(uint32_t flags) { return !flag_trusted(flags, 0); }
    // Scott's criterion (g1_in_subgroup): z^2 P = -phi(P) = (beta x, -y), with z^2 P walked over the table's plain image
    KYB_HD static bool member(const g1_aff& a, const fb::Entry<fp, 2>* __restrict__ tab) {
        const uint32_t z2[8] = {ZDiv<4>::D[0], ZDiv<4>::D[1], ZDiv<4>::D[2], ZDiv<4>::D[3], 0u, 0u, 0u, 0u};
        g1_jac q;
        fb::mul_plain<2>(q, z2, NW, tab);
        fp beta, bx, ny, zz, zzz, l, r;
        fp_const(beta, CC::BETA);
        fp_mul(bx, a.x, beta);
        fp_neg(ny, a.y);
        fp_sqr(zz, q.Z);
        fp_mul(zzz, zz, q.Z);
        fp_mul(l, bx, zz);
        fp_mul(r, ny, zzz);
        return fp_eq(l, q.X) & fp_eq(r, q.Y) & !fp_is_zero(q.Z);
    }
};

// G2: k = a0 + a1 |z| + a2 |z|^2 + a3 |z|^3 (three long divisions by |z| < 2^64; a3 < 2^65), images
// |z|^j Q = (-1)^j psi^j(Q) (z is negative: psi(Q) = [z] Q on G2 -- the relation g2_in_subgroup tests).  7 windows hold 70 bits.
struct fb_g2_policy {
    static constexpr int NI = 4, NW = 7;
    KYB_HD static void split(uint32_t (&sub)[4][8], const uint32_t (&k)[8]) {
        uint32_t q1[8], q2[8], q3[8], a0[2], a1[2], a2[2];
        divmod_z<2>(q1, a0, k);
        divmod_z<2>(q2, a1, q1);
        divmod_z<2>(q3, a2, q2);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            sub[0][i] = i < 2 ? a0[i] : 0u;
            sub[1][i] = i < 2 ? a1[i] : 0u;
            sub[2][i] = i < 2 ? a2[i] : 0u;
            sub[3][i] = i < 3 ? q3[i] : 0u;
        }
    }
    KYB_HD static void psi(fp2& x, fp2& y) {  // in place: (conj(x) cx, conj(y) cy)
        fp2 cx, cy, t;
        fp2_load_const<TC>(cx, CC::PSI_CX);
        fp2_load_const<TC>(cy, CC::PSI_CY);
        fp2_conj(t, x);
        fp2_mul_c(x, t, cx);
        fp2_conj(t, y);
        fp2_mul_c(y, t, cy);
    }
    KYB_HD static void images(fb::Entry<fp2, 4>& e, const fp2& x, const fp2& y) {
        fp2 px = x, py = y;
        e.x[0] = x;
        e.y[0] = y;
#pragma unroll 1
        for (int j = 1; j < 4; j++) {
            psi(px, py);
            e.x[j] = px;
            e.y[j] = py;
            if (j & 1) fp2_neg(e.y[j], py);  // |z|^j Q = (-1)^j psi^j(Q)
        }
    }
    KYB_HD static int decode_on_curve(g2_aff& a, const uint8_t* in, uint32_t flags) {
        const bool trusted = flag_trusted(flags, 0);
        return (flags & FLAG_UNCOMPRESSED) ? g2_decode_unc(a, in, !trusted, false) : g2_decode(a, in, false);
    }
    KYB_HD static bool needs_member(uint32_t flags) { return !flag_trusted(flags, 0); }
    // psi(Q) = [z] Q (g2_in_subgroup): |z| Q = -psi(Q), with |z| Q walked over the plain image
    KYB_HD static bool member(const g2_aff& a, const fb::Entry<fp2, 4>* __restrict__ tab) {
        const uint32_t zabs[8] = {(uint32_t)CC::X_ABS, (uint32_t)(CC::X_ABS >> 32), 0u, 0u, 0u, 0u, 0u, 0u};
        g2_jac q;
        fb::mul_plain<4>(q, zabs, NW, tab);
        fp2 px = a.x, py = a.y, zz, zzz, l, r;
        psi(px, py);
        fp2_neg(py, py);
        fp2_sqr_c(zz, q.Z);
        fp2_mul_c(zzz, zz, q.Z);
        fp2_mul_c(l, px, zz);
        fp2_mul_c(r, py, zzz);
        return fp2_eq(l, q.X) & fp2_eq(r, q.Y) & !fp2_is_zero(q.Z);
    }
};

}  // namespace bls
}  // namespace kyb
