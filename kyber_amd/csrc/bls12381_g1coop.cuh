// BLS12-381 G1Elt.Mul (kilic/g1.go:110-116) for SMALL batches: four cooperating lanes per point.
//
// A batch below ~2^14 elements leaves most SIMDs of the chip empty or with one wave, so a multiplication costs the
// LATENCY of its ladder whatever the batch: 3.4 ms per call for n = 1 .. 4096 with one element per lane (a window of the
// GLV ladder is 4 doublings + 2 additions = ~50 dependent field multiplications).  Here a point is owned by four lanes
// that split the independent products of every formula through LDS slots (coop_slots.cuh: a doubling is 3 product
// levels deep, an addition 5) -- 24 levels per window instead of 50 multiplications; the window table (P .. 8P, with
// beta x beside every x for the z^2 half) sits in LDS too: nothing in scratch, nothing in HBM but the wire bytes.
// The arithmetic is 1.6 x the per-lane ladder's (full additions, idle lanes on the short levels), which is why this is
// the small-batch path only (DESIGN.md section 5 item 36 prices the large-batch case).
//
// Same algorithm and digits as g1_mul_glv (bls12381.cuh): k = q z^2 + rem, signed radix-16 digits of both halves, table
// (j + 1) P, z^2 P = (beta x, -y).  Exceptional additions (P = +-Q, operands at infinity) are coop::add's own.
#pragma once
#include "bls12381.cuh"
#include "coop_slots.cuh"

namespace kyb {
namespace bls {
namespace g1coop {

constexpr int NDIG = 34;  // windows: 132 bits of each half + the recoding's carry
// slot map of one point's group
constexpr int ACC = 0, Q = 3, TMP = 6, TAB = TMP + coop::TEMPS, TABX = TAB + 24, NS = TABX + 8;
using Slot = coop::Slot<fp>;

// lane 0's part before the ladder: the two halves' digits (e[0 .. NDIG) each) from the scalar
KYB_HD void digits(int8_t (&e0)[NDIG], int8_t (&e1)[NDIG], const uint32_t (&k)[8]) {
    uint32_t q[8], rem[4];
    divmod_z<4>(q, rem, k);
    int8_t d0[65], d1[65];
    glv_digits(d0, rem, 4);
    glv_digits(d1, q, 5);
#pragma unroll 1
    for (int i = 0; i < NDIG; i++) {
        e0[i] = d0[i];
        e1[i] = d1[i];
    }
}

// S[ACC ..] <- k P for the point whose affine (x, y) lane 0 has put into S[TAB], S[TAB + 1] (S[TAB + 2] = 1); all four
// lanes of the group (r = 0 .. 3) call this, every group of the workgroup in step (the routines contain barriers).
// e0 / e1: the group's digits (shared memory).
KYB_COOP_FN void ladder(Slot* S, uint32_t* fl, int r, const int8_t* e0, const int8_t* e1) {
    // ---- table: T[j] = (j + 1) P, j = 0 .. 7 (Jacobian)
    if (r < 3) S[ACC + r].f = S[TAB + r].f;
    KYB_COOP_SYNC();
    coop::dbl<fp>(S, r, ACC, TMP, true);
    if (r < 3) {
        S[TAB + 3 + r].f = S[ACC + r].f;
        S[Q + r].f = S[TAB + r].f;
    }
    KYB_COOP_SYNC();
#pragma unroll 1
    for (int j = 2; j < 8; j++) {
        coop::add<fp>(S, fl, r, ACC, Q, TMP, true);
        if (r < 3) S[TAB + 3 * j + r].f = S[ACC + r].f;
        KYB_COOP_SYNC();
    }
    {  // beta x_j for the z^2 half: two products per lane
        fp beta;
        fp_const(beta, CC::BETA);
#pragma unroll 1
        for (int j = r; j < 8; j += 4) {
            fp t = S[TAB + 3 * j].f;
            fp_mul(t, t, beta);
            S[TABX + j].f = t;
        }
    }
    if (r < 3) {
        fp v;
        fp_one(v);
        if (r == 2) fp_zero(v);
        S[ACC + r].f = v;  // the accumulator starts at infinity
    }
    KYB_COOP_SYNC();
    // ---- 34 windows of 4 doublings + 2 additions
#pragma unroll 1
    for (int i = NDIG - 1; i >= 0; i--) {
        if (i != NDIG - 1) {
#pragma unroll 1
            for (int d = 0; d < 4; d++) coop::dbl<fp>(S, r, ACC, TMP, true);
        }
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const int d = h ? e1[i] : e0[i];
            const int a = d < 0 ? -d : d, j = a ? a - 1 : 0;
            // z^2 P = (beta x, -y): the second half adds the NEGATED y for a positive digit
            const bool neg = h ? d > 0 : d < 0;
            if (r < 3) {
                fp v = S[(r == 0 && h) ? TABX + j : TAB + 3 * j + r].f;
                if (r == 1) {
                    fp nv;
                    fp_neg(nv, v);
                    fp_cmov(v, nv, neg);
                }
                S[Q + r].f = v;
            }
            KYB_COOP_SYNC();
            coop::add<fp>(S, fl, r, ACC, Q, TMP, d != 0);
        }
    }
}

// S[ACC ..] <- [|z|] B for the point in slots B .. B + 2 (any Z): double-and-add over the 63 bits of |z| below its top one
// (five of them set).  Uniform control flow: the parameter is a constant.
KYB_COOP_FN void mul_z(Slot* S, uint32_t* fl, int r, int B) {
    if (r < 3) {
        const fp v = S[B + r].f;
        S[ACC + r].f = v;
        S[Q + r].f = v;
    }
    KYB_COOP_SYNC();
#pragma unroll 1
    for (int bit = 62; bit >= 0; bit--) {
        coop::dbl<fp>(S, r, ACC, TMP, true);
        if ((CC::X_ABS >> bit) & 1ull) coop::add<fp>(S, fl, r, ACC, Q, TMP, true);
    }
}
// The r-torsion rule of UnmarshalBinary (kilic/g1.go:127-131 -> InCorrectSubgroup) for the point lane 0 has put into
// S[TAB .. TAB + 2] (affine, Z = 1), on the group's four lanes: Scott's criterion z^2 P = -phi(P) = (beta x, -y) as in
// g1_in_subgroup, the two multiplications by |z| with cooperative doublings (438 product levels against ~1 040 dependent
// multiplications in one lane).  The verdict is lane 0's; uses S[TAB + 3 .. TAB + 5] (free until the table is built).
KYB_COOP_FN bool member(Slot* S, uint32_t* fl, int r) {
    mul_z(S, fl, r, TAB);
    if (r < 3) S[TAB + 3 + r].f = S[ACC + r].f;
    KYB_COOP_SYNC();
    mul_z(S, fl, r, TAB + 3);
    bool ok = true;
    if (r == 0) {
        fp beta, bx, ny, zz, zzz, l, rr;
        const fp X = S[ACC].f, Y = S[ACC + 1].f, Z = S[ACC + 2].f;
        fp_const(beta, CC::BETA);
        fp_mul(bx, S[TAB].f, beta);
        fp_neg(ny, S[TAB + 1].f);
        fp_sqr(zz, Z);
        fp_mul(zzz, zz, Z);
        fp_mul(l, bx, zz);
        fp_mul(rr, ny, zzz);
        ok = fp_eq(l, X) & fp_eq(rr, Y) & !fp_is_zero(Z);
    }
    KYB_COOP_SYNC();
    return ok;
}

#if defined(__HIPCC__)
constexpr int GROUPS = 16;  // points per workgroup of 64 lanes
// One workgroup = 16 points.  Lane 0 of a group decodes the point, splits and recodes the scalar, and -- after the ladder
// -- turns the accumulator into the wire form; the subgroup test and the ladder run on the group's four lanes.
__global__ __launch_bounds__(64, KYB_TU_WAVES) void bls12381_g1_mul_coop_kernel(size_t n, const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ pts,
                                                                  size_t pt_stride, uint8_t* __restrict__ out, uint8_t* __restrict__ status,
                                                                  uint32_t flags) {
    __shared__ Slot slots[GROUPS * NS];
    __shared__ uint32_t flg[GROUPS * 2];
    __shared__ int8_t dig[GROUPS][2][NDIG];
    __shared__ int verdict[GROUPS];  // UnmarshalBinary's status | 0x100 when the point is at infinity
    const int gi = (int)threadIdx.x >> 2, r = (int)threadIdx.x & 3;
    const size_t idx = (size_t)blockIdx.x * GROUPS + gi;
    const bool live = idx < n;
    const size_t ic = live ? idx : n - 1;  // a dead group walks the last element again: every lane meets every barrier
    Slot* S = slots + gi * NS;
    // every rule of UnmarshalBinary but the subgroup's in lane 0 (flag bits, range, the square root / the curve
    // equation); the subgroup's on the four lanes below, unless the caller vouched for the points
    const bool check = !flag_trusted(flags, 0);
    if (r == 0) {
        g1_aff a;
        const uint8_t* in = pts + pt_stride * ic;
        const int st = (flags & FLAG_UNCOMPRESSED) ? g1_decode_unc(a, in, check, false) : g1_decode(a, in, false);
        verdict[gi] = st | ((st == ST_OK && a.inf) ? 0x100 : 0);
        if (st != ST_OK || a.inf) {  // something harmless for the lanes to chew on
            fp_const(a.x, CC::G1X);
            fp_const(a.y, CC::G1Y);
        }
        uint32_t k[8];
        scalar_from_be(k, scalars + 32 * ic);
        int8_t e0[NDIG], e1[NDIG];
        digits(e0, e1, k);
#pragma unroll 1
        for (int i = 0; i < NDIG; i++) {
            dig[gi][0][i] = e0[i];
            dig[gi][1][i] = e1[i];
        }
        S[TAB].f = a.x;
        S[TAB + 1].f = a.y;
        fp one;
        fp_one(one);
        S[TAB + 2].f = one;
    }
    __syncthreads();
    if (check) {  // uniform: the flags are the call's
        const bool in_g1 = member(S, flg + gi * 2, r);
        if (r == 0 && verdict[gi] == 0 && !in_g1) {
            verdict[gi] = ST_NOT_IN_SUBGROUP;
            fp_const(S[TAB].f, CC::G1X);  // (the ladder walks the generator instead: its lanes meet every barrier)
            fp_const(S[TAB + 1].f, CC::G1Y);
        }
        __syncthreads();
    }
    ladder(S, flg + gi * 2, r, dig[gi][0], dig[gi][1]);
    if (r == 0 && live) {
        uint8_t* o = out + g1_out_size(flags) * idx;
        const int v = verdict[gi];
        if (v & 0xff) {
            zero_bytes(o, (int)g1_out_size(flags));
        } else {
            g1_aff a;
            if (v & 0x100) {
                fp_zero(a.x);
                fp_zero(a.y);
                a.inf = true;
            } else {
                g1_jac p;
                p.X = S[ACC].f;
                p.Y = S[ACC + 1].f;
                p.Z = S[ACC + 2].f;
                jac_to_aff(a, p);
            }
            g1_encode_f(o, a, flags);
        }
        if (status) status[idx] = (uint8_t)(v & 0xff);
    }
}
// G1Elt.UnmarshalBinary for small batches: lane 0 applies the flag / range / curve rules and the square root, the group's
// four lanes the subgroup rule (member()), lane 0 re-encodes -- bls12381_g1_unmarshal_kernel's answer at the latency of
// 438 product levels instead of ~1 040 dependent multiplications.
__global__ __launch_bounds__(64, KYB_TU_WAVES) void bls12381_g1_unmarshal_coop_kernel(size_t n, const uint8_t* __restrict__ pts, uint8_t* __restrict__ out,
                                                                        uint8_t* __restrict__ status, uint32_t flags) {
    constexpr int MS = TAB + 6;  // ACC, Q, temporaries and two points
    __shared__ Slot slots[GROUPS * MS];
    __shared__ uint32_t flg[GROUPS * 2];
    const int gi = (int)threadIdx.x >> 2, r = (int)threadIdx.x & 3;
    const size_t idx = (size_t)blockIdx.x * GROUPS + gi;
    const bool live = idx < n;
    const size_t ic = live ? idx : n - 1;
    Slot* S = slots + gi * MS;
    const bool check = !flag_trusted(flags, 0);
    g1_aff a;  // lane 0's
    int st = ST_OK;
    if (r == 0) {
        const uint8_t* in = pts + g1_wire_size(flags) * ic;
        st = (flags & FLAG_UNCOMPRESSED) ? g1_decode_unc(a, in, check, false) : g1_decode(a, in, false);
        const bool walk = st == ST_OK && !a.inf;
        if (walk) {
            S[TAB].f = a.x;
            S[TAB + 1].f = a.y;
        } else {
            fp_const(S[TAB].f, CC::G1X);
            fp_const(S[TAB + 1].f, CC::G1Y);
        }
        fp_one(S[TAB + 2].f);
    }
    __syncthreads();
    bool in_g1 = true;
    if (check) in_g1 = member(S, flg + gi * 2, r);
    if (r == 0 && live) {
        if (st == ST_OK && !a.inf && !in_g1) st = ST_NOT_IN_SUBGROUP;
        uint8_t* o = out + g1_out_size(flags) * idx;
        if (st != ST_OK) zero_bytes(o, (int)g1_out_size(flags));
        else g1_encode_f(o, a, flags);
        if (status) status[idx] = (uint8_t)st;
    }
}
#endif

}  // namespace g1coop
}  // namespace bls
}  // namespace kyb

// ---------------------------------------------------------------------------------------------------------------- G2
// The same for G2Elt.Mul (kilic/g2.go): a point is four lanes over Fp2 slots, the GLS ladder of g2_mul_gls -- k = a0 + a1 |z|
// + a2 |z|^2 + a3 |z|^3, |z|^j Q = (-1)^j psi^j(Q), 18 windows of 4 doublings + 4 additions -- with psi^j applied to the
// (Jacobian) table entry while it is staged: psi(X : Y : Z) = (cx conj X : cy conj Y : conj Z), one product level in which
// lanes 0 and 1 multiply by the constants (kept once per workgroup in six shared slots).  Small batches only, like G1:
// below ~2^13 elements a call costs the latency of the ladder (4.6 - 6.0 ms validated, 7.4 - 8.7 ms with the checks).
namespace kyb {
namespace bls {
namespace g2coop {

constexpr int NDIG = 18, NH = 4;
constexpr int ACC = 0, Q = 3, TMP = 6, TAB = TMP + coop::TEMPS, NS = TAB + 24;
using Slot = coop::Slot<fp2>;
// shared constants: [0] cx, [1] cy, [2] (nx, 0), [3] (ny, 0), [4] cx nx, [5] cy ny   (psi, psi^2, psi^3 scalings)
constexpr int NCONST = 6;

KYB_HD void constants(Slot* C) {
    fp2 cx, cy, t;
    fp nx, ny, u;
    fp2_load_const<TC>(cx, CC::PSI_CX);
    fp2_load_const<TC>(cy, CC::PSI_CY);
    fp_sqr(nx, cx.c0);
    fp_sqr(u, cx.c1);
    fp_add(nx, nx, u);
    fp_sqr(ny, cy.c0);
    fp_sqr(u, cy.c1);
    fp_add(ny, ny, u);
    C[0].f = cx;
    C[1].f = cy;
    t.c0 = nx;
    fp_zero(t.c1);
    C[2].f = t;
    t.c0 = ny;
    C[3].f = t;
    fp2_mul_fp(t, cx, nx);
    C[4].f = t;
    fp2_mul_fp(t, cy, ny);
    C[5].f = t;
}
KYB_HD void digits(int8_t (&e)[NH][NDIG], const uint32_t (&k)[8]) {
    uint32_t q1[8], q2[8], q3[8], a0[2], a1[2], a2[2];
    divmod_z<2>(q1, a0, k);
    divmod_z<2>(q2, a1, q1);
    divmod_z<2>(q3, a2, q2);
    int8_t d[65];
    const uint32_t* src[NH] = {a0, a1, a2, q3};
#pragma unroll 1
    for (int h = 0; h < NH; h++) {
        glv_digits(d, src[h], h == 3 ? 3 : 2);
#pragma unroll 1
        for (int i = 0; i < NDIG; i++) e[h][i] = d[i];
    }
}
// stage image h (|z|^h Q up to the sign the caller folds into `neg`) of the Jacobian point in slots B .. B + 2 into Q
KYB_COOP_FN void stage(Slot* S, const Slot* C, int r, int B, int h, bool neg) {
    if (r < 3) {
        fp2 v = S[B + r].f;
        if (h & 1) fp2_conj(v, v);
        if (h && r < 2) fp2_mul_c(v, v, C[(h == 1 ? 0 : (h == 2 ? 2 : 4)) + r].f);
        if (r == 1) {
            fp2 nv;
            fp2_neg(nv, v);
            fp2_cmov(v, nv, neg);
        }
        S[Q + r].f = v;
    }
    KYB_COOP_SYNC();
}
// S[ACC ..] <- k Q for the affine point in S[TAB], S[TAB + 1] (S[TAB + 2] = 1); e: the group's digits e[h * NDIG + i]
KYB_COOP_FN void ladder(Slot* S, const Slot* C, uint32_t* fl, int r, const int8_t* e) {
    if (r < 3) S[ACC + r].f = S[TAB + r].f;
    KYB_COOP_SYNC();
    coop::dbl<fp2>(S, r, ACC, TMP, true);
    if (r < 3) {
        S[TAB + 3 + r].f = S[ACC + r].f;
        S[Q + r].f = S[TAB + r].f;
    }
    KYB_COOP_SYNC();
#pragma unroll 1
    for (int j = 2; j < 8; j++) {
        coop::add<fp2>(S, fl, r, ACC, Q, TMP, true);
        if (r < 3) S[TAB + 3 * j + r].f = S[ACC + r].f;
        KYB_COOP_SYNC();
    }
    if (r < 3) {
        fp2 v;
        fp2_one(v);
        if (r == 2) fp2_zero(v);
        S[ACC + r].f = v;
    }
    KYB_COOP_SYNC();
#pragma unroll 1
    for (int i = NDIG - 1; i >= 0; i--) {
        if (i != NDIG - 1) {
#pragma unroll 1
            for (int d = 0; d < 4; d++) coop::dbl<fp2>(S, r, ACC, TMP, true);
        }
#pragma unroll 1
        for (int h = 0; h < NH; h++) {
            const int d = e[h * NDIG + i];
            const int a = d < 0 ? -d : d, j = a ? a - 1 : 0;
            stage(S, C, r, TAB + 3 * j, h, (h & 1) ? d > 0 : d < 0);  // |z|^h Q = (-1)^h psi^h(Q)
            coop::add<fp2>(S, fl, r, ACC, Q, TMP, d != 0);
        }
    }
}
// [|z|] B on the four lanes (B: slots of a Jacobian point), as g1coop::mul_z
KYB_COOP_FN void mul_z(Slot* S, uint32_t* fl, int r, int B) {
    if (r < 3) {
        const fp2 v = S[B + r].f;
        S[ACC + r].f = v;
        S[Q + r].f = v;
    }
    KYB_COOP_SYNC();
#pragma unroll 1
    for (int bit = 62; bit >= 0; bit--) {
        coop::dbl<fp2>(S, r, ACC, TMP, true);
        if ((CC::X_ABS >> bit) & 1ull) coop::add<fp2>(S, fl, r, ACC, Q, TMP, true);
    }
}
// the r-torsion rule (g2_in_subgroup): |z| Q = -psi(Q) for the affine point in S[TAB ..]; lane 0's verdict
KYB_COOP_FN bool member(Slot* S, const Slot* C, uint32_t* fl, int r) {
    mul_z(S, fl, r, TAB);
    bool ok = true;
    if (r == 0) {
        fp2 px, py, zz, zzz, l, rr;
        const fp2 X = S[ACC].f, Y = S[ACC + 1].f, Z = S[ACC + 2].f;
        fp2_conj(px, S[TAB].f);
        fp2_mul_c(px, px, C[0].f);
        fp2_conj(py, S[TAB + 1].f);
        fp2_mul_c(py, py, C[1].f);
        fp2_neg(py, py);
        fp2_sqr_c(zz, Z);
        fp2_mul_c(zzz, zz, Z);
        fp2_mul_c(l, px, zz);
        fp2_mul_c(rr, py, zzz);
        ok = fp2_eq(l, X) & fp2_eq(rr, Y) & !fp2_is_zero(Z);
    }
    KYB_COOP_SYNC();
    return ok;
}

#if defined(__HIPCC__)
constexpr int GROUPS = 16;
__global__ __launch_bounds__(64, KYB_TU_WAVES) void bls12381_g2_mul_coop_kernel(size_t n, const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ pts,
                                                                  size_t pt_stride, uint8_t* __restrict__ out, uint8_t* __restrict__ status,
                                                                  uint32_t flags) {
    __shared__ Slot slots[GROUPS * NS];
    __shared__ Slot consts[NCONST];
    __shared__ uint32_t flg[GROUPS * 2];
    __shared__ int8_t dig[GROUPS][NH * NDIG];
    __shared__ int verdict[GROUPS];
    const int gi = (int)threadIdx.x >> 2, r = (int)threadIdx.x & 3;
    const size_t idx = (size_t)blockIdx.x * GROUPS + gi;
    const bool live = idx < n;
    const size_t ic = live ? idx : n - 1;
    Slot* S = slots + gi * NS;
    const bool check = !flag_trusted(flags, 0);
    if (threadIdx.x == 1) constants(consts);  // (a lane that does no decoding)
    if (r == 0) {
        g2_aff a;
        const uint8_t* in = pts + pt_stride * ic;
        const int st = (flags & FLAG_UNCOMPRESSED) ? g2_decode_unc(a, in, check, false) : g2_decode(a, in, false);
        verdict[gi] = st | ((st == ST_OK && a.inf) ? 0x100 : 0);
        if (st != ST_OK || a.inf) {
            fp2_load_const<TC>(a.x, CC::G2X);
            fp2_load_const<TC>(a.y, CC::G2Y);
        }
        uint32_t k[8];
        scalar_from_be(k, scalars + 32 * ic);
        int8_t e[NH][NDIG];
        digits(e, k);
#pragma unroll 1
        for (int h = 0; h < NH; h++)
#pragma unroll 1
            for (int i = 0; i < NDIG; i++) dig[gi][h * NDIG + i] = e[h][i];
        S[TAB].f = a.x;
        S[TAB + 1].f = a.y;
        fp2 one;
        fp2_one(one);
        S[TAB + 2].f = one;
    }
    __syncthreads();
    if (check) {
        const bool in_g2 = member(S, consts, flg + gi * 2, r);
        if (r == 0 && verdict[gi] == 0 && !in_g2) {
            verdict[gi] = ST_NOT_IN_SUBGROUP;
            fp2_load_const<TC>(S[TAB].f, CC::G2X);
            fp2_load_const<TC>(S[TAB + 1].f, CC::G2Y);
        }
        __syncthreads();
    }
    ladder(S, consts, flg + gi * 2, r, dig[gi]);
    if (r == 0 && live) {
        uint8_t* o = out + g2_out_size(flags) * idx;
        const int v = verdict[gi];
        if (v & 0xff) {
            zero_bytes(o, (int)g2_out_size(flags));
        } else {
            g2_aff a;
            if (v & 0x100) {
                fp2_zero(a.x);
                fp2_zero(a.y);
                a.inf = true;
            } else {
                g2_jac p;
                p.X = S[ACC].f;
                p.Y = S[ACC + 1].f;
                p.Z = S[ACC + 2].f;
                jac_to_aff(a, p);
            }
            g2_encode_f(o, a, flags);
        }
        if (status) status[idx] = (uint8_t)(v & 0xff);
    }
}
__global__ __launch_bounds__(64, KYB_TU_WAVES) void bls12381_g2_unmarshal_coop_kernel(size_t n, const uint8_t* __restrict__ pts, uint8_t* __restrict__ out,
                                                                        uint8_t* __restrict__ status, uint32_t flags) {
    constexpr int MS = TAB + 3;
    __shared__ Slot slots[GROUPS * MS];
    __shared__ Slot consts[NCONST];
    __shared__ uint32_t flg[GROUPS * 2];
    const int gi = (int)threadIdx.x >> 2, r = (int)threadIdx.x & 3;
    const size_t idx = (size_t)blockIdx.x * GROUPS + gi;
    const bool live = idx < n;
    const size_t ic = live ? idx : n - 1;
    Slot* S = slots + gi * MS;
    const bool check = !flag_trusted(flags, 0);
    if (threadIdx.x == 1) constants(consts);
    g2_aff a;
    int st = ST_OK;
    if (r == 0) {
        const uint8_t* in = pts + g2_wire_size(flags) * ic;
        st = (flags & FLAG_UNCOMPRESSED) ? g2_decode_unc(a, in, check, false) : g2_decode(a, in, false);
        if (st == ST_OK && !a.inf) {
            S[TAB].f = a.x;
            S[TAB + 1].f = a.y;
        } else {
            fp2_load_const<TC>(S[TAB].f, CC::G2X);
            fp2_load_const<TC>(S[TAB + 1].f, CC::G2Y);
        }
        fp2_one(S[TAB + 2].f);
    }
    __syncthreads();
    bool in_g2 = true;
    if (check) in_g2 = member(S, consts, flg + gi * 2, r);
    if (r == 0 && live) {
        if (st == ST_OK && !a.inf && !in_g2) st = ST_NOT_IN_SUBGROUP;
        uint8_t* o = out + g2_out_size(flags) * idx;
        if (st != ST_OK) zero_bytes(o, (int)g2_out_size(flags));
        else g2_encode_f(o, a, flags);
        if (status) status[idx] = (uint8_t)st;
    }
}
#endif

}  // namespace g2coop
}  // namespace bls
}  // namespace kyb
