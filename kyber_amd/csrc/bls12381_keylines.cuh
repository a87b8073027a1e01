// The Miller lines of a FIXED G2 point, for the tower machine's same-key verification program (gen_tower_vm.py
// build_bls12381_verify_same_key; sign/bls/bls.go:82-96 called for many messages under one public key X):
// e(H(m), X) has X fixed, so the coefficients of every line of its Miller loop depend on X alone.  One lane walks X through
// the loop once per key -- the same formulas as gen_tower_vm.py bls_fixed_line_table, which makes the GENERATOR's table at
// build time -- and leaves, per step, (c2, c3) with  line = 1 + (c2 xP) w^2 + (c3 yP) w^3:
//   doubling  T = (X : Y : Z):  l0 = Y^2 - 12 xi Z^2,  c2 = -3 X^2 / l0,  c3 = 2 Y Z / l0
//   addition  T + Q:            theta = Y - yQ Z, lambda = X - xQ Z,  l0 = theta xQ - lambda yQ,  c2 = -theta / l0,  c3 = lambda / l0
// (the general line divided by l0: an Fp2 factor dies in the final exponentiation).  The 68 divisions share ONE
// inversion (Montgomery's trick over the l0's).  Output: 68 x 4 base-field elements as the plain integers v 2^392 mod p
// (12 little-endian words each) -- the machine's radix; the kernel turns them into its signed limbs.
#pragma once
#include "bls12381.cuh"

namespace kyb {
namespace bls {

constexpr int KEYLINE_STEPS = 68;  // 63 doublings + 5 additions: the bits of |z| below the top one, its ones among them
static_assert(CC::X_ABS == 0xd201000000010000ull, "the step count is the parameter's");

// q: a finite affine point of the twist (a decoded public key).  out[step][4][12]: c2.re, c2.im, c3.re, c3.im.
// Returns false when some l0 vanished (then X is a point of small order: never a member of G2) -- the table is unusable.
KYB_HD_NOINLINE bool g2_key_lines(uint32_t (*out)[4][12], const g2_aff& q) {
    fp2 X = q.x, Y = q.y, Z, l0[KEYLINE_STEPS], pre[KEYLINE_STEPS];
    fp2_one(Z);
    // raw (c2, c3) land in `out` as Montgomery words first; the second pass scales them by 1 / l0
    auto put = [&](int s, const fp2& c2, const fp2& c3) {
#pragma unroll
        for (int k = 0; k < 12; k++) {
            out[s][0][k] = c2.c0.v[k];
            out[s][1][k] = c2.c1.v[k];
            out[s][2][k] = c3.c0.v[k];
            out[s][3][k] = c3.c1.v[k];
        }
    };
    int s = 0;
#pragma unroll 1
    for (int bit = 62; bit >= 0; bit--) {
        fp2 XY, B, YZ, A3, E, E3, t, u, c2, c3;
        fp2_mul_c(XY, X, Y);
        fp2_sqr_c(B, Y);
        fp2_mul_c(YZ, Y, Z);
        fp2_sqr_c(t, X);
        fp2_dbl(A3, t);
        fp2_add(A3, A3, t);  // 3 X^2
        fp2_sqr_c(t, Z);
        fp_sub(u.c0, t.c0, t.c1);  // Z^2 (1 + i)
        fp_add(u.c1, t.c0, t.c1);
        fp2_dbl(E, u);
        fp2_dbl(E, E);
        fp2_dbl(t, E);
        fp2_add(E, E, t);  // 12 xi Z^2 = 3 b' Z^2
        fp2_sub(l0[s], B, E);
        fp2_neg(c2, A3);
        fp2_dbl(c3, YZ);
        put(s, c2, c3);
        s++;
        // T <- 2 T:  X3 = 2 XY (B - 3E), Y3 = B^2 + 3E (2B - E), Z3 = 8 B YZ
        fp2_dbl(E3, E);
        fp2_add(E3, E3, E);
        fp2_sub(t, B, E3);
        fp2_dbl(u, XY);
        fp2_mul_c(X, u, t);
        fp2_dbl(t, B);
        fp2_sub(t, t, E);
        fp2_mul_c(t, E3, t);
        fp2_sqr_c(u, B);
        fp2_add(Y, u, t);
        fp2_mul_c(t, B, YZ);
        fp2_dbl(t, t);
        fp2_dbl(t, t);
        fp2_dbl(Z, t);
        if ((CC::X_ABS >> bit) & 1ull) {
            fp2 TH, LA, C, D, Ee, Ff, Gg;
            fp2_mul_c(t, q.y, Z);
            fp2_sub(TH, Y, t);
            fp2_mul_c(t, q.x, Z);
            fp2_sub(LA, X, t);
            fp2_mul_c(t, TH, q.x);
            fp2_mul_c(u, LA, q.y);
            fp2_sub(l0[s], t, u);
            fp2_neg(c2, TH);
            put(s, c2, LA);
            s++;
            fp2_sqr_c(C, TH);
            fp2_sqr_c(D, LA);
            fp2_mul_c(Ee, LA, D);
            fp2_mul_c(Ff, Z, C);
            fp2_mul_c(Gg, X, D);
            fp2_add(t, Ee, Ff);  // E + F
            fp2_dbl(u, Gg);
            fp2_sub(u, t, u);
            fp2_mul_c(X, LA, u);  // lambda (E + F - 2G)
            fp2_dbl(u, Gg);
            fp2_add(u, u, Gg);
            fp2_sub(u, u, t);
            fp2_mul_c(u, TH, u);
            fp2_mul_c(t, Ee, Y);
            fp2_sub(Y, u, t);  // theta (3G - E - F) - E Y
            fp2_mul_c(Z, Z, Ee);
        }
    }
    // 1 / l0[s] for every step with one inversion: prefix products, invert the last, walk back
    pre[0] = l0[0];
#pragma unroll 1
    for (int k = 1; k < KEYLINE_STEPS; k++) fp2_mul_c(pre[k], pre[k - 1], l0[k]);
    if (fp2_is_zero(pre[KEYLINE_STEPS - 1])) return false;
    fp2 inv;
    fp2_inv(inv, pre[KEYLINE_STEPS - 1]);
#pragma unroll 1
    for (int k = KEYLINE_STEPS - 1; k >= 0; k--) {
        fp2 li = inv;
        if (k > 0) {
            fp2_mul_c(li, inv, pre[k - 1]);  // 1 / l0[k]
            fp2_mul_c(inv, inv, l0[k]);      // 1 / (l0[0] .. l0[k-1])
        }
        fp2 c2, c3;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            c2.c0.v[j] = out[k][0][j];
            c2.c1.v[j] = out[k][1][j];
            c3.c0.v[j] = out[k][2][j];
            c3.c1.v[j] = out[k][3][j];
        }
        fp2_mul_c(c2, c2, li);
        fp2_mul_c(c3, c3, li);
        // Montgomery residue v 2^390 -> the plain integer v 2^392 mod p: two modular doublings of the stored words
        fp2_dbl(c2, c2);
        fp2_dbl(c2, c2);
        fp2_dbl(c3, c3);
        fp2_dbl(c3, c3);
        put(k, c2, c3);
    }
    return true;
}

#if defined(KYB_ROWFP_INCLUDED)
// ---- The same walk with ONE LIMB PER LANE (rowfp.cuh), one wave -- round 6.  A lone lane needs ~9 ms for the 890 Fp2
// products above (the first sight of a key by kyb_bls12381_verify_g1_same_key); here an Fp2 product is one LEVEL of the
// wave's four rows (rowfp.cuh f2_mul) and the walk ~1 000 levels.  Same formulas, same outputs word for word
// (tests/test_host_harness_rowfp.py runs both on the CPU); the doubling's B = Y^2 is a product (Y comes out of an addition
// step below 13p: the squaring form's (y0 + y1)(y0 - y1) would not fit R / p) and 12 xi Z^2 is a product by the constant
// 12 (1 + i) (the sums 12 u, 36 u would not either).
// Value bounds (multiples of p, (c0, c1)), products marked M = (5, 4):
//   doubling, X (5,4) Y (13,12) Z (5,4):  XY, YZ, B = YY: M (5 x 13, 13 x 5, 169);  X^2, Z^2: (2,4) (s < 9, d < 13);
//     E = Z^2 C12: M, 3E (15,12);  l0 = B - E + 8p (13,12);  c2 = 3 (8p - X^2) (24,24), c3 = 2YZ (10,8);
//     X3 = 2XY (B - 3E + 16p): (10,8) x (21,20) = 210 -> M;  Y3 = B^2 + 3E (2B - E + 8p): (15,12) x (18,16) = 270 -> (7,8);
//     Z3 = (2B)(4YZ): (10,8) x (20,16) = 200 -> M
//   addition, X (5,4) Y (7,8) Z (5,4), q (1,1):  TH = Y - yq Z + 8p (15,16), LA = X - xq Z + 8p (13,12);
//     l0 = TH xq - LA yq + 8p (13,12);  c2 = 24p - TH (24,24), c3 = LA;  C = TH TH (256), D = LA LA (169), E = LA D, F = Z C,
//     G = X D: M;  X' = LA (E + F - 2G + 16p): 13 x 26 = 338;  Y' = TH (3G - E - F + 16p) - E Y + 8p: 16 x 31 = 496, (13,12);
//     Z' = Z E: M
//   second pass: pre = pre l0 (5 x 13), 1/l0 = inv pre (25), inv = inv l0, c2 / l0 (24 x 5), c3 / l0 (13 x 5); x 4 below 22p.
struct KeyLinesMem {  // 35 KB of LDS (device) for the per-step values the second pass needs
    uint32_t l0[KEYLINE_STEPS][2][rowfp::ROW], pre[KEYLINE_STEPS][2][rowfp::ROW], c2[KEYLINE_STEPS][2][rowfp::ROW],
        c3[KEYLINE_STEPS][2][rowfp::ROW];
    uint32_t fin[4][rowfp::ROW];
    uint32_t inv[2][12];
    uint32_t tw[6][12];  // the walk's end T = |z| q, homogeneous (X, Y, Z), packed words (x0, x1, y0, y1, z0, z1)
    uint32_t ok;
};
// The r-torsion rule read off the walk's end (round 6): psi(q) = [z] q, z < 0, i.e. T = |z| q = -psi(q) (g2_in_subgroup's
// test, bls12381.cuh -- whose own 63 doublings and 5 additions in one lane the walk has just done on the rows).  A walk
// that degenerated (T = +-q at an addition step, a doubled 2-torsion point: only off the subgroup) ends with Z = 0.
KYB_HD bool g2_walk_end_is_minus_psi(const uint32_t (*tw)[12], const uint32_t* qx0, const uint32_t* qx1, const uint32_t* qy0,
                                      const uint32_t* qy1) {
    fp2 x, y, X, Y, Z, cx, cy, px, py, l, r;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        x.c0.v[w] = qx0[w];
        x.c1.v[w] = qx1[w];
        y.c0.v[w] = qy0[w];
        y.c1.v[w] = qy1[w];
        X.c0.v[w] = tw[0][w];
        X.c1.v[w] = tw[1][w];
        Y.c0.v[w] = tw[2][w];
        Y.c1.v[w] = tw[3][w];
        Z.c0.v[w] = tw[4][w];
        Z.c1.v[w] = tw[5][w];
    }
    fp2_load_const<TC>(cx, CC::PSI_CX);
    fp2_load_const<TC>(cy, CC::PSI_CY);
    fp2_conj(px, x);
    fp2_mul_c(px, px, cx);
    fp2_conj(py, y);
    fp2_mul_c(py, py, cy);
    fp2_neg(py, py);
    fp2_mul_c(l, px, Z);
    fp2_mul_c(r, py, Z);
    return fp2_eq(l, X) & fp2_eq(r, Y) & !fp2_is_zero(Z);
}
// ---- UnmarshalBinary of a COMPRESSED key without the r-torsion rule (g2_decode(a, in, false), bls12381.cuh -- restated here
// so that the batch kernels' copy keeps its shape) with the square root's two 379-bit powers on the rows: lane 0 parses,
// forms x^3 + b and its norm; the rows raise it to (p + 1) / 4; lane 0 forms t = (a0 + s) / 2; the rows raise that to
// (p - 3) / 4; lane 0 finishes as fp2_sqrt_from_norm_root does and picks the sign.  Every lane of the one-wave workgroup
// calls it.  Returns the status (in every lane); *inf_out for the point at infinity; the point as packed words in qw[4].
struct KeyDecodeMem {
    uint32_t tab[15][rowfp::ROW];
    uint32_t io[12], fin[rowfp::ROW];
    fp2 x, rhs;
    fp t;
    int st, inf, go, sflag;
};
KYB_ROW int g2_decode_rows(KeyDecodeMem& m, uint32_t (*qw)[12], int* inf_out, const uint8_t* in) {
    using namespace rowfp;
    using C = FC;
    KYB_ROW_LONE {
        uint32_t w1[12], w0[12];
        words_from_be<12>(w1, in);
        words_from_be<12>(w0, in + 48);
        const uint32_t top = w1[11] >> 29;
        const bool c = top & 4, inf = top & 2, s = top & 1;
        w1[11] &= 0x1fffffffu;
        uint32_t any = 0;
        for (int k = 0; k < 12; k++) any |= w1[k] | w0[k];
        m.inf = 1;
        m.go = 0;
        m.sflag = s ? 1 : 0;
        if (!c) m.st = ST_BAD_POINT;
        else if (inf) m.st = (s || any) ? ST_BAD_POINT : ST_OK;
        else if (!fp_words_lt_p<FC>(w1) || !fp_words_lt_p<FC>(w0)) m.st = ST_BAD_POINT;
        else {
            fp2 x, rhs, b;
            fp n, t;
            fp_from_words<FC>(x.c0, w0);
            fp_from_words<FC>(x.c1, w1);
            fp2_load_const<TC>(b, CC::B2);
            fp2_sqr_c(rhs, x);
            fp2_mul_c(rhs, rhs, x);
            fp2_add(rhs, rhs, b);
            fp_sqr(n, rhs.c0);
            fp_sqr(t, rhs.c1);
            fp_add(n, n, t);
            m.x = x;
            m.rhs = rhs;
            for (int k = 0; k < 12; k++) m.io[k] = n.v[k];
            m.st = ST_OK;
            m.go = 1;
        }
    }
    row_sync();
    if (m.go) {  // (uniform)
        const auto cx = make_ctx<C>();
        V32 r = below_2p<C>(cx, pow_words<C>(cx, load_packed<C>(m.io), m.tab, FC::SQRT_EXP, FC::SQRT_BITS));
        store_row(m.fin, r);
        row_sync();
        KYB_ROW_LONE {
            fp s, t, inv2;
            finish_limbs<C>(s, m.fin);  // a root of the norm if it has one (checked with the final square below)
            fp_const(inv2, FC::INV2);
            fp_add(t, m.rhs.c0, s);
            fp_mul(t, t, inv2);
            fp_cmov(t, m.rhs.c0, fp_is_zero(m.rhs.c1));
            m.t = t;
            for (int k = 0; k < 12; k++) m.io[k] = t.v[k];
        }
        row_sync();
        r = below_2p<C>(cx, pow_words<C>(cx, load_packed<C>(m.io), m.tab, FC::PM3D4, FC::SQRT_BITS));  // t^((p-3)/4)
        store_row(m.fin, r);
        row_sync();
        KYB_ROW_LONE {
            fp u, c, c2, h, inv2;
            const fp t = m.t;
            const fp2 a = m.rhs;
            finish_limbs<C>(u, m.fin);
            fp_const(inv2, FC::INV2);
            fp_mul(c, u, t);  // t^((p+1)/4)
            fp_sqr(c2, c);
            const bool qr = fp_eq(c2, t);  // chi(t) = +1 ; otherwise c^2 = -t and 1/c = -u
            fp_mul(h, a.c1, inv2);
            fp_mul(h, h, u);  // a1 / (2c) up to the sign chi
            fp2 y, chk, ny;
            if (qr) {
                y.c0 = c;
                y.c1 = h;
            } else {
                fp_neg(y.c0, h);
                y.c1 = c;
            }
            fp2_sqr(chk, y);
            if (!fp2_eq(chk, a)) {
                m.st = ST_BAD_POINT;
            } else {
                fp2_neg(ny, y);
                fp2_cmov(y, ny, fp2_is_larger(y) != (m.sflag != 0));
                for (int k = 0; k < 12; k++) {
                    qw[0][k] = m.x.c0.v[k];
                    qw[1][k] = m.x.c1.v[k];
                    qw[2][k] = y.c0.v[k];
                    qw[3][k] = y.c1.v[k];
                }
                m.inf = 0;
            }
        }
        row_sync();
    }
    *inf_out = m.inf;
    return m.st;
}

// q: a finite affine point of the twist, packed words qx0, qx1, qy0, qy1 (Montgomery form).  Every lane of the (one-wave)
// workgroup calls it; out as g2_key_lines.  Returns false (in every lane) when some l0 vanished -- or, with member_test, when
// the point fails the r-torsion rule at the walk's end (before the second pass: nothing is written to out then).
KYB_ROW bool g2_key_lines_rows(KeyLinesMem& m, uint32_t (*out)[4][12], const uint32_t* qx0, const uint32_t* qx1, const uint32_t* qy0,
                               const uint32_t* qy1, bool member_test = false) {
    using namespace rowfp;
    using C = FC;
    using E2 = F2<C>;
    const auto cx = make_ctx<C>();
    const auto k = make_f2_consts<C>();
    const V32 row = row_of_lane();
    const V32 zero = splat(0u);
    const E2 qx{load_packed<C>(qx0), load_packed<C>(qx1)}, qy{load_packed<C>(qy0), load_packed<C>(qy1)};
    const V32 twelve = lane_table(K<C>::small_limbs(12));
    const E2 c12{twelve, twelve};  // 12 (1 + i)
    E2 X = qx, Y = qy, Z{lane_table(K<C>::one_limbs()), zero};
    int s = 0;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int bit = 62; bit >= 0; bit--) {
        const E2 XY = f2_mul<C>(cx, k, row, X, Y), YZ = f2_mul<C>(cx, k, row, Y, Z), B = f2_mul<C>(cx, k, row, Y, Y);
        E2 A, Zs;
        f2_sqr2<C>(cx, row, A, X, k.b8, Zs, Z, k.b8);
        const E2 Ee = f2_mul<C>(cx, k, row, Zs, c12), E3 = f2_triple<C>(Ee);
        f2_store<C>(m.l0[s], f2_sub<C>(B, Ee, k.b8));
        f2_store<C>(m.c2[s], f2_triple<C>(f2_neg<C>(A, k.b8)));
        f2_store<C>(m.c3[s], f2_dbl<C>(YZ));
        s++;
        E2 BB, unused;
        f2_sqr2<C>(cx, row, BB, B, k.b8, unused, B, k.b8);
        X = f2_mul<C>(cx, k, row, f2_dbl<C>(XY), f2_sub<C>(B, E3, k.b16));
        Y = f2_add<C>(BB, f2_mul<C>(cx, k, row, E3, f2_sub<C>(f2_dbl<C>(B), Ee, k.b8)));
        Z = f2_mul<C>(cx, k, row, f2_dbl<C>(B), f2_dbl<C>(f2_dbl<C>(YZ)));
        if ((CC::X_ABS >> bit) & 1ull) {
            const E2 TH = f2_sub<C>(Y, f2_mul<C>(cx, k, row, qy, Z), k.b8), LA = f2_sub<C>(X, f2_mul<C>(cx, k, row, qx, Z), k.b8);
            f2_store<C>(m.l0[s], f2_sub<C>(f2_mul<C>(cx, k, row, TH, qx), f2_mul<C>(cx, k, row, LA, qy), k.b8));
            f2_store<C>(m.c2[s], f2_neg<C>(TH, k.b24));
            f2_store<C>(m.c3[s], LA);
            s++;
            const E2 Cc = f2_mul<C>(cx, k, row, TH, TH), D = f2_mul<C>(cx, k, row, LA, LA);
            const E2 Ea = f2_mul<C>(cx, k, row, LA, D), Ff = f2_mul<C>(cx, k, row, Z, Cc), Gg = f2_mul<C>(cx, k, row, X, D);
            const E2 EF = f2_add<C>(Ea, Ff);
            X = f2_mul<C>(cx, k, row, LA, f2_sub<C>(EF, f2_dbl<C>(Gg), k.b16));
            const E2 t = f2_mul<C>(cx, k, row, TH, f2_sub<C>(f2_triple<C>(Gg), EF, k.b16));
            Y = f2_sub<C>(t, f2_mul<C>(cx, k, row, Ea, Y), k.b8);
            Z = f2_mul<C>(cx, k, row, Z, Ea);
        }
    }
    row_sync();
    const V32 one = lane_table(K<C>::one_limbs());
    {  // T = (X, Y, Z) below 2p, then as packed residues (X (5,4), Y (13,12), Z (5,4): all below below_2p's 22p)
        V32 p0, p1, p2, p3;
        level4<C>(cx, row, X.c0, one, X.c1, one, Y.c0, one, Y.c1, one, p0, p1, p2, p3);
        store_row(m.fin[0], p0);
        store_row(m.fin[1], p1);
        store_row(m.fin[2], p2);
        store_row(m.fin[3], p3);
        row_sync();
        KYB_ROW_LANES(4) {
            fp f;
            finish_limbs<C>(f, m.fin[j_]);
            for (int w = 0; w < 12; w++) m.tw[j_][w] = f.v[w];
        }
        row_sync();
        level4<C>(cx, row, Z.c0, one, Z.c1, one, Z.c0, one, Z.c1, one, p0, p1, p2, p3);
        store_row(m.fin[0], p0);
        store_row(m.fin[1], p1);
        row_sync();
        KYB_ROW_LANES(2) {
            fp f;
            finish_limbs<C>(f, m.fin[j_]);
            for (int w = 0; w < 12; w++) m.tw[4 + j_][w] = f.v[w];
        }
        row_sync();
        if (member_test) {  // (uniform)
            KYB_ROW_LONE { m.ok = g2_walk_end_is_minus_psi(m.tw, qx0, qx1, qy0, qy1) ? 1u : 0u; }
            row_sync();
            if (!m.ok) return false;
            row_sync();  // (m.ok is written again below)
        }
    }
    // prefix products of the l0's
    E2 pre = f2_load<C>(m.l0[0]);
    f2_store<C>(m.pre[0], pre);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int j = 1; j < KEYLINE_STEPS; j++) {
        pre = f2_mul<C>(cx, k, row, pre, f2_load<C>(m.l0[j]));
        f2_store<C>(m.pre[j], pre);
    }
    // the one inversion, in the packed form by one lane (zero: some l0 vanished)
    {
        V32 p0, p1, p2, p3;
        level4<C>(cx, row, pre.c0, one, pre.c1, one, pre.c0, one, pre.c1, one, p0, p1, p2, p3);
        store_row(m.fin[0], p0);
        store_row(m.fin[1], p1);
    }
    row_sync();
    KYB_ROW_LONE {
        fp2 v, vi;
        finish_limbs<C>(v.c0, m.fin[0]);
        finish_limbs<C>(v.c1, m.fin[1]);
        m.ok = fp2_is_zero(v) ? 0u : 1u;
        fp2_inv(vi, v);
        for (int w = 0; w < 12; w++) {
            m.inv[0][w] = vi.c0.v[w];
            m.inv[1][w] = vi.c1.v[w];
        }
    }
    row_sync();
    if (!m.ok) return false;
    E2 inv{load_packed<C>(m.inv[0]), load_packed<C>(m.inv[1])};
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int j = KEYLINE_STEPS - 1; j >= 0; j--) {
        E2 li = inv;
        if (j > 0) {
            li = f2_mul<C>(cx, k, row, inv, f2_load<C>(m.pre[j - 1]));  // 1 / l0[j]
            inv = f2_mul<C>(cx, k, row, inv, f2_load<C>(m.l0[j]));      // 1 / (l0[0] .. l0[j-1])
        }
        // Montgomery residue v 2^390 -> the plain integer v 2^392 mod p: four times the value, then the canonical words
        const E2 c2 = f2_dbl<C>(f2_dbl<C>(f2_mul<C>(cx, k, row, f2_load<C>(m.c2[j]), li)));
        const E2 c3 = f2_dbl<C>(f2_dbl<C>(f2_mul<C>(cx, k, row, f2_load<C>(m.c3[j]), li)));
        V32 p0, p1, p2, p3;
        level4<C>(cx, row, c2.c0, one, c2.c1, one, c3.c0, one, c3.c1, one, p0, p1, p2, p3);
        row_sync();  // (the lanes below have read the previous step's fin)
        store_row(m.fin[0], p0);
        store_row(m.fin[1], p1);
        store_row(m.fin[2], p2);
        store_row(m.fin[3], p3);
        row_sync();
        KYB_ROW_LANES(4) {
            fp f;
            finish_limbs<C>(f, m.fin[j_]);
            for (int w = 0; w < 12; w++) out[j][j_][w] = f.v[w];
        }
    }
    row_sync();
    return true;
}
#endif

}  // namespace bls
}  // namespace kyb
