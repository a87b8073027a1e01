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

}  // namespace bls
}  // namespace kyb
