// Jacobian ladders in LAZY LIMB FORM (fp_limbs.cuh) for the per-lane scalar multiplications of the BN suites -- G1 over
// Fp and G2 over Fp2 through one set of formulas.
//
// Why another form.  The packed ladders (curve.cuh + tower.cuh) reduce every intermediate to [0, p): an Fp2
// multiplication is three base-field multiplications, each with two unpacks, a pack and a conditional subtraction, plus
// five additions with trial subtractions; and they CALL the Fp2 multiplier, whose caller spills its live point around every
// call (bn256 G2 Mul, round 4: 3 664 B of scratch per lane, 127 KB of scratch traffic per element, VALU busy 0.76).
// Here a coordinate stays in the multiplier's own limbs from the first product of the ladder to the last:
//   * limbs of 30 bits instead of the packed code's 29 (Limb30<C>: the same nine limbs for a 254 / 256-bit prime, but
//     R' = 2^270, so R' / p > 2^14 where the 29-bit form has 2^5): sums and differences need NO reduction between
//     products -- a subtraction adds a multiple of p, the bounds below stay under R' / p by themselves;
//   * an Fp2 product is c0 = a0 b0 + a1 (K p - b1), c1 = a0 b1 + a1 b0: two two-product multiplications with ONE
//     reduction each (fpl_mul2sum; 486 multiply-adds like Karatsuba's three, none of its five additions);
//   * everything is inlined into two loop bodies (a doubling, a mixed addition); the table of the window method and the
//     digits are the only private memory.
// Elements enter through one multiplication by 2^(2 N' W' - N W) (Montgomery form for R') and leave through one by R.
//
// Bounds (multiples of p, per base-field coefficient; LzFp2's squaring returns c1 = 2 a0 a1 < 4), proved for the two
// formulas as they are written below and CHECKED at run time in the host harness (-DKYB_LZ_AUDIT, tests/test_lazy_bounds.py):
//   between operations X < 52, Y < 36, Z < 12;  the largest product sum is r^2 in the mixed addition: (2 * 76)^2 = 23 104
//   against R' / p = 29 200 (bn256) / 86 600 (bn254).
#pragma once
#include <utility>
#include "curve.cuh"

namespace kyb {

// ---- the 30-bit limb configuration of a packed field configuration C, derived at compile time
namespace limb30 {
template <class C>
constexpr uint32_t p_limb(int j) {  // bits [30 j, 30 j + 30) of p
    const int bit = 30 * j, idx = bit >> 5, sh = bit & 31;
    uint64_t x = idx < C::NWORDS ? C::PW[idx] : 0u;
    if (idx + 1 < C::NWORDS) x |= (uint64_t)C::PW[idx + 1] << 32;
    return (uint32_t)((x >> sh) & 0x3fffffffu);
}
template <class C>
constexpr uint32_t ninv() {  // -p^-1 mod 2^30
    const uint32_t p0 = C::PW[0];
    uint32_t x = p0;
    for (int k = 0; k < 5; k++) x *= 2u - p0 * x;
    return (0u - x) & 0x3fffffffu;
}
template <class C>
constexpr uint32_t pow2_limb(int e, int j) {  // limb j (30 bits) of 2^e mod p
    constexpr int NW = C::NWORDS + 1;
    uint32_t v[NW] = {};
    v[0] = 1;
    for (int it = 0; it < e; it++) {
        uint32_t c = 0;
        for (int i = 0; i < NW; i++) {  // v *= 2
            const uint32_t n = (v[i] << 1) | c;
            c = v[i] >> 31;
            v[i] = n;
        }
        bool ge = true;  // v >= p ?
        for (int i = NW - 1; i >= 0; i--) {
            const uint32_t pi = i < C::NWORDS ? C::PW[i] : 0u;
            if (v[i] != pi) {
                ge = v[i] > pi;
                break;
            }
        }
        if (ge) {
            uint64_t b = 0;
            for (int i = 0; i < NW; i++) {
                const uint64_t pi = i < C::NWORDS ? C::PW[i] : 0u;
                const uint64_t d = (uint64_t)v[i] - pi - b;
                v[i] = (uint32_t)d;
                b = (d >> 63) & 1u;
            }
        }
    }
    const int bit = 30 * j, idx = bit >> 5, sh = bit & 31;
    uint64_t x = idx < NW ? v[idx] : 0u;
    if (idx + 1 < NW) x |= (uint64_t)v[idx + 1] << 32;
    return (uint32_t)((x >> sh) & 0x3fffffffu);
}
template <class C, int NL, class Idx>
struct Arrays;
template <class C, int NL, size_t... I>
struct Arrays<C, NL, std::index_sequence<I...>> {
    static constexpr uint32_t P[NL] = {p_limb<C>((int)I)...};
    static constexpr uint32_t CONV_IN[NL] = {pow2_limb<C>(2 * NL * 30 - C::N * C::W, (int)I)...};  // R'^2 / R
    static constexpr uint32_t CONV_OUT[NL] = {pow2_limb<C>(C::N * C::W, (int)I)...};               // R
    static constexpr uint32_t ONE[NL] = {pow2_limb<C>(NL * 30, (int)I)...};                        // R' mod p
};
}  // namespace limb30

template <class C>
struct Limb30 {
    static constexpr int W = 30, PBITS = C::PBITS, NWORDS = C::NWORDS;
    static constexpr int N = (C::PBITS + 14 + 29) / 30;  // at least 14 bits of headroom
    using A = limb30::Arrays<C, N, std::make_index_sequence<N>>;
    static constexpr const uint32_t (&P)[N] = A::P;
    static constexpr const uint32_t (&PW)[NWORDS] = C::PW;
    static constexpr uint32_t NINV = limb30::ninv<C>();
    using Packed = C;
};

// packed Montgomery element (R = 2^(N W) of C) -> lazy limbs in Montgomery form for R' = 2^(30 N'); value < 2p
template <class C>
KYB_HD void lz_enter(FpL<Limb30<C>>& r, const Fp<C>& a) {
    using L = Limb30<C>;
    FpL<L> t, c;
#pragma unroll
    for (int j = 0; j < L::N; j++) {
        const int bit = 30 * j, idx = bit >> 5, sh = bit & 31;
        uint32_t x = idx < C::NWORDS ? (a.v[idx] >> sh) : 0u;
        if (sh > 2 && idx + 1 < C::NWORDS) x |= a.v[idx + 1] << (32 - sh);
        t.l[j] = x & 0x3fffffffu;
        c.l[j] = L::A::CONV_IN[j];
    }
    KYB_LZ_K(t.k = 1.0; c.k = 1.0;)
    fpl_mul(r, t, c);
}
// lazy limbs (any bound the multiplier takes against R mod p < p) -> packed, fully reduced
template <class C>
KYB_HD void lz_leave(Fp<C>& r, const FpL<Limb30<C>>& a) {
    using L = Limb30<C>;
    FpL<L> c, t;
#pragma unroll
    for (int j = 0; j < L::N; j++) c.l[j] = L::A::CONV_OUT[j];
    KYB_LZ_K(c.k = 1.0;)
    fpl_mul(t, a, c);  // a R' * R / R' = a R, below 2p
    uint32_t w[C::NWORDS + 1];
#pragma unroll
    for (int k = 0; k <= C::NWORDS; k++) {
        const int bit = 32 * k, j = bit / 30, o = bit - j * 30;
        uint32_t x = j < L::N ? (t.l[j] >> o) : 0u;
        if (j + 1 < L::N) x |= t.l[j + 1] << (30 - o);
        if (60 - o < 32 && j + 2 < L::N) x |= t.l[j + 2] << (60 - o);
        w[k] = x;
    }
    uint32_t d[C::NWORDS], borrow = 0;
#pragma unroll
    for (int k = 0; k < C::NWORDS; k++) d[k] = sbb32(w[k], C::PW[k], borrow);
    // below p exactly when the bits above the words are clear and the subtraction borrowed
    const uint32_t keep = (0u - borrow) & (w[C::NWORDS] ? 0u : ~0u);
#pragma unroll
    for (int k = 0; k < C::NWORDS; k++) r.v[k] = sel32(keep, w[k], d[k]);
}

// ---- lazy field adaptors: the operations the two formulas are written in
template <class L>
struct LzFp {
    using E = FpL<L>;
    using Packed = Fp<typename L::Packed>;
    KYB_HD static void enter(E& r, const Packed& a) { lz_enter(r, a); }
    KYB_HD static void leave(Packed& r, const E& a) { lz_leave(r, a); }
    KYB_HD static void one(E& r) {
#pragma unroll
        for (int j = 0; j < L::N; j++) r.l[j] = L::A::ONE[j];
        KYB_LZ_K(r.k = 1.0;)
    }
    template <int KB> KYB_HD static void mul(E& r, const E& a, const E& b) { fpl_mul(r, a, b); }  // KB: bound of b (unused here)
    template <int KA> KYB_HD static void sqr(E& r, const E& a) { fpl_sqr(r, a); }
    template <int KB> KYB_HD static void mulb(E& r, const E& a, const E& b) { fpl_mul(r, a, b); }  // the names jaclz_table8 uses on every adaptor
    template <int KA> KYB_HD static void sqrb(E& r, const E& a) { fpl_sqr(r, a); }
    // r = a b + c d, one reduction per coefficient; KB, KD: bounds of b, d
    template <int KB, int KD> KYB_HD static void mul2sum(E& r, const E& a, const E& b, const E& c, const E& d) { fpl_mul2sum(r, a, b, c, d); }
    KYB_HD static void add(E& r, const E& a, const E& b) { fpl_add(r, a, b); }
    KYB_HD static void add2x(E& r, const E& a, const E& b) { fpl_add_2x(r, a, b); }  // a + 2 b
    KYB_HD static void mul4(E& r, const E& a) { fpl_mul4(r, a); }
    template <int K> KYB_HD static void sub(E& r, const E& a, const E& b) { fpl_sub<K>(r, a, b); }
    template <int KMAX> KYB_HD static bool is_zero(const E& a) { return fpl_is_zero_mod_p<KMAX>(a); }
    template <int K> KYB_HD static void neg_if(E& r, const E& a, bool neg) {  // neg ? K p - a : a
        E z, n;
#pragma unroll
        for (int j = 0; j < L::N; j++) z.l[j] = 0;
        KYB_LZ_K(z.k = 0.0;)
        fpl_sub<K>(n, z, a);
#pragma unroll
        for (int j = 0; j < L::N; j++) r.l[j] = neg ? n.l[j] : a.l[j];
        KYB_LZ_K(r.k = neg ? n.k : a.k;)
    }
};
template <class L, class T>
struct LzFp2 {
    struct E {
        FpL<L> c0, c1;
    };
    using Packed = Fp2<T>;
    using B = LzFp<L>;
    KYB_HD static void enter(E& r, const Packed& a) {
        lz_enter(r.c0, a.c0);
        lz_enter(r.c1, a.c1);
    }
    KYB_HD static void leave(Packed& r, const E& a) {
        lz_leave(r.c0, a.c0);
        lz_leave(r.c1, a.c1);
    }
    KYB_HD static void one(E& r) {
        B::one(r.c0);
#pragma unroll
        for (int j = 0; j < L::N; j++) r.c1.l[j] = 0;
        KYB_LZ_K(r.c1.k = 0.0;)
    }
    // (a0 + a1 i)(b0 + b1 i), i^2 = -1: c0 = a0 b0 + a1 (KB p - b1), c1 = a0 b1 + a1 b0
    template <int KB>
    KYB_HD static void mul(E& r, const E& a, const E& b) {
        FpL<L> z, nb1, t0;
#pragma unroll
        for (int j = 0; j < L::N; j++) z.l[j] = 0;
        KYB_LZ_K(z.k = 0.0;)
        fpl_sub<KB>(nb1, z, b.c1);
        fpl_mul2sum(t0, a.c0, b.c0, a.c1, nb1);
        fpl_mul2sum(r.c1, a.c0, b.c1, a.c1, b.c0);
        r.c0 = t0;
    }
    // c0 = (a0 + a1)(a0 - a1 + KA p), c1 = 2 a0 a1   (c1 below 4p)
    template <int KA>
    KYB_HD static void sqr(E& r, const E& a) {
        FpL<L> s, d, m;
        fpl_add(s, a.c0, a.c1);
        fpl_sub<KA>(d, a.c0, a.c1);
        fpl_mul(m, a.c0, a.c1);
        fpl_mul(r.c0, s, d);
        fpl_add(r.c1, m, m);
    }
    // a b + c d in Fp2 with one reduction per coefficient: c0 = a0 b0 + a1 nb1 + c0 d0 + c1 nd1 -- four products per
    // column: done as two two-product multiplications and a lazy addition (a four-product walk would need a third
    // carry sweep of the 64-bit columns for the same multiply-adds)
    template <int KB, int KD>
    KYB_HD static void mul2sum(E& r, const E& a, const E& b, const E& c, const E& d) {
        E t, u;
        mul<KB>(t, a, b);
        mul<KD>(u, c, d);
        fpl_add(r.c0, t.c0, u.c0);
        fpl_add(r.c1, t.c1, u.c1);
    }
    KYB_HD static void add(E& r, const E& a, const E& b) {
        fpl_add(r.c0, a.c0, b.c0);
        fpl_add(r.c1, a.c1, b.c1);
    }
    KYB_HD static void add2x(E& r, const E& a, const E& b) {
        fpl_add_2x(r.c0, a.c0, b.c0);
        fpl_add_2x(r.c1, a.c1, b.c1);
    }
    KYB_HD static void mul4(E& r, const E& a) {
        fpl_mul4(r.c0, a.c0);
        fpl_mul4(r.c1, a.c1);
    }
    template <int K>
    KYB_HD static void sub(E& r, const E& a, const E& b) {
        fpl_sub<K>(r.c0, a.c0, b.c0);
        fpl_sub<K>(r.c1, a.c1, b.c1);
    }
    template <int KMAX>
    KYB_HD static bool is_zero(const E& a) {
        return fpl_is_zero_mod_p<KMAX>(a.c0) & fpl_is_zero_mod_p<KMAX>(a.c1);
    }
    template <int K>
    KYB_HD static void neg_if(E& r, const E& a, bool neg) {
        B::template neg_if<K>(r.c0, a.c0, neg);
        B::template neg_if<K>(r.c1, a.c1, neg);
    }
    template <int KB> KYB_HD static void mulb(E& r, const E& a, const E& b) { mul<KB>(r, a, b); }
    template <int KA> KYB_HD static void sqrb(E& r, const E& a) { sqr<KA>(r, a); }
    // by a base-field element / conjugation: the psi maps of the GLS walk
    KYB_HD static void mul_fp(E& r, const E& a, const FpL<L>& b) {
        fpl_mul(r.c0, a.c0, b);
        fpl_mul(r.c1, a.c1, b);
    }
    template <int K>
    KYB_HD static void conj(E& r, const E& a) {
        FpL<L> z;
#pragma unroll
        for (int j = 0; j < L::N; j++) z.l[j] = 0;
        KYB_LZ_K(z.k = 0.0;)
        r.c0 = a.c0;
        fpl_sub<K>(r.c1, z, a.c1);
    }
};

// ---- the point and the two formulas
template <class LF>
struct JacLz {
    typename LF::E X, Y, Z;
    uint32_t inf;  // the point at infinity is a flag: Z = 0 (mod p) is not readable off a lazy value for free
};
constexpr int LZ_KX = 52, LZ_KY = 36, LZ_KZ = 12;  // bounds of the coordinates between operations (see the header)

template <class LF>
KYB_HD void jaclz_set_inf(JacLz<LF>& r) {
    LF::one(r.X);
    LF::one(r.Y);
    LF::one(r.Z);
    r.inf = 1u;
}
// dbl-2009-l (a = 0), 2M + 5S.  In: X < 52, Y < 36, Z < 12.  Out: X < 52, Y < 36, Z < 4.  (Curves of odd order: no
// point has Y = 0.)
template <class LF>
KYB_HD void jaclz_dbl(JacLz<LF>& p) {
    using E = typename LF::E;
    E A, B, C, D, Ee, G, t, u;
    LF::template sqr<LZ_KX>(A, p.X);            // < 4
    LF::template sqr<LZ_KY>(B, p.Y);            // < 4
    LF::template sqr<4>(C, B);                  // < 4
    LF::add(t, p.X, B);                         // < 56
    LF::template sqr<56>(t, t);                 // < 4
    LF::add(u, A, C);                           // < 8
    LF::template sub<8>(t, t, u);               // (X + B)^2 - A - C  < 12
    LF::add(D, t, t);                           // 4 X Y^2  < 24
    LF::add2x(Ee, A, A);                        // 3 X^2  < 12
    LF::template sqr<12>(G, Ee);                // < 4
    LF::template mul<LZ_KZ>(t, p.Y, p.Z);       // < 4 (Fp2: each coefficient < 2)
    LF::add(p.Z, t, t);                         // < 4
    LF::add(u, D, D);                           // < 48
    LF::template sub<48>(p.X, G, u);            // X3 = G - 2 D  < 52
    LF::template sub<52>(t, D, p.X);            // D - X3  < 76
    LF::template mul<76>(t, Ee, t);             // < 2
    LF::mul4(u, C);                             // < 16
    LF::add(u, u, u);                           // 8 C  < 32
    LF::template sub<32>(p.Y, t, u);            // Y3 = E (D - X3) - 8 C  < 34
}
// madd-2007-bl with the exceptional cases: p += +-(x2, y2), a finite affine point with coordinates below 4p (table
// entries fresh from the multiplier).  In: X < 52, Y < 36, Z < 12.  Out: X < 16, Y < 4, Z < 12.
template <class LF>
KYB_HD void jaclz_madd(JacLz<LF>& p, const typename LF::E& x2, const typename LF::E& y2in, bool neg) {
    using E = typename LF::E;
    E y2;
    LF::template neg_if<4>(y2, y2in, neg);  // < 4
    if (p.inf) {
        p.X = x2;
        p.Y = y2;
        LF::one(p.Z);
        p.inf = 0u;
        return;
    }
    E Z1Z1, U2, S2, H, HH, I, J, r, V, t, u;
    LF::template sqr<LZ_KZ>(Z1Z1, p.Z);         // < 4
    LF::template mul<4>(U2, x2, Z1Z1);          // < 2
    LF::template mul<4>(t, p.Z, Z1Z1);          // < 2
    LF::template mul<2>(S2, y2, t);             // < 2
    LF::template sub<LZ_KX>(H, U2, p.X);        // < 54
    LF::template sub<LZ_KY>(t, S2, p.Y);        // S2 - Y1  < 38
    if (LF::template is_zero<54>(H)) {          // same x: the point itself (double it) or its inverse (cancel)
        if (LF::template is_zero<38>(t)) {
            p.X = x2;
            p.Y = y2;
            LF::one(p.Z);
            jaclz_dbl(p);
        } else {
            jaclz_set_inf(p);
        }
        return;
    }
    LF::add(r, t, t);                           // < 76
    LF::template sqr<54>(HH, H);                // < 4
    LF::mul4(I, HH);                            // < 16
    LF::template mul<16>(J, H, I);              // < 2
    LF::template mul<16>(V, p.X, I);            // < 2
    LF::template sqr<76>(t, r);                 // r^2 < 4
    LF::add2x(u, J, V);                         // J + 2 V  < 6 (Fp2 < 6)
    LF::template sub<8>(t, t, u);               // X3 = r^2 - J - 2 V  < 12
    LF::template sub<12>(u, V, t);              // V - X3  < 14
    E nj;
    LF::add(nj, J, J);                          // 2 J < 4
    {
        E z;
        LF::template neg_if<4>(z, nj, true);    // 4p - 2 J  < 4
        nj = z;
    }
    E y3;
    LF::template mul2sum<14, 4>(y3, r, u, p.Y, nj);  // Y3 = r (V - X3) - 2 Y1 J  < 4
    LF::add(u, p.Z, H);                         // < 66
    LF::template sqr<66>(u, u);                 // < 4
    LF::add(V, Z1Z1, HH);                       // < 8
    LF::template sub<8>(p.Z, u, V);             // Z3 = (Z1 + H)^2 - Z1Z1 - HH  < 12
    p.X = t;
    p.Y = y3;
}
// r = k p for a public 64-bit k and an affine p, on the signed binary (NAF) form of k with mixed additions -- the lazy
// counterpart of curve.cuh jac_mul_u64_aff (the subgroup tests of the BN twists: [u] Q).  k is the same in every lane:
// the loop is uniform.
template <class LF, class F>
KYB_HD void jaclz_mul_u64_aff(Jac<F>& r, const Aff<F>& p, uint64_t k) {
    if (p.inf) {
        jac_set_inf(r);
        return;
    }
    uint64_t pos = 0, neg = 0;
    bool top = false;
    {
        unsigned __int128 x = k;
#pragma unroll 1
        for (int i = 0; x != 0; i++) {
            if (x & 1) {
                if ((x & 3) == 1) {
                    if (i < 64) pos |= uint64_t(1) << i;
                    else top = true;
                    x -= 1;
                } else {
                    neg |= uint64_t(1) << i;
                    x += 1;
                }
            }
            x >>= 1;
        }
    }
    typename LF::E x2, y2;
    LF::enter(x2, p.x);
    LF::enter(y2, p.y);
    JacLz<LF> acc;
    jaclz_set_inf(acc);
#pragma unroll 1
    for (int i = 64; i >= 0; i--) {
        if (!acc.inf) jaclz_dbl(acc);
        const bool dp = i == 64 ? top : ((pos >> i) & 1) != 0, dn = i < 64 && ((neg >> i) & 1) != 0;
        if (dp || dn) jaclz_madd(acc, x2, y2, dn);
    }
    jaclz_leave(r, acc);
}

// ---- the same two operations for a field's NATIVE limbs when they leave only R / p >= 2^9 of headroom (BLS12-381 Fp:
// thirteen 30-bit limbs, R / p = 630) -- no conversion into another Montgomery domain, 338 multiply-adds per product
// instead of the fourteen-limb form's 392.  The formulas above square sums (X + Y^2)^2, (Z + H)^2, r^2 whose operands
// reach 56p - 76p; here those three become products of small operands (4 X Y^2 = 4 X B, Z3 = 2 Z H, r^2 = 4 rr^2):
// a doubling is 3M + 4S, a mixed addition 8M + 3S (one of them two-product), every product below 400 < R / p.
// Between operations X < 18, Y < 18, Z < 4.
template <class C>
struct LzFpN {
    using E = FpL<C>;
    using Packed = Fp<C>;
    static_assert(fpl_supported<C>(), "native lazy limbs need R / p >= 2^9");
    KYB_HD static void enter(E& r, const Packed& a) { fpl_unpack(r, a); }
    KYB_HD static void leave(Packed& r, const E& a) {  // any lazy value: one multiplication by the Montgomery one brings it below 2p
        E o, t;
        fpl_one(o);
        fpl_mul(t, a, o);
        fpl_finish(r, t);
    }
    KYB_HD static void one(E& r) { fpl_one(r); }
    KYB_HD static void mul(E& r, const E& a, const E& b) { fpl_mul(r, a, b); }
    KYB_HD static void sqr(E& r, const E& a) { fpl_sqr(r, a); }
    template <int KB> KYB_HD static void mulb(E& r, const E& a, const E& b) { fpl_mul(r, a, b); }
    template <int KA> KYB_HD static void sqrb(E& r, const E& a) { fpl_sqr(r, a); }
    KYB_HD static void mul2sum(E& r, const E& a, const E& b, const E& c, const E& d) { fpl_mul2sum(r, a, b, c, d); }
    KYB_HD static void add(E& r, const E& a, const E& b) { fpl_add(r, a, b); }
    KYB_HD static void add2x(E& r, const E& a, const E& b) { fpl_add_2x(r, a, b); }
    KYB_HD static void mul4(E& r, const E& a) { fpl_mul4(r, a); }
    template <int K> KYB_HD static void sub(E& r, const E& a, const E& b) { fpl_sub<K>(r, a, b); }
    template <int KMAX> KYB_HD static bool is_zero(const E& a) { return fpl_is_zero_mod_p<KMAX>(a); }
    template <int K> KYB_HD static void neg_if(E& r, const E& a, bool neg) {  // neg ? K p - a : a
        E z, n;
#pragma unroll
        for (int j = 0; j < C::N; j++) z.l[j] = 0;
        KYB_LZ_K(z.k = 0.0;)
        fpl_sub<K>(n, z, a);
#pragma unroll
        for (int j = 0; j < C::N; j++) r.l[j] = neg ? n.l[j] : a.l[j];
        KYB_LZ_K(r.k = neg ? n.k : a.k;)
    }
};
// In: X < 18, Y < 18, Z < 12.  Out: X < 18, Y < 18, Z < 4.
template <class LF>
KYB_HD void jaclz_dbl_t(JacLz<LF>& p) {
    using E = typename LF::E;
    E A, B, C, D, Ee, G, t, u;
    LF::sqr(A, p.X);               // 324
    LF::sqr(B, p.Y);               // 324
    LF::sqr(C, B);                 // 4
    LF::mul(t, p.X, B);            // X Y^2: 36
    LF::mul4(D, t);                // 4 X Y^2  < 8
    LF::add2x(Ee, A, A);           // 3 X^2  < 6
    LF::sqr(G, Ee);                // 36
    LF::mul(t, p.Y, p.Z);          // 216
    LF::add(p.Z, t, t);            // < 4
    LF::add(u, D, D);              // < 16
    LF::template sub<16>(p.X, G, u);   // X3 = G - 2 D  < 18
    LF::template sub<18>(t, D, p.X);   // D - X3  < 26
    LF::mul(t, Ee, t);             // 156
    LF::mul4(u, C);                // < 8
    LF::add(u, u, u);              // 8 C < 16
    LF::template sub<16>(p.Y, t, u);   // Y3  < 18
}
// p += +-(x2, y2), coordinates below 2p.  In: X < 18, Y < 18, Z < 12.  Out: X < 14, Y < 4, Z < 4.
template <class LF>
KYB_HD void jaclz_madd_t(JacLz<LF>& p, const typename LF::E& x2, const typename LF::E& y2in, bool neg) {
    using E = typename LF::E;
    E y2;
    LF::template neg_if<2>(y2, y2in, neg);  // < 2 (+ the input's bound when not negated: < 2)
    if (p.inf) {
        p.X = x2;
        p.Y = y2;
        LF::one(p.Z);
        p.inf = 0u;
        return;
    }
    E Z1Z1, U2, S2, H, HH, I, J, rr, V, t, u;
    LF::sqr(Z1Z1, p.Z);                 // 144
    LF::mul(U2, x2, Z1Z1);              // 4
    LF::mul(t, p.Z, Z1Z1);              // 24
    LF::mul(S2, y2, t);                 // 4
    LF::template sub<18>(H, U2, p.X);   // < 20
    LF::template sub<18>(rr, S2, p.Y);  // S2 - Y1 = r / 2  < 20
    if (LF::template is_zero<20>(H)) {  // same x: the point itself (double it) or its inverse (cancel)
        if (LF::template is_zero<20>(rr)) {
            p.X = x2;
            p.Y = y2;
            LF::one(p.Z);
            jaclz_dbl_t(p);
        } else {
            jaclz_set_inf(p);
        }
        return;
    }
    LF::sqr(HH, H);                     // 400
    LF::mul4(I, HH);                    // < 8
    LF::mul(J, H, I);                   // 160
    LF::mul(V, p.X, I);                 // 144
    LF::sqr(t, rr);                     // 400
    LF::mul4(t, t);                     // r^2  < 8
    LF::add2x(u, J, V);                 // J + 2 V  < 6
    LF::template sub<6>(t, t, u);       // X3  < 14
    LF::template sub<14>(u, V, t);      // V - X3  < 16
    E nj, z;
    LF::template neg_if<2>(nj, J, true);  // 2p - J
    LF::mul2sum(z, rr, u, p.Y, nj);     // (r / 2)(V - X3) - Y1 J: 320 + 36
    LF::add(p.Y, z, z);                 // Y3  < 4
    LF::mul(z, p.Z, H);                 // 240
    LF::add(p.Z, z, z);                 // Z3 = 2 Z1 H  < 4
    p.X = t;
}
template <class LF, class F>
KYB_HD void jaclz_mul_u64_aff_t(Jac<F>& r, const Aff<F>& p, uint64_t k) {
    if (p.inf) {
        jac_set_inf(r);
        return;
    }
    uint64_t pos = 0, neg = 0;
    bool top = false;
    {
        unsigned __int128 x = k;
#pragma unroll 1
        for (int i = 0; x != 0; i++) {
            if (x & 1) {
                if ((x & 3) == 1) {
                    if (i < 64) pos |= uint64_t(1) << i;
                    else top = true;
                    x -= 1;
                } else {
                    neg |= uint64_t(1) << i;
                    x += 1;
                }
            }
            x >>= 1;
        }
    }
    typename LF::E x2, y2;
    LF::enter(x2, p.x);
    LF::enter(y2, p.y);
    JacLz<LF> acc;
    jaclz_set_inf(acc);
#pragma unroll 1
    for (int i = 64; i >= 0; i--) {
        if (!acc.inf) jaclz_dbl_t(acc);
        const bool dp = i == 64 ? top : ((pos >> i) & 1) != 0, dn = i < 64 && ((neg >> i) & 1) != 0;
        if (dp || dn) jaclz_madd_t(acc, x2, y2, dn);
    }
    jaclz_leave(r, acc);
}

// The window table of a ladder, built in the lazy form: tx[j], ty[j] = affine coordinates (below 2p per coefficient) of
// (j + 1) P for a finite affine P, bit j of infmask = that multiple is the point at infinity (a point of small order:
// only a wrongly vouched-for input gets there).  One doubling and six MIXED additions (every multiple is the previous
// one plus the affine P: the packed builder paid six full additions through out-of-line calls), the eight Z's inverted
// together (Montgomery's trick; the one inversion is the packed code's division-step routine).
template <class LF, bool TIGHT, class F>
KYB_HD void jaclz_table8(typename LF::E (&tx)[8], typename LF::E (&ty)[8], uint32_t& infmask, const F& px, const F& py) {
    using E = typename LF::E;
    E zs[8], c[8];
    LF::enter(tx[0], px);
    LF::enter(ty[0], py);
    LF::one(zs[0]);
    infmask = 0;
    JacLz<LF> acc;
    acc.X = tx[0];
    acc.Y = ty[0];
    LF::one(acc.Z);
    acc.inf = 0u;
#pragma unroll 1
    for (int j = 1; j < 8; j++) {
        if (j == 1) {
            if constexpr (TIGHT) jaclz_dbl_t(acc);
            else jaclz_dbl(acc);
        } else {
            if constexpr (TIGHT) jaclz_madd_t(acc, tx[0], ty[0], false);
            else jaclz_madd(acc, tx[0], ty[0], false);
        }
        tx[j] = acc.X;
        ty[j] = acc.Y;
        if (acc.inf) {
            infmask |= 1u << j;
            LF::one(zs[j]);  // a placeholder keeps the running product invertible
        } else {
            zs[j] = acc.Z;
        }
    }
    c[0] = zs[0];
#pragma unroll 1
    for (int j = 1; j < 8; j++) LF::template mulb<LZ_KZ>(c[j], c[j - 1], zs[j]);
    E inv;
    {
        F pk, pinv;
        LF::leave(pk, c[7]);
        f_inv(pinv, pk);
        LF::enter(inv, pinv);
    }
#pragma unroll 1
    for (int j = 7; j >= 1; j--) {
        E zi, zi2;
        LF::template mulb<2>(zi, inv, c[j - 1]);      // 1 / Z_j
        LF::template mulb<LZ_KZ>(inv, inv, zs[j]);    // 1 / (Z_0 ... Z_{j-1})
        LF::template sqrb<4>(zi2, zi);
        LF::template mulb<4>(tx[j], tx[j], zi2);
        LF::template mulb<4>(zi2, zi2, zi);
        LF::template mulb<4>(ty[j], ty[j], zi2);
    }
}

template <class LF, class F>
KYB_HD void jaclz_leave(Jac<F>& r, const JacLz<LF>& p) {
    if (p.inf) {
        jac_set_inf(r);
        return;
    }
    LF::leave(r.X, p.X);
    LF::leave(r.Y, p.Y);
    LF::leave(r.Z, p.Z);
    // The doublings never set the flag (no point of an odd-order curve has Y = 0) -- a precondition that a vouched-for
    // (KYB_F_TRUSTED) operand off the curve or of even order could break: Z = 0 (mod p) then arrives here with inf = 0.
    // The packed form reads Z = 0 as infinity; make the whole point the canonical (1, 1, 0) so that nothing downstream
    // sees the meaningless X, Y.
    if (f_is_zero(r.Z)) jac_set_inf(r);
}

}  // namespace kyb
