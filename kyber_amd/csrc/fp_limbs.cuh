// Base-field elements kept as UNSATURATED LIMBS between the multiplications of a formula -- the form the hot
// accumulation loops run in since round 5 (MSM bucket pieces, fixed-base table walks: runs of XYZZ mixed additions).
//
// mont.cuh keeps elements packed (12 words for BLS12-381) and fully reduced, and pays an unpack of both operands, a
// pack and a conditional subtraction around every multiplication: ~124 of a multiplication's ~539 instructions.  In a
// mixed addition (8M + 2S against 7 subtractions) that overhead is a fifth of the instruction count.  Here an element
// stays in the multiplier's own form -- N limbs of W bits, normalised (each below 2^W), VALUE only bounded by a small
// multiple of p ("lazy") -- so that the output of fp_mul_limbs (below 2p) is the next operand as it stands, and a
// subtraction is one signed carry sweep over the limbs with a multiple of p added in (never negative, no compare, no
// select).  Fields need headroom for that: R / p >= 2^9 (BLS12-381: 13 x 30 = 390 bits over a 381-bit prime); the BN
// fields (9 x 29 over 254 / 256 bits) keep the packed code.
//
// Bounds are per formula and stated where the formula is (curve.cuh xyzzl_madd); fp_mul_limbs needs Ka Kb < R / p for
// operands below Ka p, Kb p, and returns a value below 2p.
#pragma once
#include "mont.cuh"

namespace kyb {

// -DKYB_LZ_AUDIT (host harness builds only): every element carries an upper bound of its value as a multiple of p,
// every operation checks its precondition against it (products below R / p, subtrahends below the K p added in) and
// records a failure -- the lazy bounds of the formulas in curve.cuh / jac_lazy.cuh are then TESTED on real runs
// (tests/test_lazy_bounds.py), not only argued in comments.
#ifdef KYB_LZ_AUDIT
inline double& lz_audit_max_product() { static double v = 0; return v; }
inline int& lz_audit_failures() { static int v = 0; return v; }
#define KYB_LZ_K(x) x
#else
#define KYB_LZ_K(x)
#endif

template <class C>
struct FpL {
    uint32_t l[C::N];
#ifdef KYB_LZ_AUDIT
    double k = 1.0;  // value < k p
#endif
};
template <class C>
constexpr double fpl_headroom() {  // R / p from below: 2^(N W) / ((top word of p + 1) 2^(32 (NWORDS - 1)))
    double r = 1.0;
    for (int i = 0; i < C::N * C::W - 32 * (C::NWORDS - 1); i++) r *= 2.0;
    return r / ((double)C::PW[C::NWORDS - 1] + 1.0);
}
#ifdef KYB_LZ_AUDIT
template <class C>
inline void lz_check_product(double kk) {
    if (kk > lz_audit_max_product()) lz_audit_max_product() = kk;
    if (!(kk < fpl_headroom<C>())) lz_audit_failures()++;
}
#endif

template <class C>
constexpr bool fpl_supported() {
    return C::N * C::W - C::PBITS >= 9 && C::W <= 30 && C::N * C::W >= 32 * C::NWORDS;
}

// limbs of K p (compile-time), normalised; K p < 2^(N W)
template <class C, int K>
struct KTimesPL {
    struct Arr {
        uint32_t v[C::N];
    };
    static constexpr Arr make() {
        Arr r{};
        uint64_t c = 0;
        for (int j = 0; j < C::N; j++) {
            const uint64_t x = (uint64_t)C::P[j] * K + c;
            r.v[j] = j + 1 < C::N ? (uint32_t)(x & ((1u << C::W) - 1)) : (uint32_t)x;
            c = x >> C::W;
        }
        return r;
    }
    static constexpr Arr value = make();
};

template <class C> KYB_HD void fpl_unpack(FpL<C>& r, const Fp<C>& a) {
    fp_unpack<C>(r.l, a.v);
    KYB_LZ_K(r.k = 1.0;)
}
template <class C>
KYB_HD void fpl_one(FpL<C>& r) {
    Fp<C> o;
    fp_one(o);
    fp_unpack<C>(r.l, o.v);  // constant-folded
    KYB_LZ_K(r.k = 1.0;)
}
template <class C> KYB_HD void fpl_mul(FpL<C>& r, const FpL<C>& a, const FpL<C>& b) {
    KYB_LZ_K(lz_check_product<C>(a.k * b.k);)
    fp_mul_limbs<C>(r.l, a.l, b.l);
    KYB_LZ_K(r.k = 2.0;)
}
template <class C> KYB_HD void fpl_sqr(FpL<C>& r, const FpL<C>& a) {
    KYB_LZ_K(lz_check_product<C>(a.k * a.k);)
    fp_sqr_limbs<C>(r.l, a.l);
    KYB_LZ_K(r.k = 2.0;)
}
// value below 2p -> packed, fully reduced
template <class C>
KYB_HD void fpl_finish(Fp<C>& r, const FpL<C>& a) {
    uint32_t s[C::N];
    KYB_LZ_K(if (a.k > 2.0) lz_audit_failures()++;)
#pragma unroll
    for (int j = 0; j < C::N; j++) s[j] = a.l[j];
    fp_finish<C>(r, s);
}

// r = a - b + K p  (b below K p; the value is never negative, so the top limb ends non-negative): one signed sweep
template <int K, class C>
KYB_HD void fpl_sub(FpL<C>& r, const FpL<C>& a, const FpL<C>& b) {
    constexpr uint32_t MASK = (1u << C::W) - 1;
    KYB_LZ_K(if (b.k > K) lz_audit_failures()++; const double rk = a.k + K;)
    int32_t carry = 0;
#pragma unroll
    for (int j = 0; j < C::N; j++) {
        const int32_t t = (int32_t)(a.l[j] + KTimesPL<C, K>::value.v[j]) - (int32_t)b.l[j] + carry;
        r.l[j] = j + 1 < C::N ? ((uint32_t)t & MASK) : (uint32_t)t;
        carry = t >> C::W;
    }
    KYB_LZ_K(r.k = rk;)
}
// r = (neg ? -a : a) - b + K p  (a + b below K p)
template <int K, class C>
KYB_HD void fpl_sub_signed(FpL<C>& r, const FpL<C>& a, bool neg, const FpL<C>& b) {
    constexpr uint32_t MASK = (1u << C::W) - 1;
    const uint32_t m = neg ? ~0u : 0u;
    KYB_LZ_K(if (a.k + b.k > K) lz_audit_failures()++; const double rk = a.k + K;)
    int32_t carry = 0;
#pragma unroll
    for (int j = 0; j < C::N; j++) {
        const int32_t sa = (int32_t)((a.l[j] ^ m) - m);
        const int32_t t = (int32_t)KTimesPL<C, K>::value.v[j] - (int32_t)b.l[j] + sa + carry;
        r.l[j] = j + 1 < C::N ? ((uint32_t)t & MASK) : (uint32_t)t;
        carry = t >> C::W;
    }
    KYB_LZ_K(r.k = rk;)
}
// r = a + 2 b, normalised (one unsigned sweep)
template <class C>
KYB_HD void fpl_add_2x(FpL<C>& r, const FpL<C>& a, const FpL<C>& b) {
    constexpr uint32_t MASK = (1u << C::W) - 1;
    KYB_LZ_K(const double rk = a.k + 2 * b.k;)
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < C::N; j++) {
        const uint32_t t = a.l[j] + (b.l[j] << 1) + carry;  // < 2^W + 2^(W+1) + 4 <= 2^32 for W <= 30
        r.l[j] = j + 1 < C::N ? (t & MASK) : t;
        carry = t >> C::W;
    }
    KYB_LZ_K(r.k = rk;)
}
// r = a + b, normalised
template <class C>
KYB_HD void fpl_add(FpL<C>& r, const FpL<C>& a, const FpL<C>& b) {
    constexpr uint32_t MASK = (1u << C::W) - 1;
    KYB_LZ_K(const double rk = a.k + b.k;)
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < C::N; j++) {
        const uint32_t t = a.l[j] + b.l[j] + carry;
        r.l[j] = j + 1 < C::N ? (t & MASK) : t;
        carry = t >> C::W;
    }
    KYB_LZ_K(r.k = rk;)
}
// r = 4 a, normalised (limbs below 2^30: 4 a_j + carry stays below 2^32)
template <class C>
KYB_HD void fpl_mul4(FpL<C>& r, const FpL<C>& a) {
    static_assert(C::W <= 30, "4 a_j + carry must fit 32 bits");
    constexpr uint32_t MASK = (1u << C::W) - 1;
    KYB_LZ_K(const double rk = 4 * a.k;)
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < C::N; j++) {
        const uint32_t t = (a.l[j] << 2) + carry;
        r.l[j] = j + 1 < C::N ? (t & MASK) : t;
        carry = t >> C::W;
    }
    KYB_LZ_K(r.k = rk;)
}

// s = (a b + c d) R^-1 mod p with ONE reduction: the two products share their columns (3 N^2 multiply-adds instead of
// 4 N^2).  Operands below Ka p ... Kd p with Ka Kb + Kc Kd < R / p; value out below 2p.
template <class C>
KYB_HD void fpl_mul2sum(FpL<C>& r, const FpL<C>& a, const FpL<C>& b, const FpL<C>& c, const FpL<C>& d) {
    constexpr int N = C::N, W = C::W;
    constexpr uint32_t MASK = (1u << W) - 1;
    constexpr int MAXP = (int)((~0ull) / ((uint64_t)MASK * MASK)) - 1;
    static_assert(MAXP >= 6, "limb width too large for lazy column accumulation");
    uint64_t t[N];
    KYB_LZ_K(lz_check_product<C>(a.k * b.k + c.k * d.k);)
#pragma unroll
    for (int j = 0; j < N; j++) t[j] = 0;
    int pending = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (pending + 3 > MAXP) {
#pragma unroll
            for (int j = 0; j < N - 1; j++) {
                t[j + 1] += t[j] >> W;
                t[j] &= MASK;
            }
            pending = 0;
        }
        pending += 3;
        const uint32_t ai = a.l[i], ci = c.l[i];
#pragma unroll
        for (int j = 0; j < N; j++) t[j] += (uint64_t)ai * b.l[j];
#pragma unroll
        for (int j = 0; j < N; j++) t[j] += (uint64_t)ci * d.l[j];
        const uint32_t m = ((uint32_t)t[0] * C::NINV) & MASK;
#pragma unroll
        for (int j = 0; j < N; j++) t[j] += (uint64_t)m * C::P[j];
        const uint64_t carry = t[0] >> W;
#pragma unroll
        for (int j = 0; j < N - 1; j++) t[j] = t[j + 1];
        t[N - 1] = 0;
        t[0] += carry;
    }
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
        t[j + 1] += t[j] >> W;
        r.l[j] = (uint32_t)t[j] & MASK;
    }
    r.l[N - 1] = (uint32_t)t[N - 1];
    KYB_LZ_K(r.k = 2.0;)
}

// a == 0 (mod p) for a lazy value below (KMAX + 1) p, exactly.  a = k p forces limb 0 to be k p_0 mod 2^W, so
// k = a_0 p_0^-1 mod 2^W is the only candidate: two instructions decide all but (KMAX + 1) / 2^W of the non-zero
// operands; the survivors are compared with k p limb by limb.
template <int KMAX, class C>
KYB_HD bool fpl_is_zero_mod_p(const FpL<C>& a) {
    constexpr uint32_t MASK = (1u << C::W) - 1;
    constexpr uint32_t PINV0 = (0u - C::NINV) & MASK;  // p_0^-1 mod 2^W  (NINV = -p^-1)
    KYB_LZ_K(if (a.k > KMAX + 1) lz_audit_failures()++;)
    const uint32_t k = (a.l[0] * PINV0) & MASK;
    if (k > (uint32_t)KMAX) return false;
    uint64_t c = 0;
    uint32_t diff = 0;
#pragma unroll
    for (int j = 0; j < C::N; j++) {
        const uint64_t x = (uint64_t)C::P[j] * k + c;
        const uint32_t want = j + 1 < C::N ? ((uint32_t)x & MASK) : (uint32_t)x;
        diff |= want ^ a.l[j];
        c = x >> C::W;
    }
    return diff == 0;
}

}  // namespace kyb
