// Prime-field arithmetic in Montgomery form for the pairing curves (BLS12-381 Fp, bn256 Fp),
// one field element per lane, integer VALU only.
//
// Replaces: pairing/bn256 gfP + gfpMul (gfp.go:15, gfp_generic.go:158, gfp_amd64.s) and the Fp
// layer of the external BLS12-381 backends (kilic/bls12-381 fp.go etc., go.mod:6-8).
//
// Representation.  An element is its Montgomery residue (radix R = 2^(N*W)) as NWORDS *saturated* 32-bit words in
// VGPRs, always fully reduced to [0, p).  Additions, subtractions, selects and comparisons -- more than half of the
// instructions of the tower-heavy pairing kernels, and every VALU instruction costs a wave64 the same ~4 cycles --
// run on that packed form: an addition is one v_addc_co chain, one v_subb_co trial subtraction and a select,
// ~3 instructions per word.  Multiplication unpacks its operands to N *unsaturated* limbs of W bits (BLS12-381:
// 13 x 30, bn256: 9 x 29; two instructions per limb), because unsaturated limbs are what makes v_mad_u64_u32 the
// whole inner loop: a product of two limbs is < 2^(2W) <= 2^60, so a 64-bit column accumulator absorbs up to MAXP
// products with no carry instructions between them (saturated limbs would need a v_add_co/v_addc pair per product,
// +50% VALU issue).  The product is the interleaved (CIOS-order) product/reduction over a sliding window of N column
// accumulators; when 2N products per column would overflow 64 bits (BLS12-381: 26 x 2^60) one mid-way carry sweep
// renormalises the window; the result is reduced in limb form and packed back (two instructions per word).
// (The first version kept the unsaturated limbs everywhere: an Fp381 addition cost ~105 dependent instructions.)
#pragma once
#include "hd.h"

namespace kyb {

// 32-bit add / subtract with carry in and out (v_addc_co / v_subb_co chains under clang; gcc, which only ever builds
// the host test harness, takes the 64-bit form) and a mask select (v_bfi_b32).
KYB_HD uint32_t adc32(uint32_t a, uint32_t b, uint32_t& carry) {
#if defined(__clang__)
    unsigned co;
    const uint32_t r = __builtin_addc(a, b, carry, &co);
    carry = co;
    return r;
#else
    const uint64_t x = (uint64_t)a + b + carry;
    carry = (uint32_t)(x >> 32);
    return (uint32_t)x;
#endif
}
KYB_HD uint32_t sbb32(uint32_t a, uint32_t b, uint32_t& borrow) {
#if defined(__clang__)
    unsigned bo;
    const uint32_t r = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return r;
#else
    const uint64_t x = (uint64_t)a - b - borrow;
    borrow = (uint32_t)(x >> 63);
    return (uint32_t)x;
#endif
}
KYB_HD uint32_t sel32(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

// A field configuration C provides:
//   N, W                       limb count / limb width of the multiplier
//   P[N], PW[NWORDS]           modulus as limbs / as words
//   NINV                       -p^-1 mod 2^W
//   ONE[NWORDS], R2[NWORDS]    R mod p, R^2 mod p (words)
//   NWORDS                     32-bit words of an element (12 for 48 bytes, 8 for 32 bytes)
template <class C>
struct Fp {
    uint32_t v[C::NWORDS];
};

template <class C>
KYB_HD void fp_zero(Fp<C>& r) {
#pragma unroll
    for (int j = 0; j < C::NWORDS; j++) r.v[j] = 0;
}
template <class C>
KYB_HD void fp_one(Fp<C>& r) {
#pragma unroll
    for (int j = 0; j < C::NWORDS; j++) r.v[j] = C::ONE[j];
}
template <class C>
KYB_HD bool fp_is_zero(const Fp<C>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < C::NWORDS; j++) o |= a.v[j];
    return o == 0;
}
template <class C>
KYB_HD bool fp_eq(const Fp<C>& a, const Fp<C>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < C::NWORDS; j++) o |= a.v[j] ^ b.v[j];
    return o == 0;
}
// r = c ? a : r
template <class C>
KYB_HD void fp_cmov(Fp<C>& r, const Fp<C>& a, bool c) {
#pragma unroll
    for (int j = 0; j < C::NWORDS; j++) r.v[j] = c ? a.v[j] : r.v[j];
}

// words <-> limbs: limb j = bits [W j, W j + W)
template <class C>
KYB_HD void fp_unpack(uint32_t (&l)[C::N], const uint32_t (&w)[C::NWORDS]) {
    constexpr uint32_t MASK = (1u << C::W) - 1;
#pragma unroll
    for (int j = 0; j < C::N; j++) {
        const int bit = j * C::W, idx = bit >> 5, sh = bit & 31;
        uint32_t x = idx < C::NWORDS ? (w[idx] >> sh) : 0u;
        if (sh + C::W > 32 && idx + 1 < C::NWORDS) x |= w[idx + 1] << (32 - sh);
        l[j] = x & MASK;
    }
}
// limbs normalised (< 2^W each) and the value below 2^(32 NWORDS)
template <class C>
KYB_HD void fp_pack(uint32_t (&w)[C::NWORDS], const uint32_t (&l)[C::N]) {
#pragma unroll
    for (int k = 0; k < C::NWORDS; k++) {
        const int bit = 32 * k, j = bit / C::W, o = bit - j * C::W;
        uint32_t x = l[j] >> o;
        if (j + 1 < C::N) x |= l[j + 1] << (C::W - o);
        if (2 * C::W - o < 32 && j + 2 < C::N) x |= l[j + 2] << (2 * C::W - o);
        w[k] = x;
    }
}

template <class C>
KYB_HD void fp_add(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
    constexpr int NW = C::NWORDS;
    uint32_t s[NW], d[NW];
    uint32_t carry = 0, borrow = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) s[j] = adc32(a.v[j], b.v[j], carry);
#pragma unroll
    for (int j = 0; j < NW; j++) d[j] = sbb32(s[j], C::PW[j], borrow);
    // a + b >= p  <=>  the sum overflowed the words (bn256: 2p > 2^256) or the trial subtraction did not borrow
    const bool ge = carry | (borrow ^ 1u);
#pragma unroll
    for (int j = 0; j < NW; j++) r.v[j] = ge ? d[j] : s[j];
}
template <class C>
KYB_HD void fp_dbl(Fp<C>& r, const Fp<C>& a) {
    fp_add(r, a, a);
}

template <class C>
KYB_HD void fp_sub(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
    constexpr int NW = C::NWORDS;
    uint32_t d[NW];
    uint32_t borrow = 0, carry = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) d[j] = sbb32(a.v[j], b.v[j], borrow);
    const uint32_t m = 0u - borrow;  // all ones when a < b: add p back (mod 2^(32 NW))
#pragma unroll
    for (int j = 0; j < NW; j++) r.v[j] = adc32(d[j], C::PW[j] & m, carry);
}

// ---- lazily reduced sums, for operands of a multiplication only --------------------------------------------------
// fp_mul computes a b R^-1 mod p correctly (result < 2p before its final conditional subtraction) for any operands
// with a b < R p, i.e. values below Ka p and Kb p with Ka Kb < R / p (BLS12-381: 2^9, bn256: 2^5.8).  A Karatsuba
// pre-addition (a0 + a1) therefore does not need fp_add's trial subtraction and select (two thirds of its
// instructions): the carry chain alone leaves a valid operand.  The tower code (tower.cuh) uses these for
// temporaries whose only use is as a multiplication operand, at nesting depths that keep the product of the bounds
// below R / p; everything that leaves a tower function is fully reduced as before.
// In the packed form a lazy sum must also fit the words: fields whose modulus leaves three spare bits
// (BLS12-381: 381 of 384) keep sums below 8p; bn256's modulus fills its 256 bits, so its "lazy" operations are the
// exact ones (which cost about the same there).
template <class C>
constexpr bool fp_has_headroom() {
    return C::PBITS + 3 <= 32 * C::NWORDS;
}
// r = a + b, value < (Ka + Kb) p <= 8p
template <class C>
KYB_HD void fp_add_nr(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
    if constexpr (fp_has_headroom<C>()) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < C::NWORDS; j++) r.v[j] = adc32(a.v[j], b.v[j], carry);
    } else {
        fp_add(r, a, b);
    }
}
// words of K p (compile-time)
template <class C, int K>
struct KTimesP {
    struct Arr {
        uint32_t v[C::NWORDS];
    };
    static constexpr Arr make() {
        Arr r{};
        uint64_t c = 0;
        for (int j = 0; j < C::NWORDS; j++) {
            const uint64_t x = (uint64_t)C::PW[j] * K + c;
            r.v[j] = (uint32_t)x;
            c = x >> 32;
        }
        return r;
    }
    static constexpr Arr value = make();
};
// r = a - b + K p for b < K p: value in [0, (Ka + K) p)
template <int K, class C>
KYB_HD void fp_sub_nr(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
    if constexpr (fp_has_headroom<C>()) {
        uint32_t t[C::NWORDS];
        uint32_t carry = 0, borrow = 0;
#pragma unroll
        for (int j = 0; j < C::NWORDS; j++) t[j] = adc32(a.v[j], KTimesP<C, K>::value.v[j], carry);
#pragma unroll
        for (int j = 0; j < C::NWORDS; j++) r.v[j] = sbb32(t[j], b.v[j], borrow);
    } else {
        fp_sub(r, a, b);
    }
}
template <class C>
KYB_HD void fp_neg(Fp<C>& r, const Fp<C>& a) {
    Fp<C> z;
    fp_zero(z);
    fp_sub(r, z, a);
}

// Tail of a multiplication: normalised limbs s (value < 2p) -> the packed, fully reduced element.  The value is packed
// first and the conditional subtraction runs on the words with a borrow chain (NWORDS subtract-with-borrow + NWORDS
// selects) instead of on the limbs (subtract, shift, mask per limb, then the selects).
template <class C>
KYB_HD void fp_finish(Fp<C>& r, uint32_t (&s)[C::N]) {
    if constexpr (fp_has_headroom<C>()) {
        uint32_t w[C::NWORDS], d[C::NWORDS];
        fp_pack<C>(w, s);
        uint32_t borrow = 0;
#pragma unroll
        for (int j = 0; j < C::NWORDS; j++) d[j] = sbb32(w[j], C::PW[j], borrow);
        const uint32_t keep = 0u - borrow;  // all ones when the value was already below p
#pragma unroll
        for (int j = 0; j < C::NWORDS; j++) r.v[j] = sel32(keep, w[j], d[j]);
    } else {
        // 2p exceeds the words by at most one bit (bn256: 2p < 2^257): that bit comes out of the top limb, the low
        // words take the same borrow chain, and the value is below p exactly when the bit is clear and the chain borrowed
        static_assert(C::N * C::W > 32 * C::NWORDS, "top limb must reach past the words");
        uint32_t w[C::NWORDS], d[C::NWORDS];
        fp_pack<C>(w, s);
        const uint32_t top = s[C::N - 1] >> (32 * C::NWORDS - (C::N - 1) * C::W);  // 0 or 1
        uint32_t borrow = 0;
#pragma unroll
        for (int j = 0; j < C::NWORDS; j++) d[j] = sbb32(w[j], C::PW[j], borrow);
        const uint32_t keep = (0u - borrow) & (top - 1u);
#pragma unroll
        for (int j = 0; j < C::NWORDS; j++) r.v[j] = sel32(keep, w[j], d[j]);
    }
}

// r = a * b * R^-1 mod p
template <class C>
KYB_HD void fp_mul(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
    constexpr int N = C::N, W = C::W;
    constexpr uint32_t MASK = (1u << W) - 1;
    // products of two W-bit limbs that fit a 64-bit column together with a carry-in
    constexpr int MAXP = (W >= 32) ? 0 : (int)((~0ull) / ((uint64_t)MASK * MASK)) - 1;
    static_assert(MAXP >= 4, "limb width too large for lazy column accumulation");
    uint32_t al[N], bl[N];
    fp_unpack<C>(al, a.v);
    fp_unpack<C>(bl, b.v);
    uint64_t t[N];
#pragma unroll
    for (int j = 0; j < N; j++) t[j] = 0;
    int pending = 0;  // products accumulated in the fullest column since the last sweep
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (pending + 2 > MAXP) {
#pragma unroll
            for (int j = 0; j < N - 1; j++) {
                t[j + 1] += t[j] >> W;
                t[j] &= MASK;
            }
            pending = 0;
        }
        pending += 2;
        const uint32_t ai = al[i];
#pragma unroll
        for (int j = 0; j < N; j++) t[j] += (uint64_t)ai * bl[j];
        const uint32_t m = ((uint32_t)t[0] * C::NINV) & MASK;
#pragma unroll
        for (int j = 0; j < N; j++) t[j] += (uint64_t)m * C::P[j];
        const uint64_t carry = t[0] >> W;  // low W bits are zero by construction
#pragma unroll
        for (int j = 0; j < N - 1; j++) t[j] = t[j + 1];
        t[N - 1] = 0;
        t[0] += carry;
    }
    uint32_t s[N];
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
        t[j + 1] += t[j] >> W;
        s[j] = (uint32_t)t[j] & MASK;
    }
    s[N - 1] = (uint32_t)t[N - 1];
    fp_finish<C>(r, s);
}
// r = a^2 * R^-1 mod p.  Same interleaved product/reduction walk as fp_mul, but row i only adds
// a_i^2 and the doubled cross products 2 a_i a_j (j > i): N(N+1)/2 + N^2 MADs instead of 2 N^2
// (BLS12-381: 260 vs 338).  `cnt` tracks, per window slot, how many 2^(2W)-sized products the
// column may hold (a doubled product counts twice); all of it folds at compile time after unrolling.
template <class C>
KYB_HD void fp_sqr(Fp<C>& r, const Fp<C>& a) {
    constexpr int N = C::N, W = C::W;
    constexpr uint32_t MASK = (1u << W) - 1;
    constexpr int MAXP = (W >= 32) ? 0 : (int)((~0ull) / ((uint64_t)MASK * MASK)) - 1;
    static_assert(MAXP >= 6, "limb width too large for lazy column accumulation");
    uint32_t al[N];
    fp_unpack<C>(al, a.v);
    uint64_t t[N];
    int cnt[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        t[j] = 0;
        cnt[j] = 0;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        bool need = false;
#pragma unroll
        for (int j = 0; j < N; j++) need |= cnt[j] + (j == i ? 2 : (j > i ? 3 : 1)) > MAXP;
        if (need) {
#pragma unroll
            for (int j = 0; j < N - 1; j++) {
                t[j + 1] += t[j] >> W;
                t[j] &= MASK;
            }
#pragma unroll
            for (int j = 0; j < N; j++) cnt[j] = 1;
        }
        const uint32_t ai = al[i], ai2 = ai << 1;
        t[i] += (uint64_t)ai * ai;
#pragma unroll
        for (int j = i + 1; j < N; j++) t[j] += (uint64_t)ai2 * al[j];
        const uint32_t m = ((uint32_t)t[0] * C::NINV) & MASK;
#pragma unroll
        for (int j = 0; j < N; j++) t[j] += (uint64_t)m * C::P[j];
#pragma unroll
        for (int j = 0; j < N; j++) cnt[j] += (j == i ? 2 : (j > i ? 3 : 1));
        const uint64_t carry = t[0] >> W;
        // slide the window one column up; slot indices of the not-yet-added rows shift with it, so the
        // products of row i' > i still land on slot (column - base): re-index by rotating t
#pragma unroll
        for (int j = 0; j < N - 1; j++) {
            t[j] = t[j + 1];
            cnt[j] = cnt[j + 1];
        }
        t[N - 1] = 0;
        cnt[N - 1] = 0;
        t[0] += carry;
        cnt[0] += 1;
    }
    uint32_t s[N];
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
        t[j + 1] += t[j] >> W;
        s[j] = (uint32_t)t[j] & MASK;
    }
    s[N - 1] = (uint32_t)t[N - 1];
    fp_finish<C>(r, s);
}

// Small-constant multiples
template <class C>
KYB_HD void fp_mul3(Fp<C>& r, const Fp<C>& a) {
    Fp<C> t;
    fp_add(t, a, a);
    fp_add(r, t, a);
}

// r = a^e for a public exponent held as NW little-endian 32-bit words.  Fixed 4-bit windows: nbits squarings +
// nbits/4 + 14 multiplications (the square roots' exponents have about half their bits set: ~nbits/2 with plain
// square-and-multiply).  The exponent is the same in every lane, so the table index and the branches are uniform.
template <class C>
KYB_HD_NOINLINE void fp_pow_words(Fp<C>& r, const Fp<C>& a, const uint32_t* e, int nbits) {
    Fp<C> tab[15];  // a^1 .. a^15
    tab[0] = a;
#pragma unroll 1
    for (int j = 1; j < 15; j++) fp_mul(tab[j], tab[j - 1], a);
    Fp<C> acc;
    fp_one(acc);
    const int top = (nbits + 3) / 4 - 1;
#pragma unroll 1
    for (int w = top; w >= 0; w--) {
        if (w != top) {
            fp_sqr(acc, acc);
            fp_sqr(acc, acc);
            fp_sqr(acc, acc);
            fp_sqr(acc, acc);
        }
        const int bit = 4 * w;
        const uint32_t nib = (e[bit >> 5] >> (bit & 31)) & 15u;  // windows are nibble-aligned: never straddle a word
        if (nib) fp_mul(acc, acc, tab[nib - 1]);
    }
    r = acc;
}
template <class C>
KYB_HD void fp_inv_fermat(Fp<C>& r, const Fp<C>& a) {  // a^(p-2); inv(0) = 0.  (fp_inv below is ~4x cheaper.)
    fp_pow_words<C>(r, a, C::PM2, C::PBITS);
}

// ------------------------------------------------------------ plain integers <-> elements
// w: NWORDS little-endian 32-bit words of a plain integer.  True when it is < p.
template <class C>
KYB_HD bool fp_words_lt_p(const uint32_t (&w)[C::NWORDS]) {
    uint32_t borrow = 0;
#pragma unroll
    for (int j = 0; j < C::NWORDS; j++) (void)sbb32(w[j], C::PW[j], borrow);
    return borrow != 0;
}
// plain integer words (must be < p) -> Montgomery element
template <class C>
KYB_HD_NOINLINE void fp_from_words(Fp<C>& r, const uint32_t (&w)[C::NWORDS]) {
    Fp<C> raw, r2;
#pragma unroll
    for (int j = 0; j < C::NWORDS; j++) {
        raw.v[j] = w[j];
        r2.v[j] = C::R2[j];
    }
    fp_mul(r, raw, r2);
}
// Montgomery element -> canonical plain integer words
template <class C>
KYB_HD_NOINLINE void fp_to_words(uint32_t (&w)[C::NWORDS], const Fp<C>& a) {
    Fp<C> one_raw, c;
    fp_zero(one_raw);
    one_raw.v[0] = 1;
    fp_mul(c, a, one_raw);
#pragma unroll
    for (int j = 0; j < C::NWORDS; j++) w[j] = c.v[j];
}

// Inversion, inv(0) = 0: Kaliski's almost-Montgomery inverse on 32-bit words.  Phase 1 (shifts, additions and
// subtractions only -- about 14 word operations per word per step, <= 2 * PBITS steps) turns x = a R into
// y = x^-1 2^k mod p; phase 2 multiplies by 2^(2 log2 R - k) with four Montgomery multiplications by R^2 and
// one-hot words, leaving a^-1 R.  That is roughly 1.2e5 instructions against 4.6e5 for the Fermat
// exponentiation (PBITS squarings + ~PBITS/2 multiplications of ~800 instructions each).  The step is written
// branch-free so a wave stays converged; lanes only differ in the step count k (PBITS <= k <= 2 PBITS).
// Variable time in the operand, like the reference's BLS12-381 backend (kilic fe.inverse is the same family).
template <class C>
KYB_HD_NOINLINE void fp_inv(Fp<C>& r, const Fp<C>& a) {
    constexpr int NW = C::NWORDS, NX = NW + 1;  // r, s < 2p need one more word when p fills its words (bn256)
    uint32_t u[NW], v[NW], rr[NX], ss[NX], pw[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) {
        pw[i] = C::PW[i];
        v[i] = a.v[i];
        u[i] = pw[i];
    }
#pragma unroll
    for (int i = 0; i < NX; i++) rr[i] = ss[i] = 0;
    ss[0] = 1;
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) nz |= v[i];
    const bool zero_in = nz == 0;
    int k = 0;
    while (nz) {
        // d = v - u, e = u - v (borrow chains), sum = r + s
        uint32_t d[NW], e[NW], sum[NX];
        uint32_t bd = 0, be = 0, cy = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            d[i] = sbb32(v[i], u[i], bd);
            e[i] = sbb32(u[i], v[i], be);
        }
#pragma unroll
        for (int i = 0; i < NX; i++) sum[i] = adc32(rr[i], ss[i], cy);
        // A: u even          -> u >>= 1,            s <<= 1
        // B: v even          -> v >>= 1,            r <<= 1
        // C: both odd, u > v -> u = (u - v) >> 1,   r += s, s <<= 1
        // D: both odd, else  -> v = (v - u) >> 1,   s += r, r <<= 1
        // as lane masks (all ones / zero), so the body is selects (v_bfi_b32), not branches
        const uint32_t uo = 0u - (u[0] & 1u), vo = 0u - (v[0] & 1u), lt = 0u - bd;  // lt: v < u
        const uint32_t mC = uo & vo & lt, mD = uo & vo & ~lt;
        const uint32_t mSU = ~uo | mC, mSV = (uo & ~vo) | mD;  // shift u (A | C), shift v (B | D)
#pragma unroll
        for (int i = 0; i < NW; i++) {
            u[i] = sel32(mC, e[i], u[i]);
            v[i] = sel32(mD, d[i], v[i]);
        }
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const uint32_t hu = i + 1 < NW ? u[i + 1] : 0u, hv = i + 1 < NW ? v[i + 1] : 0u;
            u[i] = sel32(mSU, (u[i] >> 1) | (hu << 31), u[i]);
            v[i] = sel32(mSV, (v[i] >> 1) | (hv << 31), v[i]);
        }
#pragma unroll
        for (int i = NX - 1; i >= 0; i--) {
            const uint32_t lr = i ? rr[i - 1] : 0u, ls = i ? ss[i - 1] : 0u;
            const uint32_t r2 = (rr[i] << 1) | (lr >> 31), s2 = (ss[i] << 1) | (ls >> 31);
            rr[i] = sel32(mSV, r2, sel32(mC, sum[i], rr[i]));  // B, D: 2r ; C: r + s ; A: r
            ss[i] = sel32(mSU, s2, sel32(mD, sum[i], ss[i]));  // A, C: 2s ; D: r + s ; B: s
        }
        nz = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) nz |= v[i];
        k++;
    }
    // r < 2p: reduce, then y = p - r = x^-1 2^k mod p
    uint32_t t[NW], tx[NX];
    uint32_t b = 0;
#pragma unroll
    for (int i = 0; i < NX; i++) tx[i] = sbb32(rr[i], i < NW ? pw[i] : 0u, b);
    const uint32_t keep = 0u - b;
#pragma unroll
    for (int i = 0; i < NW; i++) rr[i] = sel32(keep, rr[i], tx[i]);
    b = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) t[i] = sbb32(pw[i], rr[i], b);
    Fp<C> y, r2, pw2;
#pragma unroll
    for (int j = 0; j < NW; j++) {
        y.v[j] = t[j];
        r2.v[j] = C::R2[j];
    }
    // y 2^e with e = 2 * (N * W) - k split in two one-hot multiplications (each exponent < PBITS)
    const int e = 2 * C::N * C::W - k, e1 = e >> 1, e2 = e - e1;
    fp_mul(y, y, r2);  // y R
#pragma unroll
    for (int j = 0; j < NW; j++) pw2.v[j] = ((e1 >> 5) == j) ? (1u << (e1 & 31)) : 0u;
    fp_mul(y, y, pw2);  // y 2^e1
    fp_mul(y, y, r2);   // y 2^e1 R
#pragma unroll
    for (int j = 0; j < NW; j++) pw2.v[j] = ((e2 >> 5) == j) ? (1u << (e2 & 31)) : 0u;
    fp_mul(y, y, pw2);  // y 2^(e1 + e2) = x^-1 R^2 = a^-1 R
    fp_zero(r2);
    fp_cmov(y, r2, zero_in);
    r = y;
}

// Montgomery element from a small unsigned constant
template <class C>
KYB_HD void fp_from_u32(Fp<C>& r, uint32_t x) {
    uint32_t w[C::NWORDS];
#pragma unroll
    for (int k = 0; k < C::NWORDS; k++) w[k] = 0;
    w[0] = x;
    fp_from_words<C>(r, w);
}

// Big-endian byte strings <-> words.  Buffers are 4-byte aligned (element sizes are multiples of 16).
KYB_HD uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
template <int NW>
KYB_HD void words_from_be(uint32_t (&w)[NW], const uint8_t* p) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
    for (int k = 0; k < NW; k++) w[k] = bswap32(q[NW - 1 - k]);
}
template <int NW>
KYB_HD void words_to_be(uint8_t* p, const uint32_t (&w)[NW]) {
    uint32_t* q = reinterpret_cast<uint32_t*>(p);
#pragma unroll
    for (int k = 0; k < NW; k++) q[NW - 1 - k] = bswap32(w[k]);
}
// a > b on plain little-endian word arrays
template <int NW>
KYB_HD bool words_gt(const uint32_t (&a)[NW], const uint32_t (&b)[NW]) {
    bool gt = false, decided = false;
#pragma unroll
    for (int k = NW - 1; k >= 0; k--) {
        if (!decided && a[k] != b[k]) {
            gt = a[k] > b[k];
            decided = true;
        }
    }
    return gt;
}

}  // namespace kyb
