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

// s = a * b * R^-1 (mod p) on unpacked operands: normalised limbs out (below 2^W each), VALUE BELOW 2p -- not reduced
// further.  Operands are limbs below 2^W of values with a b < R p (see the lazily reduced sums above: anything
// below 2p qualifies), so the output of one call is a valid operand of the next: chains of multiplications and
// squarings (fp_pow_words) stay in this form and skip the pack / conditional subtraction / unpack between links.
template <class C>
KYB_HD void fp_mul_limbs(uint32_t (&s)[C::N], const uint32_t (&al)[C::N], const uint32_t (&bl)[C::N]) {
    constexpr int N = C::N, W = C::W;
    constexpr uint32_t MASK = (1u << W) - 1;
    // products of two W-bit limbs that fit a 64-bit column together with a carry-in
    constexpr int MAXP = (W >= 32) ? 0 : (int)((~0ull) / ((uint64_t)MASK * MASK)) - 1;
    static_assert(MAXP >= 4, "limb width too large for lazy column accumulation");
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
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
        t[j + 1] += t[j] >> W;
        s[j] = (uint32_t)t[j] & MASK;
    }
    s[N - 1] = (uint32_t)t[N - 1];
}
// r = a * b * R^-1 mod p
template <class C>
KYB_HD void fp_mul(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {
    uint32_t al[C::N], bl[C::N], s[C::N];
    fp_unpack<C>(al, a.v);
    fp_unpack<C>(bl, b.v);
    fp_mul_limbs<C>(s, al, bl);
    fp_finish<C>(r, s);
}
// r = a^2 * R^-1 mod p.  Same interleaved product/reduction walk as fp_mul, but row i only adds
// a_i^2 and the doubled cross products 2 a_i a_j (j > i): N(N+1)/2 + N^2 MADs instead of 2 N^2
// (BLS12-381: 260 vs 338).  `cnt` tracks, per window slot, how many 2^(2W)-sized products the
// column may hold (a doubled product counts twice); all of it folds at compile time after unrolling.
template <class C>
KYB_HD void fp_sqr_limbs(uint32_t (&s)[C::N], const uint32_t (&al)[C::N]) {  // limbs in / out as fp_mul_limbs
    constexpr int N = C::N, W = C::W;
    constexpr uint32_t MASK = (1u << W) - 1;
    constexpr int MAXP = (W >= 32) ? 0 : (int)((~0ull) / ((uint64_t)MASK * MASK)) - 1;
    static_assert(MAXP >= 6, "limb width too large for lazy column accumulation");
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
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
        t[j + 1] += t[j] >> W;
        s[j] = (uint32_t)t[j] & MASK;
    }
    s[N - 1] = (uint32_t)t[N - 1];
}
template <class C>
KYB_HD void fp_sqr(Fp<C>& r, const Fp<C>& a) {
    uint32_t al[C::N], s[C::N];
    fp_unpack<C>(al, a.v);
    fp_sqr_limbs<C>(s, al);
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
// The whole chain runs on unpacked limbs (fp_mul_limbs / fp_sqr_limbs: values below 2p, table entries included):
// one unpack at the start, one pack + conditional subtraction at the end instead of one of each per link -- 80 of a
// squaring's 512 instructions, 106 of a multiplication's 539 (BLS12-381).
template <class C>
KYB_HD_NOINLINE void fp_pow_words(Fp<C>& r, const Fp<C>& a, const uint32_t* e, int nbits) {
    constexpr int N = C::N;
    struct L {
        uint32_t l[N];
    };
    L tab[15];  // a^1 .. a^15
    fp_unpack<C>(tab[0].l, a.v);
#pragma unroll 1
    for (int j = 1; j < 15; j++) fp_mul_limbs<C>(tab[j].l, tab[j - 1].l, tab[0].l);
    L acc;
    bool one = true;  // the accumulator is still 1 (uniform: the exponent is public)
    const int top = (nbits + 3) / 4 - 1;
#pragma unroll 1
    for (int w = top; w >= 0; w--) {
        if (!one) {
            fp_sqr_limbs<C>(acc.l, acc.l);
            fp_sqr_limbs<C>(acc.l, acc.l);
            fp_sqr_limbs<C>(acc.l, acc.l);
            fp_sqr_limbs<C>(acc.l, acc.l);
        }
        const int bit = 4 * w;
        const uint32_t nib = (e[bit >> 5] >> (bit & 31)) & 15u;  // windows are nibble-aligned: never straddle a word
        if (nib) {
            if (one) {
                acc = tab[nib - 1];
                one = false;
            } else {
                fp_mul_limbs<C>(acc.l, acc.l, tab[nib - 1].l);
            }
        }
    }
    if (one) {
        fp_one(r);
        return;
    }
    fp_finish<C>(r, acc.l);
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

// Inversion, inv(0) = 0, by Bernstein-Yang division steps ("safegcd", eprint 2019/266) on signed 30-bit limbs: the pair
// (f, g) = (p, x) is driven to (+-1, 0) thirty division steps at a time -- the steps of a batch read only the low words
// and yield a 2 x 2 transition matrix with entries below 2^30 in magnitude, which is then applied to the full-width
// (f, g) (an exact division by 2^30) and, modulo p, to the cofactors (d, e) (f = d x, g = e x mod p).  A batch costs
// ~600 word operations for its steps (branch-free, so a wave stays converged) and ~130 multiply-adds for the two
// updates; ~13 (254 bits) to ~25 (381 bits) batches are needed, against ~230 instructions for each of Kaliski's
// 1.5 log2 p single-bit steps that this replaced (1.3e5 -> ~2e4 instructions for BLS12-381).  The loop ends when g = 0,
// so the time depends on the operand -- like everything else in the library.
//   in : a R mod p as packed words;  out: a^-1 R  (= plain inverse of the words, times R^2, by one multiplication by R^3)
template <class C>
struct Inv30 {
    static constexpr int NL = (C::PBITS + 2 + 29) / 30;  // limbs of 30 bits with room for the sign and one doubling
    static constexpr uint32_t M30 = (1u << 30) - 1;
    static constexpr uint32_t pinv() {  // p^-1 mod 2^32 by Newton iteration from p's low word
        uint32_t x = C::PW[0];
        for (int k = 0; k < 5; k++) x *= 2u - C::PW[0] * x;
        return x;
    }
};
template <class C>
KYB_HD void inv30_from_words(int32_t (&l)[Inv30<C>::NL], const uint32_t* w) {
    constexpr int NL = Inv30<C>::NL;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = 30 * i, idx = bit >> 5, sh = bit & 31;
        uint32_t x = idx < C::NWORDS ? (w[idx] >> sh) : 0u;
        if (sh > 2 && idx + 1 < C::NWORDS) x |= w[idx + 1] << (32 - sh);
        l[i] = (int32_t)(x & Inv30<C>::M30);
    }
}
template <class C>
KYB_HD_NOINLINE void fp_inv(Fp<C>& r, const Fp<C>& a) {
    using I = Inv30<C>;
    constexpr int NL = I::NL;
    constexpr uint32_t M30 = I::M30;
    constexpr uint32_t PINV = I::pinv();
    int32_t f[NL], g[NL], d[NL], e[NL], pl[NL];
    uint32_t pw[C::NWORDS];
#pragma unroll
    for (int i = 0; i < C::NWORDS; i++) pw[i] = C::PW[i];
    inv30_from_words<C>(pl, pw);
    inv30_from_words<C>(g, a.v);
#pragma unroll
    for (int i = 0; i < NL; i++) {
        f[i] = pl[i];
        d[i] = 0;
        e[i] = 0;
    }
    e[0] = 1;
    int32_t eta = -1;  // minus the delta of the paper
    uint32_t gnz = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) gnz |= (uint32_t)g[i];
#pragma unroll 1
    while (gnz) {
        // thirty division steps on the low words; matrix (u v; q r) with (f, g) <- (u f + v g, q f + r g) / 2^30
        uint32_t f0 = (uint32_t)f[0] | ((uint32_t)f[1] << 30), g0 = (uint32_t)g[0] | ((uint32_t)g[1] << 30);
        uint32_t u = 1, v = 0, q = 0, rr = 1;
#pragma unroll 6
        for (int k = 0; k < 30; k++) {
            const uint32_t c1 = (uint32_t)(eta >> 31);  // delta > 0
            const uint32_t c2 = 0u - (g0 & 1u);         // g odd
            const uint32_t x = (f0 ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // -f, -u, -v where delta > 0
            g0 += x & c2;
            q += y & c2;
            rr += z & c2;
            const uint32_t sw = c1 & c2;  // swap: f takes the old g
            eta = (int32_t)(((uint32_t)eta ^ sw) - 1u);
            f0 += g0 & sw;
            u += q & sw;
            v += rr & sw;
            g0 >>= 1;
            u <<= 1;
            v <<= 1;
        }
        const int32_t U = (int32_t)u, V = (int32_t)v, Q = (int32_t)q, R = (int32_t)rr;
        // (f, g): exact division by 2^30
        {
            int64_t cf = (int64_t)U * f[0] + (int64_t)V * g[0], cg = (int64_t)Q * f[0] + (int64_t)R * g[0];
            cf >>= 30;
            cg >>= 30;
#pragma unroll
            for (int i = 1; i < NL; i++) {
                cf += (int64_t)U * f[i] + (int64_t)V * g[i];
                cg += (int64_t)Q * f[i] + (int64_t)R * g[i];
                f[i - 1] = (int32_t)((uint32_t)cf & M30);
                g[i - 1] = (int32_t)((uint32_t)cg & M30);
                cf >>= 30;
                cg >>= 30;
            }
            f[NL - 1] = (int32_t)cf;
            g[NL - 1] = (int32_t)cg;
        }
        // (d, e) modulo p: a multiple of p makes the low 30 bits vanish (and p is added once more where d / e is
        // negative, which keeps both in (-2p, p))
        {
            const int32_t sd = d[NL - 1] >> 31, se = e[NL - 1] >> 31;
            int32_t md = (U & sd) + (V & se), me = (Q & sd) + (R & se);
            int64_t cd = (int64_t)U * d[0] + (int64_t)V * e[0], ce = (int64_t)Q * d[0] + (int64_t)R * e[0];
            md -= (int32_t)((PINV * (uint32_t)cd + (uint32_t)md) & M30);
            me -= (int32_t)((PINV * (uint32_t)ce + (uint32_t)me) & M30);
            cd += (int64_t)pl[0] * md;
            ce += (int64_t)pl[0] * me;
            cd >>= 30;
            ce >>= 30;
#pragma unroll
            for (int i = 1; i < NL; i++) {
                cd += (int64_t)U * d[i] + (int64_t)V * e[i] + (int64_t)pl[i] * md;
                ce += (int64_t)Q * d[i] + (int64_t)R * e[i] + (int64_t)pl[i] * me;
                d[i - 1] = (int32_t)((uint32_t)cd & M30);
                e[i - 1] = (int32_t)((uint32_t)ce & M30);
                cd >>= 30;
                ce >>= 30;
            }
            d[NL - 1] = (int32_t)cd;
            e[NL - 1] = (int32_t)ce;
        }
        gnz = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) gnz |= (uint32_t)g[i];
    }
    // f = +-1 (or +-p for x = 0, where d = 0): the inverse is sign(f) d, brought into [0, p)
    const int32_t sf = f[NL - 1] >> 31;  // -1 when f is negative
    {
        // d <- (d ^ sf) - sf, then + p while negative (at most twice), as limbs with a signed top
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            c += (int64_t)((d[i] ^ sf) - sf);
            d[i] = i + 1 < NL ? (int32_t)((uint32_t)c & M30) : (int32_t)c;
            if (i + 1 < NL) c >>= 30;
        }
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            const int32_t neg = d[NL - 1] >> 31;
            c = 0;
#pragma unroll
            for (int i = 0; i < NL; i++) {
                c += (int64_t)d[i] + (int64_t)(pl[i] & neg);
                d[i] = i + 1 < NL ? (int32_t)((uint32_t)c & M30) : (int32_t)c;
                if (i + 1 < NL) c >>= 30;
            }
        }
    }
    // limbs -> packed words (d in [0, p) now), then times R^3 / R = R^2
    Fp<C> y, r2, r3;
#pragma unroll
    for (int k = 0; k < C::NWORDS; k++) {
        const int bit = 32 * k, idx = bit / 30, sh = bit - 30 * idx;
        uint32_t x = (uint32_t)d[idx] >> sh;
        if (idx + 1 < NL) x |= (uint32_t)d[idx + 1] << (30 - sh);
        if (60 - sh < 32 && idx + 2 < NL) x |= (uint32_t)d[idx + 2] << (60 - sh);
        y.v[k] = x;
        r2.v[k] = C::R2[k];
    }
    fp_mul(r3, r2, r2);  // R^2 R^2 / R = R^3
    fp_mul(r, y, r3);    // x^-1 R^3 / R = (a R)^-1 R^2 = a^-1 R
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
