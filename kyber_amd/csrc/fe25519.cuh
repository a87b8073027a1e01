// GF(2^255-19) device arithmetic for gfx950 (one field element per lane).
//
// Replaces the reference's group/edwards25519/fe.go (feMul fe.go:348, feSquare
// fe.go:590, feSquare2 fe.go:749, feInvert fe.go:906, fePow22523 fe.go:961,
// feToBytes fe.go:163, feFromBytes fe.go:81).  Representation: ten signed
// limbs in radix 2^25.5 (26,25,26,25,... bits) held in VGPRs; every product is
// one v_mad_i64_i32 into a 64-bit column accumulator, so a multiplication is
// 100 integer MADs + 9 small multiplies + one carry sweep and needs no carry
// handling inside the product loop.  Bounds follow the classic analysis for
// this radix: mul/sq accept |limb| <= 1.65*2^26 (even) / 1.65*2^25 (odd) and
// return |limb| <= 1.01*2^25 / 1.01*2^24, so one add/sub of two products may
// feed the next product without a carry.
#pragma once
#include "hd.h"

namespace kyb {

struct fe {
    int32_t v[10];
#if defined(KYB_FE_AUDIT)
    // Host-only bound audit (tests/test_fe_bounds.py): `mag` bounds the limbs in units of 2^25 (even) / 2^24 (odd).
    // Carried results have 1.01, canonical constants and decoded values 2 (limbs below 2^26 / 2^25), sums add up.
    double mag = 2.0;
#endif
};

#if defined(KYB_FE_AUDIT)
// A 64-bit column of a product is at most 124.5 E_f E_g for even limbs below E (odd below E / 2): column 0 collects
// a0 b0 + 19 (4 even x even + 5 doubled odd x odd terms).  With E = mag * 2^25 that stays below 2^63 while
// mag_f * mag_g < 2^13 / 124.5 = 65.8; the audit records the largest product it meets (doubled for 2 f^2).
inline double& fe_audit_max() {
    static double m = 0;
    return m;
}
// The operand whose limbs are pre-multiplied by 19 / 38 in 32 bits (g of fe_mul, f of a squaring) must keep
// 19 * mag * 2^25 below 2^31: mag < 3.36 -- the constraint that makes the reference round its carries to signed limbs.
inline double& fe_audit_max19() {
    static double m = 0;
    return m;
}
inline void fe_audit_note(double pf, double pg, double scale) {
    const double x = pf * pg * scale;
    if (x > fe_audit_max()) fe_audit_max() = x;
    if (pg > fe_audit_max19()) fe_audit_max19() = pg;
}
#define KYB_FE_MAG_SET(h, x) ((h).mag = (x))
#define KYB_FE_MAG(f) ((f).mag)
#define KYB_FE_NOTE(pf, pg, scale) fe_audit_note(pf, pg, scale)
#else
#define KYB_FE_MAG_SET(h, x) ((void)0)
#define KYB_FE_MAG(f) 0.0
#define KYB_FE_NOTE(pf, pg, scale) ((void)0)
#endif

#if defined(__HIPCC__)
#define KYB_DEV __device__ __forceinline__
#else
#define KYB_DEV inline  // host build: tests/host_harness.cpp only
#endif

KYB_DEV void fe_0(fe& h) {
#pragma unroll
    for (int i = 0; i < 10; i++) h.v[i] = 0;
    KYB_FE_MAG_SET(h, 0.0);
}
KYB_DEV void fe_1(fe& h) {
    fe_0(h);
    h.v[0] = 1;
    KYB_FE_MAG_SET(h, 1.0 / (1 << 25));
}
KYB_DEV void fe_add(fe& h, const fe& f, const fe& g) {
    const double m_ = KYB_FE_MAG(f) + KYB_FE_MAG(g);
#pragma unroll
    for (int i = 0; i < 10; i++) h.v[i] = f.v[i] + g.v[i];
    KYB_FE_MAG_SET(h, m_);
}
KYB_DEV void fe_sub(fe& h, const fe& f, const fe& g) {
    const double m_ = KYB_FE_MAG(f) + KYB_FE_MAG(g);
#pragma unroll
    for (int i = 0; i < 10; i++) h.v[i] = f.v[i] - g.v[i];
    KYB_FE_MAG_SET(h, m_);
}
KYB_DEV void fe_neg(fe& h, const fe& f) {
    const double m_ = KYB_FE_MAG(f);
#pragma unroll
    for (int i = 0; i < 10; i++) h.v[i] = -f.v[i];
    KYB_FE_MAG_SET(h, m_);
}
// f = b ? g : f
KYB_DEV void fe_cmov(fe& f, const fe& g, bool b) {
#pragma unroll
    for (int i = 0; i < 10; i++) f.v[i] = b ? g.v[i] : f.v[i];
    KYB_FE_MAG_SET(f, KYB_FE_MAG(f) > KYB_FE_MAG(g) ? KYB_FE_MAG(f) : KYB_FE_MAG(g));  // either operand, whatever b
}
KYB_DEV void fe_cswap(fe& f, fe& g, bool b) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        int32_t x = f.v[i], y = g.v[i];
        f.v[i] = b ? y : x;
        g.v[i] = b ? x : y;
    }
    const double m_ = KYB_FE_MAG(f) > KYB_FE_MAG(g) ? KYB_FE_MAG(f) : KYB_FE_MAG(g);
    KYB_FE_MAG_SET(f, m_);
    KYB_FE_MAG_SET(g, m_);
    (void)m_;
}

// Column accumulators carry a BIAS of half a limb into the carry chain: T[i] = t[i] + 2^(w_i - 1), w_i = 26 (even i)
// / 25 (odd i).  The rounded carry of the reference (fe.go:348 ff: c = (t + 2^(w-1)) >> w, t -= c << w) is then one
// shift, and the remainder is the low w bits minus the bias -- two 32-bit operations instead of a 64-bit add and a
// shift/subtract.
// Only the even columns are handed their bias explicitly, and they carry the next (odd) column's with them:
// 2^25 + (2^24 << 26).  The second term is a multiple of 2^26, so it leaves the remainder alone and arrives in the odd
// column as exactly 2^24 on top of the carry -- one 64-bit add less per pair of columns.
KYB_DEV constexpr int64_t fe_bias(int i) { return (i & 1) ? 0 : ((int64_t)1 << 25) + ((int64_t)1 << 50); }

// The compiler reassociates bias + a0 b0 + a1 b1 + ... so that the constant (or a carry that arrives late) is added
// last, as a separate 64-bit add, and splits a long column into two partial sums.  An empty asm after a product pins
// the running sum as one value -- one v_mad_i64_i32 with the previous sum as its addend -- and emits nothing itself,
// so it cannot change a result.  fe_chain_store below says what that is worth.
#if defined(__HIP_DEVICE_COMPILE__)
#define KYB_PIN64(x) asm("" : "+v"(x))
#else
#define KYB_PIN64(x) ((void)0)
#endif
#define KYB_LOW(x, w) ((uint32_t)(x) & ((1u << (w)) - 1u))
// carry column i (biased, 64-bit) into column i + 1 (biased, 64-bit); r[i] = the reference's remainder
#define KYB_CARRY(T, r, i, w)                                        \
    {                                                                \
        const int64_t c_ = (T)[i] >> (w);                            \
        (T)[(i) + 1] += c_;                                          \
        (r)[i] = (int32_t)KYB_LOW((T)[i], w) - (1 << ((w) - 1));     \
    }

// Reduce ten biased 64-bit columns to limbs with ONE carry chain 0 -> 1 -> ... -> 9 -> (x19) 0 -> 1.  The
// reference (fe.go feMul tail) interleaves two chains (0, 4, 1, 5, ...) for the benefit of an out-of-order CPU and
// pays for it with a second visit of column 4; here three waves per SIMD hide the longer dependency chain and the
// visit is saved (11 carries instead of 12).  Limbs 4 and 5 may differ from the reference's by a carry unit -- the
// same field element, and within the same bounds: every limb is a rounded remainder, limbs 0 and 1 as in the reference.
// Column 0 is visited twice: its first remainder stays biased (b0), so the second visit needs no rounding add, and
// that last carry is small enough to be added to the finished limb 1 in 32 bits.
KYB_DEV void fe_carry_store(fe& h, int64_t T[10]) {
    int32_t r[10];
    int64_t c = T[0] >> 26;
    T[1] += c;
    const uint32_t b0 = KYB_LOW(T[0], 26);
    KYB_CARRY(T, r, 1, 25);
    KYB_CARRY(T, r, 2, 26);
    KYB_CARRY(T, r, 3, 25);
    KYB_CARRY(T, r, 4, 26);
    KYB_CARRY(T, r, 5, 25);
    KYB_CARRY(T, r, 6, 26);
    KYB_CARRY(T, r, 7, 25);
    KYB_CARRY(T, r, 8, 26);
    c = T[9] >> 25;
    r[9] = (int32_t)KYB_LOW(T[9], 25) - (1 << 24);
    const int64_t t0 = (int64_t)b0 + c * 19;
    c = t0 >> 26;
    r[0] = (int32_t)KYB_LOW(t0, 26) - (1 << 25);
    r[1] += (int32_t)c;
#pragma unroll
    for (int i = 0; i < 10; i++) h.v[i] = r[i];
}

// The same chain with the columns computed INSIDE it: col(k, addend) returns addend + (the ten products of column k).
// An odd column takes the carry out of the even column below it as that addend -- the first v_mad_i64_i32 of the
// column adds it for nothing, where fe_carry_store spends a 64-bit add (an odd column has no bias of its own: it
// arrives inside the carry, fe_bias above); even columns keep their bias as the addend and add the carry afterwards.
// The callers pin the accumulator after EVERY product (KYB_PIN64): left alone, the compiler splits each column into
// two partial sums to shorten the dependency chain and joins them with a 64-bit add, and moves the carry to the end
// of the column as another one -- 17 v_lshl_add_u64 per multiplication where this form has 4.  Every 3-operand VALU
// instruction costs what a multiply-add costs (4.6-5.2 cycles per wave64 instruction against 2.6-2.9 for a 2-operand
// 32-bit one: tools/valu_rates.hip, profiles/r02_valu_rates.jsonl), so the 13 adds are worth 13 products; the serial
// chain costs an s_nop between back-to-back dependent multiply-adds, which the other two waves of the SIMD fill.
// Window loop of ed25519_mul_kernel 5 376 -> 5 146 VALU instructions (v_lshl_add_u64 404 -> 185, scratch 920 -> 572 B);
// 2^20 variable-base scalar multiplications 11.85 -> 11.10 ms on the same box.  Same limbs as fe_carry_store.
template <class ColF>
KYB_DEV void fe_chain_store(fe& h, ColF col) {
    int32_t r[10];
    const int64_t T0 = col(0, fe_bias(0));
    int64_t c = T0 >> 26;
    const uint32_t b0 = KYB_LOW(T0, 26);
#pragma unroll
    for (int k = 1; k < 10; k++) {
        const int w = (k & 1) ? 25 : 26;
        const int64_t T = (k & 1) ? col(k, c) : col(k, fe_bias(k)) + c;
        c = T >> w;
        r[k] = (int32_t)KYB_LOW(T, w) - (1 << (w - 1));
    }
    const int64_t t0 = (int64_t)b0 + c * 19;
    c = t0 >> 26;
    r[0] = (int32_t)KYB_LOW(t0, 26) - (1 << 25);
    r[1] += (int32_t)c;
#pragma unroll
    for (int i = 0; i < 10; i++) h.v[i] = r[i];
}

// h = f * g
KYB_DEV void fe_mul(fe& h, const fe& f, const fe& g) {
    KYB_FE_NOTE(KYB_FE_MAG(f), KYB_FE_MAG(g), 1.0);
    int32_t g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        g19[i] = 19 * g.v[i];
        f2[i] = 2 * f.v[i];
    }
    fe_chain_store(h, [&](int k, int64_t acc) {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            const bool wrap = i > k;
            const bool both_odd = (i & 1) && (j & 1);
            const int32_t a = both_odd ? f2[i] : f.v[i];
            const int32_t b = wrap ? g19[j] : g.v[j];
            acc += (int64_t)a * (int64_t)b;
            KYB_PIN64(acc);
        }
        return acc;
    });
    KYB_FE_MAG_SET(h, 1.01);
}

// h = (DBL ? 2 : 1) * f^2   (55 products)
template <bool DBL>
KYB_DEV void fe_sq_t(fe& h, const fe& f) {
    KYB_FE_NOTE(KYB_FE_MAG(f), KYB_FE_MAG(f), DBL ? 2.0 : 1.0);
    int32_t f2[10], f19[10], f38[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        f2[i] = 2 * f.v[i];
        f19[i] = 19 * f.v[i];
        f38[i] = (int32_t)(2u * (uint32_t)f19[i]);  // used for odd i only; an even limb's 38 f may wrap (unsigned: defined)
    }
    fe_chain_store(h, [&](int k, int64_t addend) {
        int64_t acc = DBL ? 0 : addend;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            if (i > j) continue;
            const bool wrap = (i + j) >= 10;
            const bool both_odd = (i & 1) && (j & 1);
            const int32_t a = i == j ? f.v[i] : f2[i];
            const int32_t b = wrap ? (both_odd ? f38[j] : f19[j]) : (both_odd ? f2[j] : f.v[j]);
            acc += (int64_t)a * (int64_t)b;
            KYB_PIN64(acc);
        }
        return DBL ? (acc + acc) + addend : acc;
    });
    KYB_FE_MAG_SET(h, 1.01);
}
// h = (dbl ? 2 : 1) * f^2 with a per-lane choice (the cooperative doubling of the MSM tail squares X, Y, X + Y and
// 2-squares Z in the four lanes of one instruction stream)
KYB_DEV void fe_sq_sel(fe& h, const fe& f, bool dbl) {
    KYB_FE_NOTE(KYB_FE_MAG(f), KYB_FE_MAG(f), 2.0);
    fe g = f;
    int32_t f2[10], f19[10], f38[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        f2[i] = 2 * g.v[i];
        f19[i] = 19 * g.v[i];
        f38[i] = (int32_t)(2u * (uint32_t)f19[i]);  // used for odd i only; an even limb's 38 f may wrap (unsigned: defined)
    }
    int64_t t[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        int64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            if (i > j) continue;
            const bool wrap = (i + j) >= 10;
            const bool both_odd = (i & 1) && (j & 1);
            const int32_t a = i == j ? g.v[i] : f2[i];
            const int32_t b = wrap ? (both_odd ? f38[j] : f19[j]) : (both_odd ? f2[j] : g.v[j]);
            acc += (int64_t)a * (int64_t)b;
        }
        t[k] = (dbl ? (acc + acc) : acc) + fe_bias(k);
    }
    fe_carry_store(h, t);
    KYB_FE_MAG_SET(h, 1.01);
}
KYB_DEV void fe_sq(fe& h, const fe& f) { fe_sq_t<false>(h, f); }
KYB_DEV void fe_sq2(fe& h, const fe& f) { fe_sq_t<true>(h, f); }

// n >= 1 squarings; a real loop (not unrolled) keeps code size down in the
// 250-squaring exponentiation chains.
KYB_DEV void fe_sqn(fe& h, const fe& f, int n) {
    fe_sq(h, f);
#pragma unroll 1
    for (int i = 1; i < n; i++) fe_sq(h, h);
}

// t = z^(2^250-1), z11 = z^11
KYB_DEV void fe_pow_2_250_1(fe& out, fe& z11, const fe& z) {
    fe z2, z9, t, a5, a10, a20, a50, a100;
    fe_sq(z2, z);
    fe_sqn(t, z2, 2);
    fe_mul(z9, t, z);
    fe_mul(z11, z9, z2);
    fe_sq(t, z11);
    fe_mul(a5, t, z9);  // 2^5 - 1
    fe_sqn(t, a5, 5);
    fe_mul(a10, t, a5);
    fe_sqn(t, a10, 10);
    fe_mul(a20, t, a10);
    fe_sqn(t, a20, 20);
    fe_mul(t, t, a20);  // 2^40 - 1
    fe_sqn(t, t, 10);
    fe_mul(a50, t, a10);
    fe_sqn(t, a50, 50);
    fe_mul(a100, t, a50);
    fe_sqn(t, a100, 100);
    fe_mul(t, t, a100);  // 2^200 - 1
    fe_sqn(t, t, 50);
    fe_mul(out, t, a50);  // 2^250 - 1
}
// out = z^(p-2)
KYB_DEV void fe_invert(fe& out, const fe& z) {
    fe t, z11;
    fe_pow_2_250_1(t, z11, z);
    fe_sqn(t, t, 5);
    fe_mul(out, t, z11);
}
// out = z^((p-5)/8) = z^(2^252-3)
KYB_DEV void fe_pow22523(fe& out, const fe& z) {
    fe t, z11;
    fe_pow_2_250_1(t, z11, z);
    fe_sqn(t, t, 2);
    fe_mul(out, t, z);
}

// Canonical little-endian bytes as eight 32-bit words.
KYB_DEV void fe_towords(uint32_t w[8], const fe& f) {
    int32_t h[10];
#pragma unroll
    for (int i = 0; i < 10; i++) h[i] = f.v[i];
    // q = floor((h + 19) / 2^255) computed limb by limb: 0 or 1 for |h| < 2^255ish
    int32_t q = (19 * h[9] + (1 << 24)) >> 25;
#pragma unroll
    for (int i = 0; i < 10; i++) q = (h[i] + q) >> ((i & 1) ? 25 : 26);
    h[0] += 19 * q;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int w_ = (i & 1) ? 25 : 26;
        int32_t c = h[i] >> w_;
        h[i + 1] += c;
        h[i] -= c << w_;
    }
    h[9] &= (1 << 25) - 1;  // drop 2^255 * q
    // pack: limb offsets 0,26,51,77,102,128,153,179,204,230
    uint32_t u[10];
#pragma unroll
    for (int i = 0; i < 10; i++) u[i] = (uint32_t)h[i];
    w[0] = u[0] | (u[1] << 26);
    w[1] = (u[1] >> 6) | (u[2] << 19);
    w[2] = (u[2] >> 13) | (u[3] << 13);
    w[3] = (u[3] >> 19) | (u[4] << 6);
    w[4] = u[5] | (u[6] << 25);
    w[5] = (u[6] >> 7) | (u[7] << 19);
    w[6] = (u[7] >> 13) | (u[8] << 12);
    w[7] = (u[8] >> 20) | (u[9] << 6);
}
// Limbs from eight little-endian words; bit 255 is ignored (fe.go:91).
KYB_DEV void fe_fromwords(fe& h, const uint32_t w[8]) {
    h.v[0] = (int32_t)(w[0] & 0x3ffffff);
    h.v[1] = (int32_t)(((w[0] >> 26) | (w[1] << 6)) & 0x1ffffff);
    h.v[2] = (int32_t)(((w[1] >> 19) | (w[2] << 13)) & 0x3ffffff);
    h.v[3] = (int32_t)(((w[2] >> 13) | (w[3] << 19)) & 0x1ffffff);
    h.v[4] = (int32_t)((w[3] >> 6) & 0x3ffffff);
    h.v[5] = (int32_t)(w[4] & 0x1ffffff);
    h.v[6] = (int32_t)(((w[4] >> 25) | (w[5] << 7)) & 0x3ffffff);
    h.v[7] = (int32_t)(((w[5] >> 19) | (w[6] << 13)) & 0x1ffffff);
    h.v[8] = (int32_t)(((w[6] >> 12) | (w[7] << 20)) & 0x3ffffff);
    h.v[9] = (int32_t)((w[7] >> 6) & 0x1ffffff);
    KYB_FE_MAG_SET(h, 2.0);
}
KYB_DEV bool fe_isnegative(const fe& f) {
    uint32_t w[8];
    fe_towords(w, f);
    return w[0] & 1;
}
KYB_DEV bool fe_isnonzero(const fe& f) {
    uint32_t w[8];
    fe_towords(w, f);
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r |= w[i];
    return r != 0;
}

}  // namespace kyb
