// Base-field arithmetic with ONE LIMB PER LANE: a field element is a row of 16 lanes (limb l of its N = 13 thirty-bit
// limbs in lane l of the row, lanes 13..15 zero), four rows per wave.  For the LATENCY-bound chains of the engine -- the
// (nwin - 1) c dependent doublings at the end of an MSM (msm.cuh final stage), the doubling chain of a fixed-base table,
// a key's line walk -- where one or a few points are all the work there is and a lane-per-element kernel leaves 63
// lanes of its wave idle behind a 338-multiply-add field product.  Here a product is 57 multiply-adds deep instead.
//
// Multiplication (Montgomery, R = 2^(N W) = 2^390 -- the residue system of mont.cuh / fp_limbs.cuh, so elements move
// between the forms without conversion), separated operand scanning:
//   1 P   columns c_k = sum a_i b_(k-i), k = 0..25: step s multiplies the row's a, shifted s lanes up (DPP row_shr / row_shl,
//         zero fill), by b_s broadcast over the row (ds_swizzle); column l accumulates in lane l of `lo` (k < 16) or `hi`
//   2     two carry passes (lane l -> lane l + 1; lo lane 15 -> hi lane 0) bring every column below 2^30 + 2^5
//   3 Q   q = (c mod R) n' mod R, n' = -p^-1 mod R a compile-time constant: the same shifted walk, low half only
//   4 QP  lo / hi += q (x) p, two carry passes; the low 13 columns now sum to 0 or to R exactly, and which it is shows in
//         column 12 alone (non-zero iff R) -- no carry ever ripples through the row
//   5     result = columns 13..25, moved back to lanes 0..12.
// Limbs stay REDUNDANT (below 2^30 + 2^6, not below 2^30) and values lazy (a multiple of p is added instead of compared
// away), as in fp_limbs.cuh: the bounds of each formula are stated where the formula is and audited on the host.
//
// The same source compiles for the host (tests/host_harness.cpp): V32 / V64 are then arrays of 64 lanes and every
// primitive a loop over them, so the exact instruction-level algorithm runs on the CPU against the oracle.
#pragma once
#include "mont.cuh"
#define KYB_ROWFP_INCLUDED 1

namespace kyb {
namespace rowfp {

constexpr int ROW = 16;

#if defined(__HIPCC__)  // (both passes of hipcc: the host pass parses the kernels too; the builtins only exist in the device pass)
#define KYB_ROW __device__ __forceinline__
typedef uint32_t V32;
typedef uint64_t V64;
KYB_ROW V32 splat(uint32_t x) { return x; }
KYB_ROW V64 zero64() { return 0; }
KYB_ROW V32 lane_in_row() { return (V32)(__lane_id() & 15u); }
KYB_ROW V32 row_of_lane() { return (V32)(__lane_id() >> 4); }
template <int S> KYB_ROW V32 shr_lanes(V32 x) {  // lane l <- lane l - S of its row, zero below
    if constexpr (S == 0) return x;
    else if constexpr (S >= ROW) return 0;
    else {
#if defined(__HIP_DEVICE_COMPILE__)
        return (V32)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 + S, 0xF, 0xF, true);
#else
        return x;
#endif
    }
}
template <int S> KYB_ROW V32 shl_lanes(V32 x) {  // lane l <- lane l + S of its row, zero above
    if constexpr (S == 0) return x;
    else if constexpr (S >= ROW) return 0;
    else {
#if defined(__HIP_DEVICE_COMPILE__)
        return (V32)__builtin_amdgcn_update_dpp(0, (int)x, 0x100 + S, 0xF, 0xF, true);
#else
        return x;
#endif
    }
}
template <int S> KYB_ROW V32 bcast_lane(V32 x) {  // lane S of the row to every lane of the row
#if defined(__HIP_DEVICE_COMPILE__)
    return (V32)__builtin_amdgcn_ds_swizzle((int)x, 0x10 | (S << 5));  // bit mode: lane' = (lane & 0x10) | S
#else
    return x;
#endif
}
KYB_ROW V32 from_row(V32 x, int r) {  // lane l of row r to lane l of every row
#if defined(__HIP_DEVICE_COMPILE__)
    return (V32)__builtin_amdgcn_ds_bpermute((int)(((unsigned)r * 16u + (__lane_id() & 15u)) << 2), (int)x);
#else
    return x + (V32)r;
#endif
}
KYB_ROW V32 add(V32 a, V32 b) { return a + b; }
KYB_ROW V32 sub(V32 a, V32 b) { return a - b; }
KYB_ROW V32 band(V32 a, V32 b) { return a & b; }
KYB_ROW V32 shr(V32 a, int n) { return a >> n; }
KYB_ROW V32 ne0(V32 a) { return a != 0 ? 1u : 0u; }
KYB_ROW V32 sel(V32 cond, V32 a, V32 b) { return cond ? a : b; }  // per lane
KYB_ROW V32 eq(V32 a, uint32_t k) { return a == k ? 1u : 0u; }
KYB_ROW V64 mad64(V64 acc, V32 a, V32 b) { return acc + (uint64_t)a * b; }
KYB_ROW V64 add64(V64 a, V64 b) { return a + b; }
KYB_ROW V64 shr64(V64 a, int n) { return a >> n; }
KYB_ROW V64 low64(V64 a, uint32_t mask) { return a & (uint64_t)mask; }
KYB_ROW V32 lo32(V64 a) { return (uint32_t)a; }
KYB_ROW V32 hi32(V64 a) { return (uint32_t)(a >> 32); }
KYB_ROW V64 make64(V32 lo, V32 hi) { return ((uint64_t)hi << 32) | lo; }
KYB_ROW V64 widen(V32 a) { return (uint64_t)a; }
template <class Arr> KYB_ROW V32 lane_table(const Arr& t) {  // t.v[lane in row]: a per-lane constant (select chain, hoisted)
    const uint32_t l = __lane_id() & 15u;
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < ROW; i++) r = l == (uint32_t)i ? t.v[i] : r;
    return r;
}
// a row element parked in memory (LDS / global): 16 words, limb l at word l; every row stores the same value, every row loads it
KYB_ROW void store_row(uint32_t* m, V32 v) { m[__lane_id() & 15u] = v; }
KYB_ROW V32 load_row(const uint32_t* m) { return m[__lane_id() & 15u]; }
KYB_ROW void row_sync() { __syncthreads(); }  // (one wave per workgroup: orders its own memory traffic for the compiler)
#define KYB_ROW_LONE if (__lane_id() == 0)
#define KYB_ROW_LANES(n) for (int j_ = (int)__lane_id(); j_ < (n); j_ += 64)
#else
#define KYB_ROW inline
// host emulation of one wave: 64 lanes, four rows of 16
struct V32 {
    uint32_t v[64];
};
struct V64 {
    uint64_t v[64];
};
#define KYB_ROW_EACH for (int i_ = 0; i_ < 64; i_++)
KYB_ROW V32 splat(uint32_t x) { V32 r; KYB_ROW_EACH r.v[i_] = x; return r; }
KYB_ROW V64 zero64() { V64 r; KYB_ROW_EACH r.v[i_] = 0; return r; }
KYB_ROW V32 lane_in_row() { V32 r; KYB_ROW_EACH r.v[i_] = i_ & 15; return r; }
KYB_ROW V32 row_of_lane() { V32 r; KYB_ROW_EACH r.v[i_] = i_ >> 4; return r; }
template <int S> KYB_ROW V32 shr_lanes(V32 x) { V32 r; KYB_ROW_EACH r.v[i_] = (i_ & 15) >= S ? x.v[i_ - S] : 0; return r; }
template <int S> KYB_ROW V32 shl_lanes(V32 x) { V32 r; KYB_ROW_EACH r.v[i_] = (i_ & 15) + S < 16 ? x.v[i_ + S] : 0; return r; }
template <int S> KYB_ROW V32 bcast_lane(V32 x) { V32 r; KYB_ROW_EACH r.v[i_] = x.v[(i_ & ~15) | S]; return r; }
KYB_ROW V32 from_row(V32 x, int r0) { V32 r; KYB_ROW_EACH r.v[i_] = x.v[r0 * 16 + (i_ & 15)]; return r; }
KYB_ROW V32 add(V32 a, V32 b) { V32 r; KYB_ROW_EACH r.v[i_] = a.v[i_] + b.v[i_]; return r; }
KYB_ROW V32 sub(V32 a, V32 b) { V32 r; KYB_ROW_EACH r.v[i_] = a.v[i_] - b.v[i_]; return r; }
KYB_ROW V32 band(V32 a, V32 b) { V32 r; KYB_ROW_EACH r.v[i_] = a.v[i_] & b.v[i_]; return r; }
KYB_ROW V32 shr(V32 a, int n) { V32 r; KYB_ROW_EACH r.v[i_] = a.v[i_] >> n; return r; }
KYB_ROW V32 ne0(V32 a) { V32 r; KYB_ROW_EACH r.v[i_] = a.v[i_] != 0; return r; }
KYB_ROW V32 sel(V32 c, V32 a, V32 b) { V32 r; KYB_ROW_EACH r.v[i_] = c.v[i_] ? a.v[i_] : b.v[i_]; return r; }
KYB_ROW V32 eq(V32 a, uint32_t k) { V32 r; KYB_ROW_EACH r.v[i_] = a.v[i_] == k; return r; }
// the 64-bit accumulators must never wrap: the emulation checks what the bounds in the comments promise
inline int& overflow_count() { static int c = 0; return c; }
KYB_ROW V64 mad64(V64 acc, V32 a, V32 b) {
    V64 r;
    KYB_ROW_EACH {
        const unsigned __int128 t = (unsigned __int128)acc.v[i_] + (unsigned __int128)a.v[i_] * b.v[i_];
        if (t >> 64) overflow_count()++;
        r.v[i_] = (uint64_t)t;
    }
    return r;
}
KYB_ROW V64 add64(V64 a, V64 b) {
    V64 r;
    KYB_ROW_EACH {
        r.v[i_] = a.v[i_] + b.v[i_];
        if (r.v[i_] < a.v[i_]) overflow_count()++;
    }
    return r;
}
KYB_ROW V64 shr64(V64 a, int n) { V64 r; KYB_ROW_EACH r.v[i_] = a.v[i_] >> n; return r; }
KYB_ROW V64 low64(V64 a, uint32_t m) { V64 r; KYB_ROW_EACH r.v[i_] = a.v[i_] & m; return r; }
KYB_ROW V32 lo32(V64 a) { V32 r; KYB_ROW_EACH r.v[i_] = (uint32_t)a.v[i_]; return r; }
KYB_ROW V32 hi32(V64 a) { V32 r; KYB_ROW_EACH r.v[i_] = (uint32_t)(a.v[i_] >> 32); return r; }
KYB_ROW V64 make64(V32 lo, V32 hi) { V64 r; KYB_ROW_EACH r.v[i_] = ((uint64_t)hi.v[i_] << 32) | lo.v[i_]; return r; }
KYB_ROW V64 widen(V32 a) { V64 r; KYB_ROW_EACH r.v[i_] = a.v[i_]; return r; }
template <class Arr> KYB_ROW V32 lane_table(const Arr& t) { V32 r; KYB_ROW_EACH r.v[i_] = t.v[i_ & 15]; return r; }
KYB_ROW void store_row(uint32_t* m, V32 v) { for (int i = 0; i < 16; i++) m[i] = v.v[i]; }
KYB_ROW V32 load_row(const uint32_t* m) { V32 r; KYB_ROW_EACH r.v[i_] = m[i_ & 15]; return r; }
KYB_ROW void row_sync() {}
#define KYB_ROW_LONE
#define KYB_ROW_LANES(n) for (int j_ = 0; j_ < (n); j_++)
#endif

template <int S> KYB_ROW V64 shr_lanes64(V64 x) { return make64(shr_lanes<S>(lo32(x)), shr_lanes<S>(hi32(x))); }
template <int S> KYB_ROW V64 shl_lanes64(V64 x) { return make64(shl_lanes<S>(lo32(x)), shl_lanes<S>(hi32(x))); }

struct Arr16 {
    uint32_t v[ROW];
};

// compile-time constants of the field C (N limbs of W bits, C::P, C::NINV = -p^-1 mod 2^W)
template <class C>
struct K {
    static constexpr int N = C::N, W = C::W;
    static constexpr uint32_t MASK = (1u << W) - 1;
    static_assert(N <= 13 && W == 30, "rowfp: written for 13 x 30-bit limbs in rows of 16 lanes");
    // n' = -p^-1 mod 2^(N W): the quotient digits of the number 1 (1 + n' p = 0 mod R)
    static constexpr Arr16 nprime() {
        uint64_t t[N + 1] = {};
        t[0] = 1;
        Arr16 q{};
        for (int i = 0; i < N; i++) {
            const uint64_t m = ((t[0] & MASK) * C::NINV) & MASK;
            q.v[i] = (uint32_t)m;
            uint64_t c = 0;
            for (int j = 0; j < N; j++) {
                const uint64_t x = t[j] + m * C::P[j] + c;
                t[j] = x & MASK;
                c = x >> W;
            }
            t[N] += c;
            for (int j = 0; j < N; j++) t[j] = t[j + 1];  // the low limb is zero by construction
            t[N] = 0;
        }
        return q;
    }
    static constexpr Arr16 NPR = nprime();
    // K p as limbs that no subtrahend limb (below 2^31) can exceed: 2^31 lent to every limb but the top one and paid back one
    // limb up (2^31 at limb l = 2 at limb l + 1).  a + borrow<K> - b is then non-negative limb by limb for b below K p.
    template <int KK>
    static constexpr Arr16 borrow() {
        Arr16 d{};
        uint64_t c = 0;
        for (int j = 0; j < N; j++) {
            const uint64_t x = (uint64_t)C::P[j] * KK + c;
            d.v[j] = j + 1 < N ? (uint32_t)(x & MASK) : (uint32_t)x;
            c = x >> W;
        }
        for (int j = 0; j < N; j++) {
            uint64_t x = d.v[j];
            if (j + 1 < N) x += 1ull << 31;
            if (j > 0) x -= 2;
            d.v[j] = (uint32_t)x;
        }
        return d;
    }
    // lanes of a row that hold limbs (others: 0); and the mask that cuts q to 390 bits (limb 12 to W bits)
    static constexpr Arr16 limb_lanes() {
        Arr16 m{};
        for (int j = 0; j < N; j++) m.v[j] = 0xFFFFFFFFu;
        return m;
    }
    static constexpr Arr16 q_mask() {
        Arr16 m{};
        for (int j = 0; j < N; j++) m.v[j] = j + 1 < N ? 0xFFFFFFFFu : MASK;
        return m;
    }
    // 1 in the Montgomery domain (R mod p) as limbs: the factor that brings a lazy value below p (1 + K p / R)
    static constexpr Arr16 one_limbs() {
        Arr16 o{};
        for (int j = 0; j < N; j++) {
            const int bit = W * j, idx = bit >> 5, sh = bit & 31;
            uint64_t v = C::ONE[idx];
            if (idx + 1 < C::NWORDS) v |= (uint64_t)C::ONE[idx + 1] << 32;
            v >>= sh;
            o.v[j] = j + 1 < N ? (uint32_t)(v & MASK) : (uint32_t)v;
        }
        return o;
    }
    // k in the Montgomery domain (k R mod p) as limbs, for small k: R mod p added k times with a conditional subtraction
    static constexpr Arr16 small_limbs(int k) {
        uint64_t acc[N] = {};
        const Arr16 one = one_limbs();
        for (int t = 0; t < k; t++) {
            uint64_t c = 0;
            for (int j = 0; j < N; j++) {
                const uint64_t x = acc[j] + one.v[j] + c;
                acc[j] = j + 1 < N ? (x & MASK) : x;
                c = j + 1 < N ? (x >> W) : 0;
            }
            // acc >= p ?  compare from the top limb
            bool ge = true;
            for (int j = N - 1; j >= 0; j--) {
                if (acc[j] != C::P[j]) {
                    ge = acc[j] > C::P[j];
                    break;
                }
            }
            if (ge) {
                int64_t b = 0;
                for (int j = 0; j < N; j++) {
                    int64_t x = (int64_t)acc[j] - (int64_t)C::P[j] + b;
                    if (j + 1 < N) {
                        b = x < 0 ? -1 : 0;
                        x &= (int64_t)MASK;
                    }
                    acc[j] = (uint64_t)x;
                }
            }
        }
        Arr16 o{};
        for (int j = 0; j < N; j++) o.v[j] = (uint32_t)acc[j];
        return o;
    }
    static constexpr Arr16 only_lane(int l) {
        Arr16 m{};
        m.v[l] = 0xFFFFFFFFu;
        return m;
    }
};

// per-lane constants a formula keeps in registers across its chain
template <class C>
struct Ctx {
    V32 limb_lanes, q_mask, lane12;
};
template <class C>
KYB_ROW Ctx<C> make_ctx() {
    Ctx<C> c;
    c.limb_lanes = lane_table(K<C>::limb_lanes());
    c.q_mask = lane_table(K<C>::q_mask());
    c.lane12 = lane_table(K<C>::only_lane(C::N - 1));
    return c;
}

// one carry pass over the 26 columns held as lo (columns 0..15, one per lane) and hi (columns 16..31)
template <class C>
KYB_ROW void carry_pass(V64& lo, V64& hi) {
    constexpr int W = C::W;
    const V64 cl = shr64(lo, W), ch = shr64(hi, W);
    lo = add64(low64(lo, K<C>::MASK), shr_lanes64<1>(cl));
    hi = add64(add64(low64(hi, K<C>::MASK), shr_lanes64<1>(ch)), shl_lanes64<ROW - 1>(cl));
}
// the second pass: columns below 2^62 in, so the carries fit 32 bits and travel as one register; columns come out as
// 32-bit values (below 2^30 + the carry)
template <class C>
KYB_ROW void carry_pass_narrow(V64 lo, V64 hi, V32& l, V32& h) {
    constexpr int W = C::W;
    const V32 cl = lo32(shr64(lo, W)), ch = lo32(shr64(hi, W));
    const V32 m = splat(K<C>::MASK);
    l = add(band(lo32(lo), m), shr_lanes<1>(cl));
    h = add(add(band(lo32(hi), m), shr_lanes<1>(ch)), shl_lanes<ROW - 1>(cl));
}
// both passes over ONE accumulator whose carry out of lane 15 is not wanted (the quotient q: cut at R anyway)
template <class C>
KYB_ROW V32 carry_twice_low(V64 x) {
    constexpr int W = C::W;
    const V64 c1 = shr64(x, W);
    const V64 y = add64(low64(x, K<C>::MASK), shr_lanes64<1>(c1));  // below 2^30 + 2^34
    const V32 c2 = lo32(shr64(y, W));
    return add(band(lo32(y), splat(K<C>::MASK)), shr_lanes<1>(c2));
}

// lo / hi += x (x) y: x a row element (limbs below 2^30 + 2^6), y_s = Y::get(s) -- a broadcast limb of a second row element
// or a compile-time limb of a constant.  Column j + s of the product lands in lane (j + s) of lo, or (j + s - 16) of hi.
template <class C, class Y>
KYB_ROW void accumulate_product(V64& lo, V64& hi, V32 x, const Y& y) {
    constexpr int N = C::N;
#define KYB_ROW_STEP(S)                                                               \
    if constexpr (S < N) {                                                            \
        lo = mad64(lo, shr_lanes<S>(x), y.template get<S>());                         \
        if constexpr (S + N - 1 >= ROW) hi = mad64(hi, shl_lanes<ROW - S>(x), y.template get<S>()); \
    }
    KYB_ROW_STEP(0) KYB_ROW_STEP(1) KYB_ROW_STEP(2) KYB_ROW_STEP(3) KYB_ROW_STEP(4) KYB_ROW_STEP(5) KYB_ROW_STEP(6)
    KYB_ROW_STEP(7) KYB_ROW_STEP(8) KYB_ROW_STEP(9) KYB_ROW_STEP(10) KYB_ROW_STEP(11) KYB_ROW_STEP(12)
#undef KYB_ROW_STEP
}
template <class C>
struct BcastLimbs {  // the limbs of a row element, each broadcast over its row
    V32 b[C::N];
    template <int S> KYB_ROW V32 get() const { return b[S]; }
};
template <class C>
KYB_ROW BcastLimbs<C> broadcast_limbs(V32 y) {
    BcastLimbs<C> r;
#define KYB_ROW_B(S) if constexpr (S < C::N) r.b[S] = bcast_lane<S>(y);
    KYB_ROW_B(0) KYB_ROW_B(1) KYB_ROW_B(2) KYB_ROW_B(3) KYB_ROW_B(4) KYB_ROW_B(5) KYB_ROW_B(6)
    KYB_ROW_B(7) KYB_ROW_B(8) KYB_ROW_B(9) KYB_ROW_B(10) KYB_ROW_B(11) KYB_ROW_B(12)
#undef KYB_ROW_B
    return r;
}
template <class C>
struct PLimbs {
    template <int S> KYB_ROW V32 get() const { return splat(C::P[S]); }
};
template <class C>
struct NprLimbs {
    template <int S> KYB_ROW V32 get() const { return splat(K<C>::NPR.v[S]); }
};

// r = a b R^-1 mod p as a value below (Ka Kb p / R + 1 + 2^-20) p for operands below Ka p, Kb p (Ka Kb below R / p: the
// result is below 2p + a hair), limbs of the operands below 2^30 + 2^6, limbs of the result below 2^30 + 2^5 + 2.
// Column bound: 13 (2^30 + 2^6)^2 < 2^63.71 after P; 2^30 + 2^5 + 13 (2^30 + 2^6) 2^30 < 2^63.71 after QP.
template <class C>
KYB_ROW V32 mul(const Ctx<C>& cx, V32 a, V32 b) {
    constexpr int N = C::N;
    V64 lo = zero64(), hi = zero64();
    accumulate_product<C>(lo, hi, a, broadcast_limbs<C>(b));
    V32 l32, h32;
    carry_pass<C>(lo, hi);                     // columns below 2^30 + 2^34
    carry_pass_narrow<C>(lo, hi, l32, h32);    // below 2^30 + 2^5 (a carry of 2^4 + 1 on top of a 30-bit rest)
    // q = (c mod R) n' mod R from the low N columns; columns N.. of this half product are not computed (lanes >= N are
    // cut by q_mask together with the bits of limb N - 1 above R)
    V64 qa = zero64();
    {
        const V32 cl = band(l32, cx.limb_lanes);
        const NprLimbs<C> npr;
        constexpr int Nq = C::N;
#define KYB_ROW_Q(S) if constexpr (S < Nq) qa = mad64(qa, shr_lanes<S>(cl), npr.template get<S>());
        KYB_ROW_Q(0) KYB_ROW_Q(1) KYB_ROW_Q(2) KYB_ROW_Q(3) KYB_ROW_Q(4) KYB_ROW_Q(5) KYB_ROW_Q(6)
        KYB_ROW_Q(7) KYB_ROW_Q(8) KYB_ROW_Q(9) KYB_ROW_Q(10) KYB_ROW_Q(11) KYB_ROW_Q(12)
#undef KYB_ROW_Q
    }
    const V32 q = band(carry_twice_low<C>(qa), cx.q_mask);  // below 2^30 + 2^5 per limb, limb N - 1 below 2^30: q below R (1 + 2^-24)
    lo = widen(l32);
    hi = widen(h32);
    accumulate_product<C>(lo, hi, q, PLimbs<C>());
    carry_pass<C>(lo, hi);
    carry_pass_narrow<C>(lo, hi, l32, h32);
    // the low N columns now hold 0 or R in total; R shows as a non-zero column N - 1 and is one unit of column N
    const V32 top = band(ne0(l32), band(cx.lane12, splat(1)));
    const V32 l13 = add(l32, shr_lanes<1>(top));
    return add(shl_lanes<N>(l13), shr_lanes<ROW - N>(h32));
}

// one carry pass over a row element (limbs below 2^32): limbs below 2^30 + 4 afterwards.  The top limb keeps what it
// has: values stay below 2^390, so it never carries out.
template <class C>
KYB_ROW V32 carry(V32 x) {
    return add(band(x, splat(K<C>::MASK)), shr_lanes<1>(shr(x, C::W)));
}
template <class C>
KYB_ROW V32 add2(V32 a, V32 b) { return carry<C>(add(a, b)); }
template <class C>
KYB_ROW V32 dbl(V32 a) { return carry<C>(add(a, a)); }
template <class C>
KYB_ROW V32 triple(V32 a) { return carry<C>(add(add(a, a), a)); }
// a - b + KK p for b below KK p: the borrow-proof limbs of KK p make every limb difference non-negative
template <int KK, class C>
KYB_ROW V32 sub_k(V32 a, V32 b, V32 borrow_kk) {
    return carry<C>(sub(add(a, borrow_kk), b));
}

// ---- a point of y^2 = x^3 + b in Jacobian coordinates, one row per coordinate: (X, Y, Z), infinity = (Z = 0).
// Doubling (dbl-2009-l with the squarings that only save work in a lane-per-element multiplier written as products):
//   A = X^2, B = Y^2, Z3 = (2Y) Z | D = (2X)(2B) = 4 X Y^2, C8 = (2B)(4B) = 8 Y^4, G = (3A)^2 | Y3 = 3A (D - X3) - C8,
//   X3 = G - 2D -- seven products in three levels.  Value bounds (multiples of p) for inputs X < 7, Y < 5, Z < 2:
//   2X < 14, 2Y < 10; A, B, Z3 < 2 (products of 49, 25, 20 < 512); 2B < 4, 4B < 8, E = 3A < 6; D, C8, G < 2 (56, 32,
//   36); 2D < 4; X3 = G + 5p - 2D < 7; t = D + 8p - X3 < 10; E t: 60; Y3 = E t + 3p - C8 < 5.  Outputs X3 < 7, Y3 < 5,
//   Z3 < 2: the chain closes.
template <class C>
struct JacRow {
    V32 X, Y, Z;
};
template <class C>
struct DblConsts {
    V32 b3, b5, b8;  // borrow-proof limbs of 3p, 5p, 8p
};
template <class C>
KYB_ROW DblConsts<C> make_dbl_consts() {
    DblConsts<C> d;
    d.b3 = lane_table(K<C>::template borrow<3>());
    d.b5 = lane_table(K<C>::template borrow<5>());
    d.b8 = lane_table(K<C>::template borrow<8>());
    return d;
}
// every row computes the whole doubling of its own point (four points per wave)
template <class C>
KYB_ROW void jac_dbl(const Ctx<C>& cx, const DblConsts<C>& dc, JacRow<C>& p) {
    const V32 X2 = dbl<C>(p.X), Y2 = dbl<C>(p.Y);
    const V32 A = mul<C>(cx, p.X, p.X), B = mul<C>(cx, p.Y, p.Y);
    const V32 Z3 = mul<C>(cx, Y2, p.Z);
    const V32 B2 = dbl<C>(B), B4 = dbl<C>(B2), E = triple<C>(A);
    const V32 D = mul<C>(cx, X2, B2), C8 = mul<C>(cx, B2, B4), G = mul<C>(cx, E, E);
    const V32 X3 = sub_k<5, C>(G, dbl<C>(D), dc.b5);
    const V32 t = sub_k<8, C>(D, X3, dc.b8);
    const V32 Y3 = sub_k<3, C>(mul<C>(cx, E, t), C8, dc.b3);
    p.X = X3;
    p.Y = Y3;
    p.Z = Z3;
}
// the four rows of the wave hold the SAME point and share the products of each level (row r computes the r-th product,
// the results are gathered over the rows): a doubling is three multiplications deep
template <class C>
KYB_ROW V32 pick4(V32 row, V32 a0, V32 a1, V32 a2, V32 a3) {
    return sel(eq(row, 0), a0, sel(eq(row, 1), a1, sel(eq(row, 2), a2, a3)));
}
template <class C>
KYB_ROW void jac_dbl_wave(const Ctx<C>& cx, const DblConsts<C>& dc, V32 row, JacRow<C>& p) {
    const V32 X2 = dbl<C>(p.X), Y2 = dbl<C>(p.Y);
    V32 m = mul<C>(cx, pick4<C>(row, p.X, p.Y, Y2, Y2), pick4<C>(row, p.X, p.Y, p.Z, p.Z));
    const V32 A = from_row(m, 0), B = from_row(m, 1), Z3 = from_row(m, 2);
    const V32 B2 = dbl<C>(B), B4 = dbl<C>(B2), E = triple<C>(A);
    m = mul<C>(cx, pick4<C>(row, X2, B2, E, E), pick4<C>(row, B2, B4, E, E));
    const V32 D = from_row(m, 0), C8 = from_row(m, 1), G = from_row(m, 2);
    const V32 X3 = sub_k<5, C>(G, dbl<C>(D), dc.b5);
    const V32 t = sub_k<8, C>(D, X3, dc.b8);
    const V32 Y3 = sub_k<3, C>(mul<C>(cx, E, t), C8, dc.b3);  // the one product of the last level, in every row
    p.X = X3;
    p.Y = Y3;
    p.Z = Z3;
}

// ---- packed elements (NWORDS x 32-bit words, fully reduced: Fp<C>) -> rows: lane l takes limb l out of the two words
// that hold its bits; every row of the wave loads the same element
#if defined(__HIPCC__)
template <class C>
KYB_ROW V32 load_packed(const uint32_t* __restrict__ w) {
    const uint32_t l = __lane_id() & 15u;
    const uint32_t bit = (uint32_t)C::W * l, idx = bit >> 5, sh = bit & 31u;
    const bool limb = l < (uint32_t)C::N;
    const uint32_t lo = limb ? w[idx < (uint32_t)C::NWORDS ? idx : 0] : 0u;
    const uint32_t hi = limb && idx + 1 < (uint32_t)C::NWORDS ? w[idx + 1] : 0u;
    const uint64_t v = (((uint64_t)hi << 32) | lo) >> sh;
    return l + 1 < (uint32_t)C::N ? (uint32_t)v & K<C>::MASK : (limb ? (uint32_t)v : 0u);
}
#else
template <class C>
KYB_ROW V32 load_packed(const uint32_t* w) {
    V32 r;
    KYB_ROW_EACH {
        const int l = i_ & 15;
        uint64_t v = 0;
        if (l < C::N) {
            const int bit = C::W * l, idx = bit >> 5, sh = bit & 31;
            v = w[idx];
            if (idx + 1 < C::NWORDS) v |= (uint64_t)w[idx + 1] << 32;
            v >>= sh;
            if (l + 1 < C::N) v &= K<C>::MASK;
        }
        r.v[i_] = (uint32_t)v;
    }
    return r;
}
#endif
// a lazy element (value below 22p, redundant limbs) as a value below p (1 + 22 p / R) < 2p: one product with R mod p.
// Its limbs, carried once through by ONE lane (they are redundant: a ripple, but of 13 additions, once per chain), are
// what fp_finish takes.
template <class C>
KYB_ROW V32 below_2p(const Ctx<C>& cx, V32 x) {
    return mul<C>(cx, x, lane_table(K<C>::one_limbs()));
}
// a^e for a PUBLIC exponent (nbits bits in 32-bit words; nibble windows as mont.cuh fp_pow_words: nbits / 4 + 14 products
// next to nbits squarings), every row computing the same power: a chain stays a chain, but each link is a row product
// (~0.5 us) instead of a lane's (~1.1 us).  a: limbs below 2^30 + 2^6, value below 2p; tab: 15 row elements of memory
// (LDS) for a^1 .. a^15.  Result below 2p + a hair like any product (operands of every product below 3p: 9 < R / p).
template <class C>
KYB_ROW V32 pow_words(const Ctx<C>& cx, V32 a, uint32_t (*tab)[ROW], const uint32_t* e, int nbits) {
    V32 t = a;
    store_row(tab[0], t);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int j = 1; j < 15; j++) {
        t = mul<C>(cx, t, a);
        store_row(tab[j], t);
    }
    row_sync();
    V32 acc = a;
    bool one = true;  // (uniform: the exponent is public)
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int w = (nbits + 3) / 4 - 1; w >= 0; w--) {
        if (!one) {
            acc = mul<C>(cx, acc, acc);
            acc = mul<C>(cx, acc, acc);
            acc = mul<C>(cx, acc, acc);
            acc = mul<C>(cx, acc, acc);
        }
        const int bit = 4 * w;
        const uint32_t nib = (e[bit >> 5] >> (bit & 31)) & 15u;
        if (nib) {
            const V32 f = load_row(tab[nib - 1]);
            acc = one ? f : mul<C>(cx, acc, f);
            one = false;
        }
    }
    row_sync();  // (the table may be written again)
    return one ? lane_table(K<C>::one_limbs()) : acc;
}
// the 13 redundant limbs of a row element -> a packed, fully reduced Fp (run by one lane on limbs it read from LDS / memory)
template <class C>
KYB_HD void finish_limbs(Fp<C>& r, const uint32_t* limbs) {
    uint32_t s[C::N];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < C::N; j++) {
        const uint64_t t = (uint64_t)limbs[j] + c;
        s[j] = j + 1 < C::N ? (uint32_t)(t & K<C>::MASK) : (uint32_t)t;
        c = t >> C::W;
    }
    fp_finish<C>(r, s);
}

// ---- Fp2 = Fp[i] / (i^2 + 1) on the wave: an element is two row elements (c0, c1) held by all four rows; a LEVEL is four
// base-field products at once, row r computing the r-th and the results gathered over the rows.  An Fp2 product is one
// level (a0 b0, a1 b1, a0 b1, a1 b0), two Fp2 squarings share one ((a0 + a1)(a0 - a1), a0 a1 each).
// Value bounds: a product comes out as c0 < 5p (P0 - P1 + 3p), c1 < 4p; the caller keeps every operand product below R / p.
template <class C>
struct F2 {
    V32 c0, c1;
};
template <class C>
struct F2Consts {
    V32 b3, b8, b16, b24;  // borrow-proof limbs of 3p, 8p, 16p, 24p: a subtrahend must stay a whole p BELOW the constant used
};
template <class C>
KYB_ROW F2Consts<C> make_f2_consts() {
    F2Consts<C> d;
    d.b3 = lane_table(K<C>::template borrow<3>());
    d.b8 = lane_table(K<C>::template borrow<8>());
    d.b16 = lane_table(K<C>::template borrow<16>());
    d.b24 = lane_table(K<C>::template borrow<24>());
    return d;
}
template <class C>
KYB_ROW void level4(const Ctx<C>& cx, V32 row, V32 x0, V32 y0, V32 x1, V32 y1, V32 x2, V32 y2, V32 x3, V32 y3, V32& p0, V32& p1,
                    V32& p2, V32& p3) {
    const V32 m = mul<C>(cx, pick4<C>(row, x0, x1, x2, x3), pick4<C>(row, y0, y1, y2, y3));
    p0 = from_row(m, 0);
    p1 = from_row(m, 1);
    p2 = from_row(m, 2);
    p3 = from_row(m, 3);
}
template <class C>
KYB_ROW F2<C> f2_mul(const Ctx<C>& cx, const F2Consts<C>& k, V32 row, const F2<C>& a, const F2<C>& b) {
    V32 p0, p1, p2, p3;
    level4<C>(cx, row, a.c0, b.c0, a.c1, b.c1, a.c0, b.c1, a.c1, b.c0, p0, p1, p2, p3);
    return F2<C>{sub_k<3, C>(p0, p1, k.b3), add2<C>(p2, p3)};
}
// a^2 and b^2 in one level; ba / bb: borrow constants at least a whole p above a.c1 / b.c1.  Results: c0 < 2p, c1 < 4p.
template <class C>
KYB_ROW void f2_sqr2(const Ctx<C>& cx, V32 row, F2<C>& ra, const F2<C>& a, V32 ba, F2<C>& rb, const F2<C>& b, V32 bb) {
    const V32 sa = add2<C>(a.c0, a.c1), da = carry<C>(sub(add(a.c0, ba), a.c1));
    const V32 sb = add2<C>(b.c0, b.c1), db = carry<C>(sub(add(b.c0, bb), b.c1));
    V32 p0, p1, p2, p3;
    level4<C>(cx, row, sa, da, a.c0, a.c1, sb, db, b.c0, b.c1, p0, p1, p2, p3);
    ra = F2<C>{p0, dbl<C>(p1)};
    rb = F2<C>{p2, dbl<C>(p3)};
}
template <class C> KYB_ROW F2<C> f2_add(const F2<C>& a, const F2<C>& b) { return F2<C>{add2<C>(a.c0, b.c0), add2<C>(a.c1, b.c1)}; }
template <class C> KYB_ROW F2<C> f2_dbl(const F2<C>& a) { return F2<C>{dbl<C>(a.c0), dbl<C>(a.c1)}; }
template <class C> KYB_ROW F2<C> f2_triple(const F2<C>& a) { return F2<C>{triple<C>(a.c0), triple<C>(a.c1)}; }
template <class C> KYB_ROW F2<C> f2_sub(const F2<C>& a, const F2<C>& b, V32 bk) {  // a - b + K p per component
    return F2<C>{carry<C>(sub(add(a.c0, bk), b.c0)), carry<C>(sub(add(a.c1, bk), b.c1))};
}
template <class C> KYB_ROW F2<C> f2_neg(const F2<C>& a, V32 bk) {  // K p - a
    return F2<C>{carry<C>(sub(bk, a.c0)), carry<C>(sub(bk, a.c1))};
}
template <class C> KYB_ROW void f2_store(uint32_t (*m)[ROW], const F2<C>& a) {
    store_row(m[0], a.c0);
    store_row(m[1], a.c1);
}
template <class C> KYB_ROW F2<C> f2_load(const uint32_t (*m)[ROW]) { return F2<C>{load_row(m[0]), load_row(m[1])}; }

}  // namespace rowfp
}  // namespace kyb
