// Ed25519 batched scalar multiplication kernels for gfx950 + their C-ABI entry
// points.  One scalar multiplication per lane, uniform control flow across the
// wave (signed radix-16 fixed windows), integer VALU only.
//
// Replaces, in the reference (group/edwards25519):
//   geScalarMultBase  ge.go:373-417      -> ed25519_mul_base_kernel
//   geScalarMult      ge.go:443-502      -> ed25519_mul_kernel
//   geScalarMultVartime ge_mult_vartime.go:11 (semantics via KYB_F_VARTIME)
//   FromBytes/ToBytes ge.go:99-150       -> fused into the kernels
// The element-wise kernels of this unit that named no register budget (Add, UnmarshalBinary, Hash, the encoder) take two
// waves per SIMD (hd.h KYB_TU_WAVES): at the default they came out at 268-317 registers, one wave per SIMD.  2^20
// elements, same box: UnmarshalBinary 2.03 -> 1.66 ms, Add 3.31 -> 2.61 ms, Hash 13.7 -> 9.0 ms
// (profiles/r04_tu_wave_budgets.json); a three-wave budget gives the same.  The MSM's one-lane reduce kernel (msm.cuh)
// keeps its registers: a latency-bound grid of 544 waves, 2 % slower at 256.
#ifndef KYB_TU_WAVES
#define KYB_TU_WAVES 2
#endif
#include "context.h"
#include "ge25519.cuh"
#include "msm.cuh"
#include "ed25519_h2c.cuh"
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace kyb {

// Fixed-base table: entry (pos, j) = (j + 1) * 256^pos * B in affine (y+x, y-x, 2dxy) form, j < 136: a signed
// radix-256 digit is d0 + 16 d1 of two signed radix-16 digits in [-8, 8], so |digit| <= 136.  One entry is one
// 128-byte line (30 limbs + 2 pad words); the 574 KB table is read through L2 -- it does not fit LDS, but it
// halves the additions of the reference's 32 x 8 table (ge.go:373-417: 64 additions + 4 doublings) to 32.
constexpr int ED_TAB_POS = 33;  // 32 byte positions + 2^256*B for the 65th radix-16 digit
constexpr int ED_TAB_ENT = 136;
constexpr int ED_TAB_STRIDE = 32;
constexpr int ED_TAB_WORDS = ED_TAB_POS * ED_TAB_ENT * ED_TAB_STRIDE;

KYB_DEV void load_words8(uint32_t w[8], const uint32_t* __restrict__ p) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0];
    const uint4 b = reinterpret_cast<const uint4*>(p)[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
KYB_DEV void store_words8(uint32_t* __restrict__ p, const uint32_t w[8]) {
    reinterpret_cast<uint4*>(p)[0] = make_uint4(w[0], w[1], w[2], w[3]);
    reinterpret_cast<uint4*>(p)[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// ---------------------------------------------------------------- table build
// Thread (i, j) computes (j+1) * 256^i * B by MSB-first double-and-add and
// stores its affine (y+x, y-x, 2dxy).  Runs once per device for the standard base (point == nullptr), and once per
// call for the shared base of a large kyb_ed25519_mul_same_base batch (share.PriPoly.Commit with b != nil,
// share/poly.go:143-149): the table costs 4 544 short multiplications, after which every coefficient is 32 mixed
// additions instead of a 64-window ladder on a freshly decoded point.  *ok = 0 when `point` does not decode.
__global__ void ed25519_build_base_table_kernel(int32_t* __restrict__ tab, const uint32_t* __restrict__ point,
                                                uint32_t* __restrict__ ok) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ED_TAB_POS * ED_TAB_ENT) return;
    const int pos = t / ED_TAB_ENT, j = t - pos * ED_TAB_ENT;
    ge_p3 B;
    if (point) {
        uint32_t pw[8];
        load_words8(pw, point);
        const bool good = ge_p3_fromwords(B, pw);
        if (t == 0) *ok = good ? 1u : 0u;
    } else {
        B.X = fe_bx(); B.Y = fe_by(); fe_1(B.Z); B.T = fe_bt();
    }
    ge_cached cB;
    ge_p3_to_cached(cB, B);
    ge_p3 acc;
    ge_p3_0(acc);
    ge_p1p1 r;
    // multiplier = (j+1) << (8*pos): 8 significant bits then 8*pos doublings
#pragma unroll 1
    for (int bit = 7; bit >= 0; bit--) {
        ge_dbl(r, acc.X, acc.Y, acc.Z);
        ge_p1p1_to_p3(acc, r);
        if (((j + 1) >> bit) & 1) {
            ge_add(r, acc, cB);
            ge_p1p1_to_p3(acc, r);
        }
    }
#pragma unroll 1
    for (int k = 0; k < 8 * pos; k++) {
        ge_dbl(r, acc.X, acc.Y, acc.Z);
        ge_p1p1_to_p3(acc, r);
    }
    fe zi, x, y, ypx, ymx, xy2d, z;
    fe_invert(zi, acc.Z);
    fe_mul(x, acc.X, zi);
    fe_mul(y, acc.Y, zi);
    fe_add(ypx, y, x);
    fe_sub(ymx, y, x);
    fe_mul(xy2d, x, y);
    fe_mul(xy2d, xy2d, fe_d2());
    // one pass through mul-by-one normalises y+x / y-x to reduced limbs
    fe_1(z);
    fe_mul(ypx, ypx, z);
    fe_mul(ymx, ymx, z);
    int32_t* o = tab + (size_t)t * ED_TAB_STRIDE;
#pragma unroll
    for (int l = 0; l < 10; l++) {
        o[l] = ypx.v[l];
        o[10 + l] = ymx.v[l];
        o[20 + l] = xy2d.v[l];
    }
    o[30] = o[31] = 0;
}

// ------------------------------------------------------ deferred encoding (batched inversion)
// The field inversion of ToBytes (254 squarings, 22 % of a fixed-base and 6 % of a variable-base
// multiplication) is amortised: the multiplication kernels park (X, Y, Z) in HBM (30 limbs, 120 B per
// element) and a second kernel inverts ENC_CHUNK Z's per lane with Montgomery's trick -- 3 multiplications
// per element plus 1/ENC_CHUNK of an inversion -- before encoding.
constexpr int ENC_CHUNK = 16;

KYB_DEV void store_proj(int32_t* __restrict__ proj, size_t idx, const ge_p3& h) {
    int32_t* o = proj + idx * 30;
#pragma unroll
    for (int l = 0; l < 10; l++) {
        o[l] = h.X.v[l];
        o[10 + l] = h.Y.v[l];
        o[20 + l] = h.Z.v[l];
    }
}
KYB_DEV void load_fe(fe& f, const int32_t* __restrict__ p) {
#pragma unroll
    for (int l = 0; l < 10; l++) f.v[l] = p[l];
}
__global__ __launch_bounds__(64, KYB_TU_WAVES) void ed25519_encode_kernel(size_t n, const int32_t* __restrict__ proj,
                                                            const uint8_t* __restrict__ status,
                                                            uint32_t* __restrict__ out) {
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t lo = lane * ENC_CHUNK;
    if (lo >= n) return;
    const int cnt = (int)((n - lo) < (size_t)ENC_CHUNK ? (n - lo) : (size_t)ENC_CHUNK);
    fe pre[ENC_CHUNK];  // pre[j] = Z_0 ... Z_j
    fe z, acc;
    load_fe(acc, proj + lo * 30 + 20);
    pre[0] = acc;
#pragma unroll 1
    for (int j = 1; j < cnt; j++) {
        load_fe(z, proj + (lo + j) * 30 + 20);
        fe_mul(acc, acc, z);
        pre[j] = acc;
    }
    fe inv;
    fe_invert(inv, acc);  // 1 / (Z_0 ... Z_{cnt-1})
#pragma unroll 1
    for (int j = cnt - 1; j >= 0; j--) {
        fe zi, X, Y;
        load_fe(z, proj + (lo + j) * 30 + 20);
        if (j > 0) {
            fe_mul(zi, inv, pre[j - 1]);  // 1 / Z_j
            fe_mul(inv, inv, z);
        } else {
            zi = inv;
        }
        load_fe(X, proj + (lo + j) * 30);
        load_fe(Y, proj + (lo + j) * 30 + 10);
        uint32_t w[8];
        ge_encode_with_zinv(w, X, Y, zi);
        if (status && status[lo + j]) {
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = 0;
        }
        store_words8(out + (lo + j) * 8, w);
    }
}

// ------------------------------------------------------------ fixed-base mul
// Every lane gathers its own 128-byte entry (eight 16-byte loads from one line, served by L2).
KYB_DEV void select_precomp_tab(ge_precomp& t, const int32_t* __restrict__ tab, int pos, int b) {
    const bool neg = b < 0;
    const int babs = neg ? -b : b;
    const int4* e = reinterpret_cast<const int4*>(tab + (size_t)(pos * ED_TAB_ENT + (babs ? babs - 1 : 0)) * ED_TAB_STRIDE);
    int32_t w[32];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int4 x = e[i];
        w[4 * i] = x.x;
        w[4 * i + 1] = x.y;
        w[4 * i + 2] = x.z;
        w[4 * i + 3] = x.w;
    }
#pragma unroll
    for (int l = 0; l < 10; l++) {
        t.ypx.v[l] = w[l];
        t.ymx.v[l] = w[10 + l];
        t.xy2d.v[l] = w[20 + l];
    }
    if (babs == 0) {  // identity: (1, 1, 0)
        fe_1(t.ypx);
        fe_1(t.ymx);
        fe_0(t.xy2d);
    }
    ge_precomp_cneg(t, neg);
}

// h = sum_k D_k 256^k B with D_k = e[2k] + 16 e[2k+1] built from the reference's signed radix-16 digits, so the
// value -- including the reference's behaviour for scalars >= 2^255 on the constant-time path (recode16) -- and
// therefore the encoding is exactly that of geScalarMultBase.
// three waves per SIMD (168 registers): the chained columns of fe_mul leave a lone pair of waves waiting on each other
// (1.61 -> 1.56 ms per 2^20 against the two-wave budget, same box)
__global__ __launch_bounds__(256, 3) void ed25519_mul_base_kernel(
    size_t n, const uint32_t* __restrict__ scalars, uint32_t* __restrict__ out,
    const int32_t* __restrict__ tab, uint32_t flags, int32_t* __restrict__ proj) {
    const bool full = (flags & KYB_F_VARTIME) != 0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (size_t)gridDim.x * blockDim.x) {
        uint32_t a[8];
        load_words8(a, scalars + idx * 8);
        int8_t e[65];
        recode16(e, a, full);
        ge_p3 h;
        ge_p3_0(h);
        ge_precomp t;
        ge_p1p1 r;
        const int npos = full ? 33 : 32;  // uniform across the grid
#pragma unroll 1
        for (int k = 0; k < npos; k++) {
            const int d = k < 32 ? (int)e[2 * k] + 16 * (int)e[2 * k + 1] : (int)e[64];
            if (full && __ballot(d != 0) == 0) continue;  // variable-time: nothing to add at this position in any lane
            select_precomp_tab(t, tab, k, d);
            ge_madd(r, h, t);
            ge_p1p1_to_p3(h, r);
        }
        if (proj) {  // encoding deferred to ed25519_encode_kernel
            store_proj(proj, idx, h);
            continue;
        }
        uint32_t w[8];
        ge_p3_towords(w, h);
        store_words8(out + idx * 8, w);
    }
}

// KYB_F_UNIFORM: the fixed-base multiplication with memory addresses and control flow that do not depend on the scalar --
// the shape of the reference's constant-time geScalarMultBase (ge.go:373-417 with selectPreComputed's CMove scan,
// ge.go:352-371): one mixed addition per signed radix-16 digit, its operand picked by reading ALL eight candidates of
// the digit's position (the same lines for every lane: the radix-256 table holds j 256^k B for j <= 136, so the
// candidates m 16^(2k) B and m 16^(2k+1) B = 16 m 256^k B, m = 1..8, are its rows m - 1 and 16 m - 1) and keeping one
// by mask; the sign by conditional move; no window skipped, no lane-dependent branch.  64 additions instead of 32 and
// 240 selects per digit: the price of not indexing by the digit (measured: DESIGN.md section 5).
KYB_DEV void select_precomp_uniform(ge_precomp& t, const int32_t* __restrict__ tab, int i, int b) {
    const bool neg = b < 0;
    const int babs = neg ? -b : b;
    const int pos = i >> 1, mult = (i & 1) ? 16 : 1;
    fe_1(t.ypx);
    fe_1(t.ymx);
    fe_0(t.xy2d);
#pragma unroll 1
    for (int m = 1; m <= 8; m++) {
        const int4* e = reinterpret_cast<const int4*>(tab + (size_t)(pos * ED_TAB_ENT + (m * mult - 1)) * ED_TAB_STRIDE);
        int32_t w[32];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int4 x = e[q];
            w[4 * q] = x.x;
            w[4 * q + 1] = x.y;
            w[4 * q + 2] = x.z;
            w[4 * q + 3] = x.w;
        }
        const int32_t keep = -(int32_t)(babs == m);  // all ones for the digit's own multiple
#pragma unroll
        for (int l = 0; l < 10; l++) {
            t.ypx.v[l] ^= (t.ypx.v[l] ^ w[l]) & keep;
            t.ymx.v[l] ^= (t.ymx.v[l] ^ w[10 + l]) & keep;
            t.xy2d.v[l] ^= (t.xy2d.v[l] ^ w[20 + l]) & keep;
        }
    }
    ge_precomp_cneg(t, neg);
}
__global__ __launch_bounds__(256, 3) void ed25519_mul_base_uniform_kernel(
    size_t n, const uint32_t* __restrict__ scalars, uint32_t* __restrict__ out,
    const int32_t* __restrict__ tab, int32_t* __restrict__ proj) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (size_t)gridDim.x * blockDim.x) {
        uint32_t a[8];
        load_words8(a, scalars + idx * 8);
        int8_t e[65];
        recode16(e, a, false);
        ge_p3 h;
        ge_p3_0(h);
        ge_precomp t;
        ge_p1p1 r;
#pragma unroll 1
        for (int i = 0; i < 64; i++) {
            select_precomp_uniform(t, tab, i, (int)e[i]);
            ge_madd(r, h, t);
            ge_p1p1_to_p3(h, r);
        }
        if (proj) {
            store_proj(proj, idx, h);
            continue;
        }
        uint32_t w[8];
        ge_p3_towords(w, h);
        store_words8(out + idx * 8, w);
    }
}

// --------------------------------------------------------- variable-base mul
// Shared ladder: h = sum_i e[i] 16^i * A using an 8-entry cached table.  The table (8 x 160 B) is per-lane state
// that fits neither registers nor LDS at a useful occupancy.  Small batches keep it in the lane's private (scratch)
// memory; large ones in a global slab of 1280 contiguous bytes per lane: scratch is interleaved per dword across
// the lanes of a wave, so an *indexed* entry read drags in up to 8 rows per dword (measured: 40 GB fetched per 2^20
// launch for 11 GB of entries), while a lane-contiguous entry is ten 16-byte loads from two or three cache lines.
struct TabScratch {
    ge_cached tab[8];
    KYB_DEV void put(int j, const ge_cached& c) { tab[j] = c; }
    KYB_DEV void get(ge_cached& c, int j) const { c = tab[j]; }
};
struct TabGlobal {
    int4* base;  // this lane's 8 x 10 int4
    KYB_DEV void put(int j, const ge_cached& c) {
        int4* q = base + j * 10;
        const fe* f[4] = {&c.YpX, &c.YmX, &c.Z, &c.T2d};
        int32_t w[40];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int l = 0; l < 10; l++) w[10 * k + l] = f[k]->v[l];
#pragma unroll
        for (int i = 0; i < 10; i++) q[i] = make_int4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
    }
    KYB_DEV void get(ge_cached& c, int j) const {
        const int4* q = base + j * 10;
        int32_t w[40];
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int4 x = q[i];
            w[4 * i] = x.x;
            w[4 * i + 1] = x.y;
            w[4 * i + 2] = x.z;
            w[4 * i + 3] = x.w;
        }
        fe* f[4] = {&c.YpX, &c.YmX, &c.Z, &c.T2d};
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int l = 0; l < 10; l++) f[k]->v[l] = w[10 * k + l];
    }
};
template <bool UNI = false, class Tab>
KYB_DEV void select_cached(ge_cached& c, const Tab& tab, int b) {
    const bool neg = b < 0;
    const int babs = neg ? -b : b;
    if constexpr (UNI) {
        // KYB_F_UNIFORM: selectCached's scan (ge.go:419-435) -- all eight entries read, one kept by mask
        ge_cached_0(c);
#pragma unroll 1
        for (int m = 1; m <= 8; m++) {
            ge_cached x;
            tab.get(x, m - 1);
            const int32_t keep = -(int32_t)(babs == m);
#pragma unroll
            for (int l = 0; l < 10; l++) {
                c.YpX.v[l] ^= (c.YpX.v[l] ^ x.YpX.v[l]) & keep;
                c.YmX.v[l] ^= (c.YmX.v[l] ^ x.YmX.v[l]) & keep;
                c.Z.v[l] ^= (c.Z.v[l] ^ x.Z.v[l]) & keep;
                c.T2d.v[l] ^= (c.T2d.v[l] ^ x.T2d.v[l]) & keep;
            }
        }
    } else {
        tab.get(c, babs ? babs - 1 : 0);
        if (babs == 0) ge_cached_0(c);
    }
    ge_cached_cneg(c, neg);
}

// Highest radix-16 digit position that can be non-zero for ANY lane of the wave (the recoding may carry one digit past
// the scalar's top nibble).  Wave-uniform by construction: the variable-time path below starts its ladder there.
KYB_DEV int wave_top_digit(const uint32_t a[8]) {
    int bits = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (a[i]) bits = 32 * i + 32 - __builtin_clz(a[i]);
    int t = (bits + 3) >> 2;  // digits 0 .. t may be non-zero (t: the carry)
    if (t > 64) t = 64;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int o = __shfl_xor(t, off, 64);
        t = o > t ? o : t;
    }
    t = __builtin_amdgcn_readfirstlane(t);
#endif
    return t;
}

// KYB_F_VARTIME (geScalarMultVartime, ge_mult_vartime.go:11-73: every scalar bit counts, running time depends on the
// scalar).  The reference's sliding-window NAF has its non-zero digits at scalar-dependent positions; 64 lanes in
// lock step would execute the union of all of them -- an addition at nearly every bit.  What IS data-dependent and
// still wave-uniform: the ladder starts at the highest digit any lane of the wave needs (`top`: short scalars -- the
// 128-bit coefficients of sign/bdn, small Lagrange indices -- run proportionally fewer windows) and a window whose
// digit is zero in EVERY lane skips its addition and the conversion that feeds it.  Random 253-bit scalars take the
// same 64 windows as the constant-structure path.
template <bool UNI = false, class Tab>
KYB_DEV void ge_scalarmult_w4(ge_p3& h, const int8_t e[65], const ge_p3& A, bool full, Tab& tab, int vt_top) {
    ge_p1p1 t;
    ge_p3 u;
    ge_p2 r;
    ge_cached c;
    ge_p3_to_cached(c, A);
    tab.put(0, c);
#pragma unroll 1
    for (int i = 0; i < 7; i++) {
        ge_add(t, A, c);
        ge_p1p1_to_p3(u, t);
        ge_p3_to_cached(c, u);
        tab.put(i + 1, c);
    }
    ge_p3_0(u);
    int top = 63;
    if (full) top = vt_top;  // uniform across the wave
    select_cached<UNI>(c, tab, e[top]);
    ge_add(t, u, c);
#pragma unroll 1
    for (int i = top - 1; i >= 0; i--) {
        ge_p1p1_to_p2(r, t);
        ge_dbl(t, r.X, r.Y, r.Z);
        ge_p1p1_to_p2(r, t);
        ge_dbl(t, r.X, r.Y, r.Z);
        ge_p1p1_to_p2(r, t);
        ge_dbl(t, r.X, r.Y, r.Z);
        ge_p1p1_to_p2(r, t);
        ge_dbl(t, r.X, r.Y, r.Z);
#if defined(__HIP_DEVICE_COMPILE__)
        if (!UNI && full && __ballot(e[i] != 0) == 0) continue;  // no lane adds anything in this window
#endif
        ge_p1p1_to_p3(u, t);
        select_cached<UNI>(c, tab, e[i]);
        ge_add(t, u, c);
    }
    ge_p1p1_to_p3(h, t);
}

// points_stride = 8 words for per-element points, 0 for one shared base
// Register budget for three waves per SIMD (<= 170 registers): the default allocation (160 VGPRs + 63 AGPRs of
// spill space = 2 waves) was 7 % slower, a budget for four waves (128) 18 % slower (spills reach scratch).
// UNI (KYB_F_UNIFORM): the window table is scanned, not indexed -- geScalarMult's access pattern (ge.go:443-502)
template <bool GTAB, bool UNI = false>
__global__ __launch_bounds__(128, 3) void ed25519_mul_kernel(
    size_t n, const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ points,
    size_t points_stride, uint32_t* __restrict__ out, uint8_t* __restrict__ status,
    uint32_t flags, int32_t* __restrict__ proj, int4* __restrict__ gtab) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const bool full = !UNI && (flags & KYB_F_VARTIME) != 0;
    uint32_t pw[8], a[8];
    load_words8(pw, points + idx * points_stride);
    load_words8(a, scalars + idx * 8);
    ge_p3 A;
    const bool ok = ge_p3_fromwords(A, pw);
    int8_t e[65];
    recode16(e, a, full);
    ge_p3 h;
    const int vt_top = full ? wave_top_digit(a) : 63;
    if constexpr (GTAB) {
        TabGlobal tab{gtab + idx * 80};
        ge_scalarmult_w4<UNI>(h, e, A, full, tab, vt_top);
    } else {
        TabScratch tab;
        ge_scalarmult_w4<UNI>(h, e, A, full, tab, vt_top);
    }
    if (proj) {  // encoding deferred to ed25519_encode_kernel (status carries the decode verdict)
        store_proj(proj, idx, h);
        if (status) status[idx] = ok ? KYB_ST_OK : KYB_ST_BAD_POINT;
        return;
    }
    uint32_t w[8];
    ge_p3_towords(w, h);
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = 0;
    }
    store_words8(out + idx * 8, w);
    if (status) status[idx] = ok ? KYB_ST_OK : KYB_ST_BAD_POINT;
}

// ---------------------------------------------------------------- host side
int ed25519_build_tables(DeviceCtx* ctx) {
    KYB_HIP_CHECK(hipMalloc(&ctx->ed_base_tab, ED_TAB_WORDS * sizeof(int32_t)));
    hipLaunchKernelGGL(ed25519_build_base_table_kernel, dim3((ED_TAB_POS * ED_TAB_ENT + 63) / 64), dim3(64), 0,
                       nullptr, ctx->ed_base_tab, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    KYB_HIP_CHECK(hipGetLastError());
    KYB_HIP_CHECK(hipStreamSynchronize(nullptr));
    return KYB_OK;
}
void ed25519_free_tables(DeviceCtx* ctx) {
    if (ctx->ed_base_tab) hipFree(ctx->ed_base_tab);
    ctx->ed_base_tab = nullptr;
}

// Batches of at least this many elements take the deferred-encoding path (below it the extra launch costs
// more than the inversions it saves).
constexpr size_t ENC_DEFER_MIN = 4096;

// Grow-only per-stream buffer (context.h WS_ED) for the window tables, the parked (X, Y, Z) triples and a status
// array when the caller passes none: calls on one stream are ordered and reuse it, other streams have their own.
static int ed_proj_workspace(DeviceCtx* ctx, hipStream_t st, size_t n, bool need_status, int32_t** proj,
                             uint8_t** status, int4** gtab = nullptr) {
    // [ window tables: n x 1280 B (variable-base only) | (X, Y, Z): n x 120 B | status: n ]
    const size_t tab_bytes = gtab ? n * 1280 : 0;
    const size_t want = tab_bytes + n * 30 * sizeof(int32_t) + (need_status ? n : 0) + 256;
    void* base;
    int rc = ctx_workspace(ctx, WS_ED, st, want, &base);
    if (rc) return rc;
    if (gtab) *gtab = (int4*)base;
    *proj = (int32_t*)((uint8_t*)base + tab_bytes);
    if (status) *status = (uint8_t*)base + tab_bytes + n * 30 * sizeof(int32_t);
    return KYB_OK;
}

static int launch_mul_base(DeviceCtx* ctx, size_t n, const void* d_scalars, void* d_out, uint32_t flags,
                           hipStream_t st, const int32_t* tab = nullptr) {
    if (!tab) tab = ctx->ed_base_tab;
    if (n == 0) return KYB_OK;
    const int block = 256;
    size_t want = (n + block - 1) / block;
    size_t cap = (size_t)ctx->num_cu * 8;  // grid-stride
    int grid = (int)(want < cap ? want : cap);
    int32_t* proj = nullptr;
    std::lock_guard<std::recursive_mutex> enq_lock(ctx->enq_mu);  // context.h: workspace + its kernels as one unit
    if (n >= ENC_DEFER_MIN) {
        int rc = ed_proj_workspace(ctx, st, n, false, &proj, nullptr);
        if (rc) return rc;
    }
    if (flags & KYB_F_UNIFORM)
        hipLaunchKernelGGL(ed25519_mul_base_uniform_kernel, dim3(grid), dim3(block), 0, st, n,
                           (const uint32_t*)d_scalars, (uint32_t*)d_out, tab, proj);
    else
        hipLaunchKernelGGL(ed25519_mul_base_kernel, dim3(grid), dim3(block), 0, st, n,
                           (const uint32_t*)d_scalars, (uint32_t*)d_out, tab, flags, proj);
    if (proj) {
        const size_t lanes = (n + ENC_CHUNK - 1) / ENC_CHUNK;
        hipLaunchKernelGGL(ed25519_encode_kernel, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, n, proj,
                           (const uint8_t*)nullptr, (uint32_t*)d_out);
    }
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
static int launch_mul(size_t n, const void* d_scalars, const void* d_points, size_t stride, void* d_out,
                      void* d_status, uint32_t flags, hipStream_t st) {
    if (n == 0) return KYB_OK;
    const int block = 128;
    size_t grid = (n + block - 1) / block;
    int32_t* proj = nullptr;
    int4* gtab = nullptr;
    uint8_t* stat = (uint8_t*)d_status;
    DeviceCtx* ctx;
    int rc = get_ctx(&ctx);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> enq_lock(ctx->enq_mu);
    if (n >= ENC_DEFER_MIN) {
        rc = ed_proj_workspace(ctx, st, n, stat == nullptr, &proj, stat ? nullptr : &stat, &gtab);
        if (rc) return rc;
        if (flags & KYB_F_UNIFORM)
            hipLaunchKernelGGL((ed25519_mul_kernel<true, true>), dim3((unsigned)grid), dim3(block), 0, st, n,
                               (const uint32_t*)d_scalars, (const uint32_t*)d_points, stride, (uint32_t*)d_out, stat, flags,
                               proj, gtab);
        else
            hipLaunchKernelGGL(ed25519_mul_kernel<true>, dim3((unsigned)grid), dim3(block), 0, st, n,
                               (const uint32_t*)d_scalars, (const uint32_t*)d_points, stride, (uint32_t*)d_out, stat, flags,
                               proj, gtab);
    } else if (flags & KYB_F_UNIFORM) {
        hipLaunchKernelGGL((ed25519_mul_kernel<false, true>), dim3((unsigned)grid), dim3(block), 0, st, n,
                           (const uint32_t*)d_scalars, (const uint32_t*)d_points, stride, (uint32_t*)d_out, stat, flags,
                           proj, gtab);
    } else {
        hipLaunchKernelGGL(ed25519_mul_kernel<false>, dim3((unsigned)grid), dim3(block), 0, st, n,
                           (const uint32_t*)d_scalars, (const uint32_t*)d_points, stride, (uint32_t*)d_out, stat, flags,
                           proj, gtab);
    }
    if (proj) {
        const size_t lanes = (n + ENC_CHUNK - 1) / ENC_CHUNK;
        hipLaunchKernelGGL(ed25519_encode_kernel, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, n, proj,
                           (const uint8_t*)stat, (uint32_t*)d_out);
    }
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}

}  // namespace kyb

using namespace kyb;

// flags of the scalar-multiplication entry points: KYB_F_VARTIME or KYB_F_UNIFORM, never both (one asks for the
// scalar-dependent schedule, the other forbids it)
static bool ed_mul_flags_bad(uint32_t flags) {
    return (flags & ~(KYB_F_VARTIME | KYB_F_UNIFORM)) || (flags & (KYB_F_VARTIME | KYB_F_UNIFORM)) == (KYB_F_VARTIME | KYB_F_UNIFORM);
}

extern "C" {

int kyb_ed25519_mul_base_dev(size_t n, const void* d_scalars, void* d_out, uint32_t flags, void* stream) {
    if ((n && (!d_scalars || !d_out)) || ed_mul_flags_bad(flags)) {
        set_error("kyb_ed25519_mul_base_dev: bad argument");
        return KYB_E_ARG;
    }
    DeviceCtx* ctx;
    int rc = get_ctx(&ctx);
    if (rc) return rc;
    return launch_mul_base(ctx, n, d_scalars, d_out, flags, (hipStream_t)stream);
}

int kyb_ed25519_mul_dev(size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
                        uint32_t flags, void* stream) {
    if ((n && (!d_scalars || !d_points || !d_out)) || ed_mul_flags_bad(flags)) {
        set_error("kyb_ed25519_mul_dev: bad argument");
        return KYB_E_ARG;
    }
    DeviceCtx* ctx;
    int rc = get_ctx(&ctx);
    if (rc) return rc;
    return launch_mul(n, d_scalars, d_points, 8, d_out, d_status, flags, (hipStream_t)stream);
}

static int mul_host(size_t n, const uint8_t* scalars, const uint8_t* points, size_t stride, uint8_t* out,
                    uint8_t* status, uint32_t flags);

int kyb_ed25519_mul_base(size_t n, const uint8_t* scalars, uint8_t* out, uint32_t flags) {
    if ((n && (!scalars || !out)) || ed_mul_flags_bad(flags)) {
        set_error("kyb_ed25519_mul_base: bad argument");
        return KYB_E_ARG;
    }
    if (n == 0) return KYB_OK;
    if (md_active(n))
        return md_run(n, [&](int, size_t lo, size_t hi) { return kyb_ed25519_mul_base(hi - lo, scalars + 32 * lo, out + 32 * lo, flags); });
    return mul_host(n, scalars, nullptr, 0, out, nullptr, flags);
}

// Host-buffer batches of at least two chunks (pipe_chunk) are cut in chunks and software-pipelined over three
// page-locked staging slots: while chunk i computes, chunk i+1 is copied in and chunk i-1 is copied out.  The caller's
// memory is pageable, and a copy straight from it is staged by the runtime at ~10 GB/s while it blocks the issuing
// thread; instead host threads memcpy a chunk into a pinned slot (and results out of one) and the kernels work on the
// slots in place, over PCIe.  (97 bytes per variable-base element: a quarter of the kernel's time at memcpy speed.)
constexpr size_t PIPE_CHUNK_MAX = PIN_SLOT_ELEMS;  // capacity of a staging slot, in elements (context.h)
constexpr int PIPE_SLOTS = DeviceCtx::NPIN, PIPE_STREAMS = 3;
// Chunks of 2^17 elements alternate between two compute streams, so the tail of one chunk's kernel overlaps the head
// of the next and the pipeline fills in the time one chunk takes to copy.  Same-box sweep at the C ABI, 2^20 elements,
// fixed-base / variable-base ms (tools/gpu/r02_host5.sh): 1 stream x 2^18: 3.13 / 15.30; 1 x 2^17: 3.90 / 15.90;
// 2 x 2^17: 2.78 / 13.93; 2 x 2^18: 2.79 / 14.60; 3 x 2^17: 2.61 / 14.29; 3 x 2^16: 3.24 / 17.11 (resident: 1.6 / 12.0).
// KYB_PIPE_CHUNK / KYB_PIPE_STREAMS override the defaults for such sweeps.
static size_t pipe_chunk(const DeviceCtx* ctx) {
    if (const char* e = getenv("KYB_PIPE_CHUNK")) {  // tuning knob (elements, a multiple of 64)
        const size_t v = (size_t)strtoull(e, nullptr, 10) & ~size_t(63);
        if (v >= 4096) return std::min(PIPE_CHUNK_MAX, v);
    }
    (void)ctx;
    return size_t(1) << 17;
}
static int pipe_nstreams() {
    if (const char* e = getenv("KYB_PIPE_STREAMS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= PIPE_STREAMS) return v;
    }
    return 2;
}
constexpr size_t SAME_BASE_TABLE_MIN = 16384;  // below this the table (4 544 short multiplications) does not pay

static int pipe_streams(DeviceCtx::StagePool* pool) {
    for (int i = 0; i < PIPE_STREAMS; i++)
        if (!pool->pipe[i]) KYB_HIP_CHECK(hipStreamCreateWithFlags(&pool->pipe[i], hipStreamNonBlocking));
    return KYB_OK;
}

// points == nullptr: fixed-base (kyb_ed25519_mul_base); stride 0: one shared base point
static int mul_host(size_t n, const uint8_t* scalars, const uint8_t* points, size_t stride, uint8_t* out,
                    uint8_t* status, uint32_t flags) {
    DeviceCtx* ctx;
    int rc = get_ctx(&ctx);
    if (rc) return rc;
    bool fixed = points == nullptr;
    const size_t npts = fixed ? 0 : (stride ? n : 1);
    StageScope sc_(ctx);
    StageBuf d_s, d_p, d_o, d_st, d_tab;
    if ((rc = d_s.alloc(n * 32))) return rc;
    if ((rc = d_p.alloc(npts * 32))) return rc;
    if ((rc = d_o.alloc(n * 32))) return rc;
    if ((rc = d_st.alloc(n))) return rc;
    const int32_t* tab = nullptr;  // nullptr: the device's table of the standard base
    if (!fixed && !stride && (n >= SAME_BASE_TABLE_MIN || (flags & KYB_F_UNIFORM))) {  // (uniform: the shared table at any size)
        // one shared base and many coefficients: give the base a radix-256 table of its own and take the fixed-base path
        if ((rc = d_tab.alloc(ED_TAB_WORDS * sizeof(int32_t) + 256))) return rc;
        uint32_t* d_ok = (uint32_t*)((uint8_t*)d_tab.p + ED_TAB_WORDS * sizeof(int32_t));
        KYB_HIP_CHECK(hipMemcpy(d_p.p, points, 32, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ed25519_build_base_table_kernel, dim3((ED_TAB_POS * ED_TAB_ENT + 63) / 64), dim3(64), 0, sc_.stream(),
                           (int32_t*)d_tab.p, (const uint32_t*)d_p.p, d_ok);
        uint32_t ok = 0;
        KYB_HIP_CHECK(hipMemcpy(&ok, d_ok, 4, hipMemcpyDeviceToHost));
        if (!ok) {  // the reference's UnmarshalBinary fails once, for every coefficient
            memset(out, 0, n * 32);
            if (status) memset(status, KYB_ST_BAD_POINT, n);
            return KYB_OK;
        }
        if (status) memset(status, 0, n);
        tab = (const int32_t*)d_tab.p;
        fixed = true;
    }
    auto launch = [&](size_t off, size_t cnt, hipStream_t st) -> int {
        uint8_t* s = (uint8_t*)d_s.p + off * 32;
        uint8_t* o = (uint8_t*)d_o.p + off * 32;
        if (fixed) return launch_mul_base(ctx, cnt, s, o, flags, st, tab);
        const uint8_t* p = (const uint8_t*)d_p.p + (stride ? off * 32 : 0);
        return launch_mul(cnt, s, p, stride, o, (uint8_t*)d_st.p + off, flags, st);
    };
    const size_t PIPE_CHUNK = pipe_chunk(ctx);
    if (n < 2 * PIPE_CHUNK) {
        KYB_HIP_CHECK(hipMemcpy(d_s.p, scalars, n * 32, hipMemcpyHostToDevice));
        if (npts) KYB_HIP_CHECK(hipMemcpy(d_p.p, points, npts * 32, hipMemcpyHostToDevice));
        if ((rc = launch(0, n, nullptr))) return rc;
        if ((rc = d_o.download(out, n * 32))) return rc;
        if (status && !fixed) rc = d_st.download(status, n);
        return rc;
    }
    DeviceCtx::StagePool* pool = sc_.pool;  // this call's page-locked slots and pipeline streams
    if ((rc = pipe_streams(pool))) return rc;
    const int nstreams = pipe_nstreams();
    const bool per_point = !fixed && stride;
    const bool want_status = status && !fixed;
    // page-locked slots: [scalars | points] in, [points | status] out, one chunk each, PIPE_SLOTS of each
    if ((rc = ctx_pin_slots(pool))) return rc;
    const size_t nchunks = (n + PIPE_CHUNK - 1) / PIPE_CHUNK;
    std::vector<hipEvent_t> ev_out(nchunks);
    for (size_t i = 0; i < nchunks; i++) KYB_HIP_CHECK(hipEventCreateWithFlags(&ev_out[i], hipEventDisableTiming));
    // chunk i: the host copies it into slot i % PIPE_SLOTS, the kernels run on it in place on stream i % PIPE_STREAMS, a
    // second host thread copies the results out of the slot's out half.  Before a slot is refilled, the chunk that
    // held it has been drained (so its kernel has read the slot's inputs, too): the host's memcpys run while the
    // kernels of the chunks in between execute.
    auto enqueue = [&](size_t i) -> int {
        const size_t off = i * PIPE_CHUNK, cnt = std::min(PIPE_CHUNK, n - off);
        uint8_t* pin = (uint8_t*)pool->pin_in[i % PIPE_SLOTS];
        uint8_t* pout = (uint8_t*)pool->pin_out[i % PIPE_SLOTS];
        hipStream_t s_k = pool->pipe[i % nstreams];
        par_memcpy(pin, scalars + off * 32, cnt * 32);
        if (per_point) par_memcpy(pin + PIPE_CHUNK_MAX * 32, points + off * 32, cnt * 32);
        // zero copy: the kernels read the chunk's scalars / points straight from the page-locked slot over PCIe (one
        // coalesced 2 KB read per wave at the start of ~10 us of arithmetic) and write the encoded results straight
        // into the slot's out half.  (Through hipMemcpyAsync the runtime ran most of these transfers as blit KERNELS,
        // 6 ms of them per 2^20 elements competing with the ladder for the CUs.)
        int r;
        if (fixed) {
            r = launch_mul_base(ctx, cnt, pin, pout, flags, s_k, tab);
        } else {
            const uint8_t* p = per_point ? pin + PIPE_CHUNK_MAX * 32 : (const uint8_t*)d_p.p;
            r = launch_mul(cnt, pin, p, stride, pout, (uint8_t*)d_st.p + off, flags, s_k);
        }
        if (r) return r;
        if (want_status)
            KYB_HIP_CHECK(hipMemcpyAsync(pout + PIPE_CHUNK_MAX * 32, (uint8_t*)d_st.p + off, cnt, hipMemcpyDeviceToHost, s_k));
        KYB_HIP_CHECK(hipEventRecord(ev_out[i], s_k));
        return KYB_OK;
    };
    auto drain = [&](size_t i) -> int {
        const size_t off = i * PIPE_CHUNK, cnt = std::min(PIPE_CHUNK, n - off);
        KYB_HIP_CHECK(hipEventSynchronize(ev_out[i]));
        const uint8_t* pout = (const uint8_t*)pool->pin_out[i % PIPE_SLOTS];
        par_memcpy(out + off * 32, pout, cnt * 32);
        if (want_status) memcpy(status + off, pout + PIPE_CHUNK_MAX * 32, cnt);
        return KYB_OK;
    };
    if (!fixed && !stride) KYB_HIP_CHECK(hipMemcpy(d_p.p, points, 32, hipMemcpyHostToDevice));
    // a second host thread copies results out (chunks in order) while this one copies inputs in: the host's memcpy
    // bandwidth is what bounds the fixed-base path (64 bytes moved per 1.5 us of kernel)
    std::atomic<size_t> enqueued{0}, drained{0};
    std::atomic<int> drain_rc{KYB_OK};
    std::atomic<bool> stop{false};
    const int device = ctx->device;
    std::string drain_err;  // g_err is thread_local: the drainer's message is carried over to the caller after the join
    std::thread drainer([&] {
        if (hipSetDevice(device) != hipSuccess) {
            drain_err = "ed25519 host pipeline: hipSetDevice failed on the drain thread";
            drain_rc.store(KYB_E_HIP);
            return;
        }
        for (size_t i = 0; i < nchunks; i++) {
            while (enqueued.load(std::memory_order_acquire) <= i) {
                if (stop.load(std::memory_order_acquire)) return;
                std::this_thread::yield();
            }
            const int r = drain(i);
            if (r) {
                drain_err = kyb_last_error();
                drain_rc.store(r);
                return;
            }
            drained.store(i + 1, std::memory_order_release);
        }
    });
    for (size_t i = 0; i < nchunks && rc == KYB_OK; i++) {
        while (i >= (size_t)PIPE_SLOTS && drained.load(std::memory_order_acquire) + PIPE_SLOTS <= i && drain_rc.load() == KYB_OK)
            std::this_thread::yield();  // the slot still belongs to chunk i - PIPE_SLOTS
        if (drain_rc.load() != KYB_OK) break;
        rc = enqueue(i);
        if (rc == KYB_OK) enqueued.store(i + 1, std::memory_order_release);
    }
    if (rc != KYB_OK) stop.store(true, std::memory_order_release);
    drainer.join();
    if (rc == KYB_OK && (rc = drain_rc.load()) != KYB_OK) set_error(drain_err);
    hipError_t e2 = hipSuccess;
    for (int i = 0; i < PIPE_STREAMS; i++) {
        const hipError_t e = hipStreamSynchronize(pool->pipe[i]);
        if (e != hipSuccess) e2 = e;
    }
    for (size_t i = 0; i < nchunks; i++) hipEventDestroy(ev_out[i]);
    if (rc == KYB_OK && e2 != hipSuccess) {
        set_error("ed25519 host pipeline: stream synchronisation failed");
        rc = KYB_E_HIP;
    }
    return rc;
}

int kyb_ed25519_mul(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out, uint8_t* status,
                    uint32_t flags) {
    if ((n && (!scalars || !points || !out)) || ed_mul_flags_bad(flags)) {
        set_error("kyb_ed25519_mul: bad argument");
        return KYB_E_ARG;
    }
    if (n == 0) return KYB_OK;
    if (md_active(n))
        return md_run(n, [&](int, size_t lo, size_t hi) {
            return kyb_ed25519_mul(hi - lo, scalars + 32 * lo, points + 32 * lo, out + 32 * lo, status ? status + lo : nullptr, flags);
        });
    return mul_host(n, scalars, points, 8, out, status, flags);
}

int kyb_ed25519_mul_same_base(size_t n, const uint8_t* scalars, const uint8_t point[32], uint8_t* out,
                              uint8_t* status, uint32_t flags) {
    if ((n && (!scalars || !out)) || !point || ed_mul_flags_bad(flags)) {
        set_error("kyb_ed25519_mul_same_base: bad argument");
        return KYB_E_ARG;
    }
    if (n == 0) return KYB_OK;
    if (md_active(n))
        return md_run(n, [&](int, size_t lo, size_t hi) {
            return kyb_ed25519_mul_same_base(hi - lo, scalars + 32 * lo, point, out + 32 * lo, status ? status + lo : nullptr, flags);
        });
    return mul_host(n, scalars, point, 0, out, status, flags);
}

int kyb_ed25519_debug_base_table(int32_t* out) {
    if (!out) return KYB_E_ARG;
    DeviceCtx* ctx;
    int rc = get_ctx(&ctx);
    if (rc) return rc;
    KYB_HIP_CHECK(hipMemcpy(out, ctx->ed_base_tab, ED_TAB_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost));
    return KYB_OK;
}
}

// ---------------------------------------------------------------- MSM (msm.cuh pipeline)
// sum_i a_i * A_i with a_i plain 256-bit integers: what PubPoly.Eval / RecoverCommit
// (share/poly.go:340-348, 449-476) compute with N x (Mul + Add).
namespace kyb {
struct EdMsm {
    using Aff = ge_precomp;  // (y + x, y - x, 2dxy): one unified mixed addition = 7M
    using Acc = ge_p3;
    static constexpr int WIRE = 32, OUT = 32;
    static constexpr bool SCALAR_BE = false;
    static constexpr uint32_t COMBINE_FLAGS = 0;
    static constexpr int DECODE_WAVES = 3;  // 1.73 ms at two waves, 1.34 at three, 1.38 at four (spills)
    __host__ __device__ static size_t wire_size(uint32_t) { return 32; }
    __device__ static int decode(Aff& a, const uint8_t* wire, uint32_t) {
        uint32_t w[8];
        load_words8(w, reinterpret_cast<const uint32_t*>(wire));
        ge_p3 p;
        const bool ok = ge_p3_fromwords(p, w);
        fe one;
        fe_1(one);
        fe_add(a.ypx, p.Y, p.X);
        fe_sub(a.ymx, p.Y, p.X);
        fe_mul(a.ypx, a.ypx, one);  // reduce the sums so table entries stay in the multiplier's input range
        fe_mul(a.ymx, a.ymx, one);
        fe_mul(a.xy2d, p.T, fe_d2());
        return ok ? KYB_ST_OK : KYB_ST_BAD_POINT;
    }
    __device__ static void scalar_words(uint32_t (&k)[8], const uint8_t* wire) {
        load_words8(k, reinterpret_cast<const uint32_t*>(wire));
    }
    // msm.cuh Effective: the integer geScalarMult multiplies by (ge25519.cuh ed_effective_scalar); a negative one
    // takes -P.  Scalars cut short by KYB_F_SCALAR_BITS never reach the top digit.
    static constexpr bool HAS_EFFECTIVE = true;
    __device__ static void effective(uint32_t (&k)[8], Aff& p, int bits) {
        if (bits < 253) return;
        if (ed_effective_scalar(k)) ge_precomp_cneg(p, true);
    }
    __device__ static void identity(Acc& a) { ge_p3_0(a); }
    __device__ static void madd(Acc& acc, const Aff& p, bool neg) {
        ge_precomp t = p;
        ge_precomp_cneg(t, neg);
        ge_p1p1 r;
        ge_madd(r, acc, t);
        ge_p1p1_to_p3(acc, r);
    }
    __device__ static void add(Acc& r, const Acc& a, const Acc& b) {
        ge_cached c;
        ge_p3_to_cached(c, b);
        ge_p1p1 t;
        ge_add(t, a, c);
        ge_p1p1_to_p3(r, t);
    }
    __device__ static void dbl(Acc& r, const Acc& a) {
        ge_p1p1 t;
        ge_dbl(t, a.X, a.Y, a.Z);
        ge_p1p1_to_p3(r, t);
    }
    // Cooperative doubling for the MSM tail (see msm_ws.cuh): four lanes hold the same point; the four squarings of
    // ge_dbl, then the four products of p1p1 -> p3, one per lane, exchanged through LDS: 2 field operations deep
    // instead of 8.  Every thread of the block must call it (barriers inside).
    static constexpr int COOP = 4;
    using Field = fe;
    __device__ static void dbl_coop(Acc& s, int r, fe* sh) {
        fe in = s.X, t, m;
        fe_add(t, s.X, s.Y);
        fe_cmov(in, s.Y, r == 1);
        fe_cmov(in, s.Z, r == 2);
        fe_cmov(in, t, r == 3);
        fe_sq_sel(m, in, r == 2);  // X^2 | Y^2 | 2 Z^2 | (X + Y)^2
        sh[r] = m;
        __syncthreads();
        const fe XX = sh[0], YY = sh[1], ZZ2 = sh[2], t0 = sh[3];
        __syncthreads();
        fe Xp, Yp, Zp, Tp;  // the completed point, as in ge_dbl
        fe_add(Yp, YY, XX);
        fe_sub(Zp, YY, XX);
        fe_sub(Xp, t0, Yp);
        fe_sub(Tp, ZZ2, Zp);
        fe a = Xp, b = Tp;  // X3 = Xp Tp | Y3 = Yp Zp | Z3 = Zp Tp | T3 = Xp Yp
        fe_cmov(a, Yp, r == 1);
        fe_cmov(a, Zp, r == 2);
        fe_cmov(b, Zp, r == 1);
        fe_cmov(b, Yp, r == 3);
        fe_mul(m, a, b);
        sh[r] = m;
        __syncthreads();
        s.X = sh[0];
        s.Y = sh[1];
        s.Z = sh[2];
        s.T = sh[3];
        __syncthreads();
    }
    __device__ static void encode(uint8_t* out, const Acc& a) {
        uint32_t w[8];
        ge_p3_towords(w, a);
        store_words8(reinterpret_cast<uint32_t*>(out), w);
    }
};
}  // namespace kyb

extern "C" {
int kyb_ed25519_msm(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out[32], uint8_t* status) {
    return kyb::msm::run_host<kyb::EdMsm>(n, scalars, points, out, status);
}
int kyb_ed25519_msm_flags(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out[32], uint8_t* status,
                          uint32_t flags) {
    if (flags & ~KYB_F_SCALAR_BITS_MASK) {
        kyb::set_error("kyb_ed25519_msm_flags: only KYB_F_SCALAR_BITS applies");
        return KYB_E_ARG;
    }
    return kyb::msm::run_host<kyb::EdMsm>(n, scalars, points, out, status, flags);
}
int kyb_ed25519_poly_eval(size_t n, const uint32_t* idx, size_t t, const uint8_t* commits, uint8_t* out, uint8_t* status) {
    return kyb::msm::poly_eval_host<kyb::EdMsm>(n, idx, t, commits, out, status, 0);
}
int kyb_ed25519_poly_eval_dev(size_t n, const void* d_idx, size_t t, const void* d_commits, void* d_out, void* d_status,
                              void* stream) {
    kyb::DeviceCtx* ctx;
    int rc = kyb::get_ctx(&ctx);
    if (rc) return rc;
    return kyb::msm::poly_eval_run<kyb::EdMsm>(ctx, n, d_idx, t, d_commits, d_out, d_status, 0, (hipStream_t)stream);
}
int kyb_ed25519_msm_dev(size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
                        void* stream) {
    kyb::DeviceCtx* ctx;
    int rc = kyb::get_ctx(&ctx);
    if (rc) return rc;
    return kyb::msm::run<kyb::EdMsm>(ctx, n, d_scalars, d_points, d_out, d_status, (hipStream_t)stream);
}
}

// ---------------------------------------------------------------- (*point).Hash (point.go:325-334)
namespace kyb {
__global__ __launch_bounds__(64, KYB_TU_WAVES) void ed25519_hash_kernel(size_t n, const uint8_t* __restrict__ msgs, size_t msg_len,
                                                          EdDstArg dst, uint8_t* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    ed_hash_wire(out + 32 * idx, msgs + msg_len * idx, msg_len, dst);
}
}  // namespace kyb

extern "C" {
int kyb_ed25519_hash_dev(size_t n, const void* d_msgs, size_t msg_len, const uint8_t* dst, size_t dst_len, void* d_out,
                         void* stream) {
    if ((n && ((!d_msgs && msg_len) || !d_out)) || !dst || dst_len == 0 || dst_len > 255) {
        kyb::set_error("kyb_ed25519_hash_dev: bad argument (the domain separation tag must be 1..255 bytes)");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    kyb::EdDstArg d;
    memset(&d, 0, sizeof d);
    memcpy(d.b, dst, dst_len);
    d.len = (uint32_t)dst_len;
    hipLaunchKernelGGL(kyb::ed25519_hash_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, n,
                       (const uint8_t*)d_msgs, msg_len, d, (uint8_t*)d_out);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
int kyb_ed25519_hash(size_t n, const uint8_t* msgs, size_t msg_len, const uint8_t* dst, size_t dst_len, uint8_t* out) {
    if (n && ((!msgs && msg_len) || !out)) {
        kyb::set_error("kyb_ed25519_hash: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    kyb::DeviceCtx* ctx;
    int rc = kyb::get_ctx(&ctx);
    if (rc) return rc;
    kyb::StageScope sc_(ctx);
    kyb::StageBuf d_m, d_o;
    rc = d_m.upload(msgs, n * msg_len);
    if (rc == KYB_OK) rc = d_o.alloc(n * 32);
    if (rc == KYB_OK) rc = kyb_ed25519_hash_dev(n, d_m.p, msg_len, dst, dst_len, d_o.p, sc_.stream());
    if (rc == KYB_OK) rc = d_o.download(out, n * 32);
    return rc;
}
}

// ---------------------------------------------------------------- batch Point.Add (point.go:216-223 -> ge.go:183)
namespace kyb {
__global__ __launch_bounds__(128, KYB_TU_WAVES) void ed25519_add_kernel(size_t n, const uint32_t* __restrict__ a,
                                                          const uint32_t* __restrict__ b, uint32_t* __restrict__ out,
                                                          uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    uint32_t wa[8], wb[8], w[8];
    load_words8(wa, a + idx * 8);
    load_words8(wb, b + idx * 8);
    ge_p3 A, B, R;
    const bool ok = ge_p3_fromwords(A, wa) & ge_p3_fromwords(B, wb);
    ge_cached c;
    ge_p3_to_cached(c, B);
    ge_p1p1 t;
    ge_add(t, A, c);
    ge_p1p1_to_p3(R, t);
    ge_p3_towords(w, R);
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = 0;
    }
    store_words8(out + idx * 8, w);
    if (status) status[idx] = ok ? KYB_ST_OK : KYB_ST_BAD_POINT;
}
// out = Marshal(Unmarshal(in)): (*point).UnmarshalBinary (group/edwards25519/point.go:65-70 -> ge.go:110-150) as a batch
// validity check -- bit 255 of y ignored for the field element, y >= p accepted, fails only when x^2 has no root --
// followed by the canonical encoding MarshalBinary (ge.go:99-107) would give back.
__global__ __launch_bounds__(128, KYB_TU_WAVES) void ed25519_unmarshal_kernel(size_t n, const uint32_t* __restrict__ in,
                                                                uint32_t* __restrict__ out,
                                                                uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    uint32_t w[8];
    load_words8(w, in + idx * 8);
    ge_p3 A;
    const bool ok = ge_p3_fromwords(A, w);
    ge_p3_towords(w, A);
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = 0;
    }
    store_words8(out + idx * 8, w);
    if (status) status[idx] = ok ? KYB_ST_OK : KYB_ST_BAD_POINT;
}
}  // namespace kyb
extern "C" {
int kyb_ed25519_unmarshal_dev(size_t n, const void* d_points, void* d_out, void* d_status, void* stream) {
    if (n && (!d_points || !d_out)) {
        kyb::set_error("kyb_ed25519_unmarshal_dev: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    hipLaunchKernelGGL(kyb::ed25519_unmarshal_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0,
                       (hipStream_t)stream, n, (const uint32_t*)d_points, (uint32_t*)d_out, (uint8_t*)d_status);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
int kyb_ed25519_unmarshal(size_t n, const uint8_t* points, uint8_t* out, uint8_t* status) {
    if (n && (!points || !out)) {
        kyb::set_error("kyb_ed25519_unmarshal: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    kyb::DeviceCtx* ctx;
    int rc = kyb::get_ctx(&ctx);
    if (rc) return rc;
    kyb::StageScope sc_(ctx);
    kyb::StageBuf d_p, d_o, d_st;
    rc = d_p.upload(points, n * 32);
    if (rc == KYB_OK) rc = d_o.alloc(n * 32);
    if (rc == KYB_OK) rc = d_st.alloc(n);
    if (rc == KYB_OK) rc = kyb_ed25519_unmarshal_dev(n, d_p.p, d_o.p, d_st.p, sc_.stream());
    if (rc == KYB_OK) rc = d_o.download(out, n * 32);
    if (rc == KYB_OK && status) rc = d_st.download(status, n);
    return rc;
}
int kyb_ed25519_add_dev(size_t n, const void* d_a, const void* d_b, void* d_out, void* d_status, void* stream) {
    if (n && (!d_a || !d_b || !d_out)) {
        kyb::set_error("kyb_ed25519_add_dev: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    hipLaunchKernelGGL(kyb::ed25519_add_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream, n,
                       (const uint32_t*)d_a, (const uint32_t*)d_b, (uint32_t*)d_out, (uint8_t*)d_status);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
int kyb_ed25519_add(size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* status) {
    if (n && (!a || !b || !out)) {
        kyb::set_error("kyb_ed25519_add: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    kyb::DeviceCtx* ctx;
    int rc = kyb::get_ctx(&ctx);
    if (rc) return rc;
    kyb::StageScope sc_(ctx);
    kyb::StageBuf d_a, d_b, d_o, d_st;
    rc = d_a.upload(a, n * 32);
    if (rc == KYB_OK) rc = d_b.upload(b, n * 32);
    if (rc == KYB_OK) rc = d_o.alloc(n * 32);
    if (rc == KYB_OK) rc = d_st.alloc(n);
    if (rc == KYB_OK) rc = kyb_ed25519_add_dev(n, d_a.p, d_b.p, d_o.p, d_st.p, sc_.stream());
    if (rc == KYB_OK) rc = d_o.download(out, n * 32);
    if (rc == KYB_OK && status) rc = d_st.download(status, n);
    return rc;
}
}
