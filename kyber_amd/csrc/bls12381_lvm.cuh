// BLS12-381 G1Elt.Mul / G2Elt.Mul (kilic/g1.go:110-116, kilic/g2.go) on the lane machine (lane_vm.cuh; programs from
// gen_lane_vm.py).  A batch of at least LVM_MIN elements runs as
//
//   0. (inputs that still need UnmarshalBinary's checks, or are compressed) bls12381_g*_unmarshal_kernel, the per-lane
//      code: ZCash rules, square root, subgroup test -> validated uncompressed affine points + the status bytes;
//   1. bls12381_lvm_prep_kernel, one lane per element: the scalar is split (GLV: base z^2, GLS: base |z| -- the long
//      divisions of bls12381.cuh) and recoded into regular signed odd digits (bytes: table index | sign << 7), the
//      coordinates become plain little-endian words per machine lane; elements that are rejected or at infinity are
//      marked and replaced by the generator, so that every lane of the machine computes something harmless;
//   2. bls12381_lvm_mul_kernel<PAIR>: the program -- table of odd multiples, 32 (16) windows of four doublings and two
//      (four) mixed additions, corrections, affine result;
//   3. bls12381_lvm_encode_kernel: canonical words -> the wire encoding the flags ask for; elements marked in step 1
//      or whose accumulator ended with Z = 0 (k P = infinity, or an addition met its exceptional case) are marked
//      "redo" and
//   4. the per-lane kernel of round 1 recomputes exactly those (its `only` mask) from the caller's original input.
//
// Large batches go through in rounds of 2^19 (G1) / 2^18 (G2) elements, so that the window tables (2.3 KB per G1 lane,
// 4.6 KB per G2 lane) stay bounded at 1.2 / 2.4 GB.
//
// Below the machine's threshold (round 4; lvm_mul and unmarshal_small decide, all of it behind the same entry points):
//   n <= 2^14 (G1) / 2^13 (G2)   four cooperating lanes per point out of LDS slots, ladder and subgroup rule
//                                (bls12381_g1coop.cuh): the chip is mostly empty, a call costs one ladder's latency;
//   2^14 < n <= 2^15, G1, checks  the r-torsion test and the multiplication in different workgroups (bls12381_g1split.hip);
//   otherwise                     one lane does it all (pairing_abi.cuh's kernels);
// and UnmarshalBinary of batches with two or more waves per SIMD in flight takes the per-lane kernels compiled on a
// two-wave register budget (bls12381_unm2.hip).
#pragma once
#include <stdlib.h>

#include <atomic>

#include "bls12381.cuh"
#include "bls12381_g1coop.cuh"
#include "context.h"
#include "lane_vm.cuh"
#include "lane_vm_bls12381.inc"

namespace kyb {
// the per-lane kernels (pairing_abi.cuh stamps them out later in the translation unit)
__global__ void bls12381_g1_unmarshal_kernel(size_t n, const uint8_t* __restrict__ pts, uint8_t* __restrict__ out, uint8_t* __restrict__ status,
                                             uint32_t flags);
__global__ void bls12381_g2_unmarshal_kernel(size_t n, const uint8_t* __restrict__ pts, uint8_t* __restrict__ out, uint8_t* __restrict__ status,
                                             uint32_t flags);

namespace bls {
constexpr int WS_LVM = 3;
constexpr int LVM_G1_DSTRIDE = 68, LVM_G2_DSTRIDE = 72;        // digit bytes per lane (2 x 34, 4 x 18)
constexpr int LVM_G1_NPOS = 33, LVM_G2_NPOS = 17;
static_assert(LVM_BLS12381_G1_MUL_NDIGITS == 2 * (LVM_G1_NPOS + 1) && LVM_BLS12381_G2_MUL_NDIGITS == 4 * (LVM_G2_NPOS + 1), "digit layout");

// x R -> x^-1 R1^2 on packed words: mont.cuh fp_inv (radix R1 = 2^390); the program multiplies by the constant that moves
// the result to the machine's radix (gen_lane_vm.py Curve.consts)
struct LvmInv {
    __device__ static void inv(uint32_t (&w)[12]) {
        fp x;
#pragma unroll
        for (int k = 0; k < 12; k++) x.v[k] = w[k];
        fp_inv(x, x);
#pragma unroll
        for (int k = 0; k < 12; k++) w[k] = x.v[k];
    }
};

// EXT: the program comes with the arguments (tools/lvm_microbench.py).  A separate instantiation: the product kernels
// assign their program from the __device__ arrays unconditionally, so that the compiler knows the address space
// (global loads it may leave in flight; through generic pointers every record fetch became a flat load followed by
// a full wait) and keeps the schedule walk in scalar registers.
template <bool PAIR, bool EXT = false>
__global__ __launch_bounds__(64, 2) void bls12381_lvm_mul_kernel(lvm::Args a) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[lvm::NL * Bls12381Lvm::N * lvm::LANES];
    if (!EXT) {
        a.prog = PAIR ? LVM_BLS12381_G2_MUL_PROG : LVM_BLS12381_G1_MUL_PROG;
        a.sched = reinterpret_cast<const lvm::Sched*>(PAIR ? LVM_BLS12381_G2_MUL_SCHED : LVM_BLS12381_G1_MUL_SCHED);
        a.nsched = PAIR ? LVM_BLS12381_G2_MUL_NSCHED : LVM_BLS12381_G1_MUL_NSCHED;
        a.consts = reinterpret_cast<const int32_t*>(PAIR ? LVM_BLS12381_G2_MUL_CONSTS : LVM_BLS12381_G1_MUL_CONSTS);
    }
    lvm::run<Bls12381Lvm, LvmInv, PAIR>(a, lds);
}

// One sub-scalar (NWS little-endian words, below 16^npos - 2) -> npos regular signed odd digits + the correction byte
// (gen_lane_vm.py digit_bytes: the value is made odd by adding 1 or 2; digit i = (((k >> 4 i) | 1) & 31) - 16, the top
// digit (k >> 4 (npos - 1)) | 1; flip: the sub-scalar multiplies the negative of its table variant).
template <int NWS>
KYB_HD void lvm_digits(uint8_t* dst, const uint32_t (&w)[NWS], int npos, bool flip) {
    uint32_t k[NWS + 1];
    const uint32_t odd = w[0] & 1u;
    uint64_t c = odd ? 2u : 1u;
#pragma unroll
    for (int i = 0; i < NWS; i++) {
        c += w[i];
        k[i] = (uint32_t)c;
        c >>= 32;
    }
    k[NWS] = (uint32_t)c;
    for (int i = 0; i < npos; i++) {
        const int bit = 4 * i, wi = bit >> 5, sh = bit & 31;
        uint32_t u = k[wi] >> sh;
        if (sh > 27 && wi + 1 <= NWS) u |= k[wi + 1] << (32 - sh);
        int d;
        if (i < npos - 1) d = (int)((u | 1u) & 31u) - 16;
        else d = (int)((u | 1u) & 15u);
        const bool neg = (d < 0) != flip;
        const int ad = d < 0 ? -d : d;
        dst[i] = (uint8_t)(((ad - 1) >> 1) | (neg ? 0x80 : 0));
    }
    dst[npos] = (uint8_t)((odd ? 8 : 0) | (flip ? 0 : 0x80));
}

// pts: uncompressed wire points (96 / 192 bytes, stride pt_stride, may be 0: one shared base).  st_in: the unmarshal
// kernel's status bytes when it ran (then pts is its output), or null: the flag / range rules of the uncompressed form
// are applied here (bls12381.cuh g*_decode_unc with validate = false -- the caller vouched for the rest).
template <bool G2>
__global__ __launch_bounds__(64, KYB_TU_WAVES) void bls12381_lvm_prep_kernel(size_t n, const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ pts,
                                                             size_t pt_stride, const uint8_t* __restrict__ st_in, size_t st_stride,
                                                             uint32_t* __restrict__ in,
                                                             uint8_t* __restrict__ digits, uint8_t* __restrict__ redo) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NC = G2 ? 4 : 2;  // 48-byte field elements of the point
    const uint8_t* p = pts + pt_stride * i;
    uint32_t w[NC][12];
#pragma unroll
    for (int j = 0; j < NC; j++) words_from_be<12>(w[j], p + 48 * j);
    int st;
    if (st_in) {
        st = st_in[i * st_stride];
    } else {  // the flag and range rules of g*_decode_unc (validate = false): compression and sort bits clear, an infinity
              // encoding all-zero otherwise, every coordinate below p
        const uint32_t top = w[0][11] >> 29;
        uint32_t any = 0;
        bool lt = true;
#pragma unroll
        for (int j = 0; j < NC; j++) {
            uint32_t v[12];
#pragma unroll
            for (int k = 0; k < 12; k++) v[k] = (j == 0 && k == 11) ? (w[j][k] & 0x1fffffffu) : w[j][k];
#pragma unroll
            for (int k = 0; k < 12; k++) any |= v[k];
            lt = lt & fp_words_lt_p<FC>(v);
        }
        st = ((top & 5u) || ((top & 2u) ? any != 0 : !lt)) ? ST_BAD_POINT : ST_OK;
    }
    const bool inf = (w[0][11] >> 30) & 1u;
    const bool flagged = st != ST_OK || inf;
    redo[i] = flagged ? 1 : 0;
    const size_t nl = G2 ? 2 * n : n;
    if constexpr (G2) {  // wire order x.c1, x.c0, y.c1, y.c0; lane 2i takes the c0 halves, lane 2i + 1 the c1 halves
#pragma unroll
        for (int k = 0; k < 12; k++) {
            in[(0 * nl + 2 * i) * 12 + k] = flagged ? LVM_BLS12381_G2_GEN[0][k] : w[1][k];
            in[(0 * nl + 2 * i + 1) * 12 + k] = flagged ? LVM_BLS12381_G2_GEN[1][k] : w[0][k];
            in[(1 * nl + 2 * i) * 12 + k] = flagged ? LVM_BLS12381_G2_GEN[2][k] : w[3][k];
            in[(1 * nl + 2 * i + 1) * 12 + k] = flagged ? LVM_BLS12381_G2_GEN[3][k] : w[2][k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 12; k++) {
            in[(0 * nl + i) * 12 + k] = flagged ? LVM_BLS12381_G1_GEN[0][k] : w[0][k];
            in[(1 * nl + i) * 12 + k] = flagged ? LVM_BLS12381_G1_GEN[1][k] : w[1][k];
        }
    }
    uint32_t k[8];
    scalar_from_be(k, scalars + 32 * i);
    if constexpr (G2) {
        uint32_t q1[8], q2[8], q3[8], a0[2], a1[2], a2[2];
        divmod_z<2>(q1, a0, k);
        divmod_z<2>(q2, a1, q1);
        divmod_z<2>(q3, a2, q2);  // q3 = a3 < 2^66
        uint8_t b[LVM_G2_DSTRIDE];
        const uint32_t s0[3] = {a0[0], a0[1], 0}, s1[3] = {a1[0], a1[1], 0}, s2[3] = {a2[0], a2[1], 0}, s3[3] = {q3[0], q3[1], q3[2]};
        lvm_digits<3>(b, s0, LVM_G2_NPOS, false);
        lvm_digits<3>(b + 18, s1, LVM_G2_NPOS, true);
        lvm_digits<3>(b + 36, s2, LVM_G2_NPOS, false);
        lvm_digits<3>(b + 54, s3, LVM_G2_NPOS, true);
        uint32_t* o0 = reinterpret_cast<uint32_t*>(digits + (2 * i) * LVM_G2_DSTRIDE);
        uint32_t* o1 = reinterpret_cast<uint32_t*>(digits + (2 * i + 1) * LVM_G2_DSTRIDE);
        for (int j = 0; j < LVM_G2_DSTRIDE / 4; j++) {
            const uint32_t v = b[4 * j] | (b[4 * j + 1] << 8) | (b[4 * j + 2] << 16) | ((uint32_t)b[4 * j + 3] << 24);
            o0[j] = v;
            o1[j] = v;
        }
    } else {
        uint32_t q[8], rem[4];
        divmod_z<4>(q, rem, k);  // by z^2: q < 2^129
        uint8_t b[LVM_G1_DSTRIDE];
        const uint32_t s0[5] = {rem[0], rem[1], rem[2], rem[3], 0}, s1[5] = {q[0], q[1], q[2], q[3], q[4]};
        lvm_digits<5>(b, s0, LVM_G1_NPOS, false);
        lvm_digits<5>(b + 34, s1, LVM_G1_NPOS, true);
        uint32_t* o0 = reinterpret_cast<uint32_t*>(digits + i * LVM_G1_DSTRIDE);
        for (int j = 0; j < LVM_G1_DSTRIDE / 4; j++) o0[j] = b[4 * j] | (b[4 * j + 1] << 8) | (b[4 * j + 2] << 16) | ((uint32_t)b[4 * j + 3] << 24);
    }
}

// canonical words of the affine result -> wire bytes; marks what the per-lane kernel has to redo
template <bool G2>
__global__ __launch_bounds__(64, KYB_TU_WAVES) void bls12381_lvm_encode_kernel(size_t n, const uint32_t* __restrict__ res, const uint8_t* __restrict__ zflag,
                                                               uint8_t* __restrict__ redo, uint8_t* __restrict__ out, uint8_t* __restrict__ status,
                                                               uint32_t flags) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool z = G2 ? (zflag[2 * i] & zflag[2 * i + 1]) : zflag[i];
    if (redo[i] || z) {
        redo[i] = 1;
        return;
    }
    const size_t nl = G2 ? 2 * n : n;
    if (status) status[i] = 0;
    if constexpr (G2) {
        uint32_t x0[12], x1[12], y0[12], y1[12];
#pragma unroll
        for (int k = 0; k < 12; k++) {
            x0[k] = res[(0 * nl + 2 * i) * 12 + k];
            x1[k] = res[(0 * nl + 2 * i + 1) * 12 + k];
            y0[k] = res[(1 * nl + 2 * i) * 12 + k];
            y1[k] = res[(1 * nl + 2 * i + 1) * 12 + k];
        }
        uint8_t* o = out + g2_out_size(flags) * i;
        if (flags & FLAG_UNCOMPRESSED_OUT) {
            words_to_be<12>(o, x1);
            words_to_be<12>(o + 48, x0);
            words_to_be<12>(o + 96, y1);
            words_to_be<12>(o + 144, y0);
        } else {
            uint32_t any = 0;
#pragma unroll
            for (int k = 0; k < 12; k++) any |= y1[k];
            const bool larger = any ? words_gt<12>(y1, FC::HALF) : words_gt<12>(y0, FC::HALF);
            x1[11] |= 0x80000000u | (larger ? 0x20000000u : 0u);
            words_to_be<12>(o, x1);
            words_to_be<12>(o + 48, x0);
        }
    } else {
        uint32_t x[12], y[12];
#pragma unroll
        for (int k = 0; k < 12; k++) {
            x[k] = res[(0 * nl + i) * 12 + k];
            y[k] = res[(1 * nl + i) * 12 + k];
        }
        uint8_t* o = out + g1_out_size(flags) * i;
        if (flags & FLAG_UNCOMPRESSED_OUT) {
            words_to_be<12>(o, x);
            words_to_be<12>(o + 48, y);
        } else {
            x[11] |= 0x80000000u | (words_gt<12>(y, FC::HALF) ? 0x20000000u : 0u);
            words_to_be<12>(o, x);
        }
    }
}

// Smallest batch the machine takes.  G2: a point is two lanes, so the machine has twice the waves of the per-lane
// kernel at any size and wins from the first full wave per CU on (measured 1.4x at 2^16 and 2^18).  G1: with one wave per
// SIMD (up to 64 lanes x 4 SIMDs x CUs elements) both formulations are bound by the latency of a lone wave and the
// per-lane kernel is slightly ahead (3.77 against 3.97 ms per 2^16); from two waves per SIMD the machine's 202
// registers let them overlap and it leads by 1.3x (profiles/r03_lvm_*).  KYB_LVM_MIN overrides both (A/B runs; a huge
// value sends everything to the per-lane kernels).
inline std::atomic<long long>& lvm_min_override() {  // atomic: the debug setter may race with calls on other threads
    static std::atomic<long long> v{[] {
        const char* e = getenv("KYB_LVM_MIN");
        return e ? (long long)strtoull(e, nullptr, 10) : -1ll;
    }()};
    return v;
}
inline size_t lvm_min_batch(bool g2, int num_cu) {
    const long long env = lvm_min_override().load(std::memory_order_relaxed);
    if (env >= 0) return (size_t)env;
    return g2 ? (size_t)1024 : (size_t)num_cu * 4 * 64 + 1;
}
inline size_t lvm_al(size_t x) { return (x + 255) & ~size_t(255); }

// Enqueues steps 0-3 for elements [0, n).  *only receives the device mask of the elements the per-lane kernel still has
// to compute (null: the batch is too small for the machine, compute everything there).
// *handled: the whole batch was done here (nothing left for the per-lane kernel): G1 batches of at most g1_coop_max()
// elements take the kernel on four cooperating lanes per point (bls12381_g1coop.cuh) -- the chip is mostly empty at such
// sizes and a call costs the latency of one ladder: 24 product levels per window instead of ~50 dependent
// multiplications.  KYB_G1_COOP_MAX overrides the threshold (0 = never).
inline size_t g1_coop_max(int num_cu) {
    static const long long env = [] {
        const char* e = getenv("KYB_G1_COOP_MAX");
        return e ? (long long)strtoull(e, nullptr, 10) : -1ll;
    }();
    if (env >= 0) return (size_t)env;
    return (size_t)num_cu * g1coop::GROUPS * 4;  // four workgroups of 16 points per CU: one wave per SIMD
}
inline size_t g2_coop_max(int num_cu) {  // the same for G2 (g2coop): 61 KB of slots per workgroup, two per CU
    static const long long env = [] {
        const char* e = getenv("KYB_G2_COOP_MAX");
        return e ? (long long)strtoull(e, nullptr, 10) : -1ll;
    }();
    if (env >= 0) return (size_t)env;
    return (size_t)num_cu * g2coop::GROUPS * 2;
}
// G1Elt.Mul with UnmarshalBinary's checks for a batch that leaves the chip half empty: bls12381_g1split.hip (a translation
// unit of its own -- its kernel wants two waves per SIMD, and the out-of-line field code takes the loosest register
// budget of the kernels that reach it).  st: n bytes of scratch on the device.  KYB_G1_SPLIT=0: never (A/B).
void launch_g1_mul_split(size_t n, const uint8_t* d_scalars, const uint8_t* d_points, uint8_t* d_out, uint8_t* d_st, uint8_t* d_status,
                         uint32_t flags, hipStream_t st, uint32_t* d_tabs);
// UnmarshalBinary of a batch with at least two waves per SIMD in flight: bls12381_unm2.hip (the per-lane kernels on a
// two-wave register budget).  KYB_UNM_W2=0: always the kernels of this unit (A/B).
void launch_unmarshal_w2(bool g2, size_t n, const uint8_t* d_points, uint8_t* d_out, uint8_t* d_status, uint32_t flags, hipStream_t st);
inline bool unmarshal_w2(bool g2, size_t n, int num_cu) {
    static const bool on = [] {
        const char* e = getenv("KYB_UNM_W2");
        return !(e && e[0] == '0');
    }();
    // measured crossovers: G1 from two waves per SIMD (2^17: +10 %, 2^20: +22 %), G2 level up to 2^18 and +16 % at 2^20
    return on && n >= (size_t)num_cu * 4 * 64 * (g2 ? 8 : 2);
}
inline bool g1_split_enabled() {
    static const bool on = [] {
        const char* e = getenv("KYB_G1_SPLIT");
        return !(e && e[0] == '0');
    }();
    return on;
}
// UnmarshalBinary of a small batch on cooperating lanes (the subgroup rule is what a lone lane spends its time on)
inline int unmarshal_small(bool g2, size_t n, const uint8_t* d_points, uint8_t* d_out, uint8_t* d_status, uint32_t flags, hipStream_t st,
                           bool* handled) {
    *handled = false;
    DeviceCtx* ctx;
    int rc = get_ctx(&ctx);
    if (rc) return rc;
    if (n > (g2 ? g2_coop_max(ctx->num_cu) : g1_coop_max(ctx->num_cu))) {
        if (unmarshal_w2(g2, n, ctx->num_cu)) {
            launch_unmarshal_w2(g2, n, d_points, d_out, d_status, flags, st);
            KYB_HIP_CHECK(hipGetLastError());
            *handled = true;
        }
        return KYB_OK;
    }
    const unsigned grid = (unsigned)((n + 15) / 16);
    if (g2) hipLaunchKernelGGL(g2coop::bls12381_g2_unmarshal_coop_kernel, dim3(grid), dim3(64), 0, st, n, d_points, d_out, d_status, flags);
    else hipLaunchKernelGGL(g1coop::bls12381_g1_unmarshal_coop_kernel, dim3(grid), dim3(64), 0, st, n, d_points, d_out, d_status, flags);
    KYB_HIP_CHECK(hipGetLastError());
    *handled = true;
    return KYB_OK;
}
inline int lvm_mul(bool g2, size_t n, const uint8_t* d_scalars, const uint8_t* d_points, size_t point_stride, uint8_t* d_out,
                   uint8_t* d_status, uint32_t flags, hipStream_t st, const uint8_t** only, bool* handled = nullptr,
                   int32_t* trace = nullptr) {
    *only = nullptr;
    if (handled) *handled = false;
    DeviceCtx* ctx;
    int rc = get_ctx(&ctx);
    if (rc) return rc;
    if (!g2 && handled && !trace && point_stride && n <= g1_coop_max(ctx->num_cu)) {
        hipLaunchKernelGGL(g1coop::bls12381_g1_mul_coop_kernel, dim3((unsigned)((n + g1coop::GROUPS - 1) / g1coop::GROUPS)), dim3(64), 0, st, n,
                           d_scalars, d_points, point_stride, d_out, d_status, flags);
        KYB_HIP_CHECK(hipGetLastError());
        *handled = true;
        return KYB_OK;
    }
    if (g2 && handled && !trace && point_stride && n <= g2_coop_max(ctx->num_cu)) {
        hipLaunchKernelGGL(g2coop::bls12381_g2_mul_coop_kernel, dim3((unsigned)((n + g2coop::GROUPS - 1) / g2coop::GROUPS)), dim3(64), 0, st, n,
                           d_scalars, d_points, point_stride, d_out, d_status, flags);
        KYB_HIP_CHECK(hipGetLastError());
        *handled = true;
        return KYB_OK;
    }
    if (n < lvm_min_batch(g2, ctx->num_cu)) {
        if (!g2 && handled && !trace && point_stride && !(flags & KYB_F_TRUSTED(0)) && n <= (size_t)ctx->num_cu * 2 * 64 && g1_split_enabled()) {
            uint8_t* stt;  // the test's verdicts; the caller holds enq_mu until the merge kernel is enqueued
            if ((rc = ctx_workspace(ctx, WS_LVM, st, lvm_al(n), (void**)&stt))) return rc;
            void* tabs;  // the ladder's per-lane table slab (bls12381.cuh g1_mul_glv_lz<true>)
            if ((rc = ctx_workspace(ctx, WS_G1TAB, st, n * G1_TAB_WORDS * sizeof(uint32_t), &tabs))) return rc;
            launch_g1_mul_split(n, d_scalars, d_points, d_out, stt, d_status, flags, st, (uint32_t*)tabs);
            KYB_HIP_CHECK(hipGetLastError());
            *handled = true;
        }
        return KYB_OK;
    }
    std::lock_guard<std::recursive_mutex> lk(ctx->enq_mu);
    // Elements per round of (unmarshal, prep, mul, encode): the window tables are 2.3 KB per G1 lane and 2 x 4.6 KB per G2
    // element, i.e. 1.2 / 2.4 GB of the (WS_LVM, stream) workspace at these sizes -- small change on a 288 GB device, and
    // every round boundary drains the chip: 2^17 / 2^16 elements per round measured 4.6-8.5 % slower at 2^18 - 2^20 elements
    // (profiles/r04_lvm_chunks.json).
#ifndef KYB_LVM_G1_CHUNK_LOG
#define KYB_LVM_G1_CHUNK_LOG 19
#endif
#ifndef KYB_LVM_G2_CHUNK_LOG
#define KYB_LVM_G2_CHUNK_LOG 18
#endif
    const size_t chunk = g2 ? (size_t(1) << KYB_LVM_G2_CHUNK_LOG) : (size_t(1) << KYB_LVM_G1_CHUNK_LOG);
    const size_t cn = n < chunk ? n : chunk, cl = g2 ? 2 * cn : cn;
    const size_t unc = g2 ? 192 : 96, dstr = g2 ? LVM_G2_DSTRIDE : LVM_G1_DSTRIDE;
    const uint32_t ncoord = g2 ? LVM_BLS12381_G2_MUL_NCOORD : LVM_BLS12381_G1_MUL_NCOORD, nentry = LVM_BLS12381_G1_MUL_NENTRY;
    const bool need_unmarshal = !(flags & KYB_F_UNCOMPRESSED) || !(flags & KYB_F_TRUSTED(0));
    const size_t b_redo = lvm_al(n), b_unm = need_unmarshal ? lvm_al(cn * unc) : 0, b_st = need_unmarshal ? lvm_al(cn) : 0;
    const size_t b_io = lvm_al(2 * cl * 48), b_dig = lvm_al(cl * dstr), b_z = lvm_al(cl);
    const size_t b_tab = lvm_al(cl * (size_t)nentry * ncoord * lvm::TAB_WORDS * 4);
    uint8_t* base;
    rc = ctx_workspace(ctx, WS_LVM, st, b_redo + b_unm + b_st + 2 * b_io + b_dig + b_z + b_tab, (void**)&base);
    if (rc) return rc;
    uint8_t* redo = base;
    uint8_t* unm = redo + b_redo;
    uint8_t* sta = unm + b_unm;
    uint32_t* in = (uint32_t*)(sta + b_st);
    uint32_t* res = (uint32_t*)((uint8_t*)in + b_io);
    uint8_t* dig = (uint8_t*)res + b_io;
    uint8_t* zf = dig + b_dig;
    int32_t* tab = (int32_t*)(zf + b_z);
    const size_t wire = g2 ? g2_wire_size(flags) : g1_wire_size(flags), osz = g2 ? g2_out_size(flags) : g1_out_size(flags);
    for (size_t lo = 0; lo < n; lo += cn) {
        const size_t m = n - lo < cn ? n - lo : cn, ml = g2 ? 2 * m : m;
        const uint8_t* pts = d_points + (point_stride ? wire * lo : 0);
        size_t pstride = point_stride ? wire : 0;
        const uint8_t* stp = nullptr;
        const unsigned g = (unsigned)((m + 63) / 64);
        if (need_unmarshal) {  // UnmarshalBinary's checks, per lane, into validated uncompressed points
            const size_t mu = point_stride ? m : 1;
            const uint32_t uf = (flags & (KYB_F_UNCOMPRESSED | KYB_F_TRUSTED(0))) | KYB_F_UNCOMPRESSED_OUT;
            if (unmarshal_w2(g2, mu, ctx->num_cu)) launch_unmarshal_w2(g2, mu, pts, unm, sta, uf, st);
            else if (g2) hipLaunchKernelGGL(bls12381_g2_unmarshal_kernel, dim3((unsigned)((mu + 63) / 64)), dim3(64), 0, st, mu, pts, unm, sta, uf);
            else hipLaunchKernelGGL(bls12381_g1_unmarshal_kernel, dim3((unsigned)((mu + 63) / 64)), dim3(64), 0, st, mu, pts, unm, sta, uf);
            pts = unm;
            pstride = point_stride ? unc : 0;
            stp = sta;
        }
        const size_t sstride = point_stride ? 1 : 0;  // a shared base has ONE status byte
        if (g2)
            hipLaunchKernelGGL(bls12381_lvm_prep_kernel<true>, dim3(g), dim3(64), 0, st, m, d_scalars + 32 * lo, pts, pstride, stp, sstride, in, dig,
                               redo + lo);
        else
            hipLaunchKernelGGL(bls12381_lvm_prep_kernel<false>, dim3(g), dim3(64), 0, st, m, d_scalars + 32 * lo, pts, pstride, stp, sstride, in, dig,
                               redo + lo);
        lvm::Args a{};
        a.in = in;
        a.out = res;
        a.digits = dig;
        a.dstride = (uint32_t)dstr;
        a.table = tab;
        a.nentry = nentry;
        a.ncoord = ncoord;
        a.zflag = zf;
        a.nlanes = ml;
        a.trace = trace;
        const unsigned gw = (unsigned)((ml + 63) / 64);
        if (g2) hipLaunchKernelGGL(bls12381_lvm_mul_kernel<true>, dim3(gw), dim3(64), 0, st, a);
        else hipLaunchKernelGGL(bls12381_lvm_mul_kernel<false>, dim3(gw), dim3(64), 0, st, a);
        if (g2)
            hipLaunchKernelGGL(bls12381_lvm_encode_kernel<true>, dim3(g), dim3(64), 0, st, m, res, zf, redo + lo, d_out + osz * lo,
                               d_status ? d_status + lo : nullptr, flags);
        else
            hipLaunchKernelGGL(bls12381_lvm_encode_kernel<false>, dim3(g), dim3(64), 0, st, m, res, zf, redo + lo, d_out + osz * lo,
                               d_status ? d_status + lo : nullptr, flags);
        KYB_HIP_CHECK(hipGetLastError());
    }
    *only = redo;
    return KYB_OK;
}

}  // namespace bls
}  // namespace kyb
