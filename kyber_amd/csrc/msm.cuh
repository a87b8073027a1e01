// Multi-scalar multiplication  sum_i k_i * P_i -> one point  (Pippenger bucket method) for any of
// the engine's groups, written once over a small curve adapter.
//
// Replaces the reference's sequential N x (Mul + Add) at its MSM-shaped call sites
// (share/poly.go:340-348 PubPoly.Eval, :449-476 RecoverCommit, sign/bdn/bdn.go:126-161
// AggregateSignatures, sign/bdn/mask.go:57-61); the reference has no MSM function, and because
// encodings are canonical any correct summation order gives the same bytes.
//
// Pipeline (all on one stream, no host synchronisation):
//   1 decode    : lane i decodes point i to affine form in HBM (AoS, one gather unit per point; stored through LDS as
//                 whole lines) and recodes scalar i into signed c-bit digits, window-major.  Adapters with a light decode
//                 (vouched-for uncompressed points) have a kernel of their own for it.
//   2 sort      : counting sort of point indices (sign in bit 31) by (window, bucket) -- large plans in two passes whose
//                 stores are runs (coarse bins of 256 buckets, then one workgroup per bin), small ones in one pass with a
//                 counter per bucket in LDS; either way offs[] = the first entry of every bucket
//   3 accumulate: one lane per bucket piece (at most twice the mean bucket length; long buckets from skewed digits are
//                 split) adds its points with mixed additions -- in XYZZ form on the Weierstrass curves.  A bucket's only
//                 piece is stored as the bucket; buckets of several pieces are joined afterwards (four lanes per bucket on
//                 the Weierstrass curves; a bucket of many pieces by workgroups of the cooperative fold, in two launches)
//   4 reduce    : per window, chunks of buckets -> sum_b b * B_b by running sums; Weierstrass curves: FOUR lanes per
//                 chunk cooperating through LDS slots (reduce_coop_kernel), others: one lane per chunk
//   5 fold+final: chunk partials are folded 32 / 64 at a time as trees, then the window sums are shifted by 2^(c w)
//                 (3 - 4 lanes per point share the products of a doubling), added and encoded.  The Weierstrass curves
//                 (the split tail): the chunks keep their lo * run apart, the fold tree yields the bit sums D_k, and every
//                 term 2^(c w) W_w, 2^(c w + tz + k) D_k(w) has a doubling chain of its own -- limb-per-lane rows
//                 (rowfp.cuh) on BLS12-381 G1, three cooperating lanes elsewhere.
//                 If any input failed to decode the output is all-zero bytes.
// Sorting instead of atomics on ~150-byte points: the only atomics are 32-bit counters.
//
// Adapter A: typename Aff (decoded input), Acc (accumulator); WIRE (input bytes), OUT (output bytes);
//   decode(Aff&, const uint8_t*, flags) -> status, wire_size(flags), scalar_words(uint32_t(&)[8], const uint8_t*),
//   identity(Acc&), madd(Acc&, const Aff&, bool neg), add(Acc&, const Acc&, const Acc&),
//   dbl(Acc&, const Acc&), encode(uint8_t*, const Acc&).
#pragma once
#include <stdlib.h>
#include <vector>
#include "context.h"

namespace kyb {
namespace msm {

struct Plan {
    size_t n;
    int c;        // window bits
    int nwin;     // number of windows (incl. the carry window)
    int nb;       // buckets per window = 2^(c-1)
    int chunk;    // buckets per reduce lane
    int nchunks;  // chunks per window
    uint32_t flags;  // the call's KYB_F_* flags (input format / trusted operands), read by the adapter's decode
    int bits;        // scalar bits that count (KYB_F_SCALAR_BITS: bdn's 128-bit coefficients); higher bits are ignored
};

inline Plan make_plan(size_t n, int scalar_bits = 256, int cmax = 16) {
    Plan p;
    p.n = n;
    p.flags = 0;
    p.bits = scalar_bits;
    int lg = 0;
    while ((size_t(1) << (lg + 1)) <= n) lg++;
    int c = lg - 3;
    if (c < 3) c = 3;  // at most 86 windows: final_kernel gives each window up to 4 lanes of its 512
    if (c > cmax) c = cmax;
    p.c = c;
    p.nwin = (scalar_bits + c) / c;  // ceil((bits + 1) / c): the scalar bits + the recoding carry
    p.nb = 1 << (c - 1);
    // buckets per reduce lane: the running-sum chain of a lane is latency-bound (2 dependent additions per bucket, and
    // a lone wave already saturates its SIMD's issue rate), so take the shortest chains that still leave every wave
    // a SIMD of its own: at most 1024 waves = 65536 lanes, between 8 and 64 buckets each
    static const int chunk_min = [] {  // KYB_MSM_CHUNK: buckets per reduce chain at least (experiments; 8)
        const char* e = getenv("KYB_MSM_CHUNK");
        const int v = e ? atoi(e) : 0;
        return v >= 2 && v <= 64 && (v & (v - 1)) == 0 ? v : 8;
    }();
    int chunk = chunk_min;
    while (chunk < 64 && (size_t)p.nwin * (p.nb / chunk) > 65536) chunk <<= 1;
    p.chunk = p.nb < chunk ? p.nb : chunk;
    p.nchunks = p.nb / p.chunk;
    return p;
}

// signed digit w of k (c-bit windows, digits in [-2^(c-1), 2^(c-1)]): digit = raw_w + carry_in(w) - (carry_out << c),
// carry_in(w) = 1 iff the lower part, recoded, overflowed.  Sequential recoding is cheap (<= 129 steps), so each lane
// recodes its whole scalar once, handing every digit to `emit(w, d)` as it is produced.
template <class Emit>
__device__ __forceinline__ void recode_each(const uint32_t (&k)[8], int c, int nwin, int bits, Emit emit) {
    int carry = 0;
    const int half = 1 << (c - 1);
#pragma unroll 1
    for (int w = 0; w < nwin; w++) {
        const int bit = w * c;
        uint32_t raw = 0;
        if (bit < bits) {
            const int idx = bit >> 5, sh = bit & 31;
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {  // k[idx], k[idx + 1] by selects: a dynamic index would put k in scratch
                lo = j == idx ? k[j] : lo;
                hi = j == idx + 1 ? k[j] : hi;
            }
            raw = lo >> sh;
            if (sh + c > 32) raw |= hi << ((32 - sh) & 31);
            raw &= (1u << (bits - bit < c ? bits - bit : c)) - 1;  // the window, clipped at the last bit that counts
        }
        int d = (int)raw + carry;
        carry = d > half ? 1 : 0;
        d -= carry << c;
        emit(w, d);
    }
}

// Split<A>: an adapter may split every input (k, P) into SPLIT pairs (k_h, P_h) with shorter scalars (SPLIT_BITS bits)
// through an endomorphism (BLS12-381 G1: k = k1 z^2 + k0, z^2 P = (beta x, -y)).  The pipeline then runs over
// SPLIT * n points with proportionally fewer windows -- and, what matters, a proportionally shorter doubling chain in
// its serial tail.
template <class A, class = void>
struct Split {
    static constexpr int value = 1, bits = 256, cmax = 16;
};
template <class A>
struct Split<A, decltype((void)A::SPLIT)> {
    static constexpr int value = A::SPLIT, bits = A::SPLIT_BITS, cmax = 16;
};

// Effective<A>: an adapter may replace (k, P) by an equivalent pair before the digits are cut -- Ed25519 maps a scalar
// the reference's radix-16 recoding would mangle (top digit above 8, dropped by the table select: ge.go:374-390,
// 419-435) to the integer the reference actually multiplies by, negating the point when that integer is negative, so
// that the sum equals N x Mul + N x Add for EVERY 32-byte scalar, not only the reduced ones its callers produce.
template <class A, class = void>
struct Effective {
    template <class Aff>
    __device__ static void apply(uint32_t (&)[8], Aff&, int) {}
};
template <class A>
struct Effective<A, decltype((void)A::HAS_EFFECTIVE)> {
    template <class Aff>
    __device__ static void apply(uint32_t (&k)[8], Aff& a, int bits) { A::effective(k, a, bits); }
};

// Register budget of the decode kernel in waves per SIMD: it is bound by its histogram atomics and digit stores, so
// co-resident waves pay (two for the Weierstrass adapters: 1.27 -> 0.98 ms; an adapter may ask for more).
template <class A, class = void>
struct DecodeWaves {
    static constexpr int value = 2;
};
template <class A>
struct DecodeWaves<A, decltype((void)A::DECODE_WAVES)> {
    static constexpr int value = A::DECODE_WAVES;
};

// LightDecode<A>: the adapter has a decode of its own for vouched-for points in their uncompressed form
// (decode_split_light, LIGHT_DECODE_WAVES): decode_kernel<A, true>
template <class A, class = void>
struct LightDecode {
    static constexpr bool value = false;
    static constexpr int waves = 2;
};
template <class A>
struct LightDecode<A, decltype((void)A::LIGHT_DECODE_WAVES)> {
    static constexpr bool value = true;
    static constexpr int waves = A::LIGHT_DECODE_WAVES;
};
template <class A, bool LIGHT = false>
__global__ __launch_bounds__(64, LIGHT ? LightDecode<A>::waves : DecodeWaves<A>::value) void decode_kernel(Plan p, size_t n_in, const uint8_t* __restrict__ scalars,
                                                    const uint8_t* __restrict__ points,
                                                    typename A::Aff* __restrict__ aff, int32_t* __restrict__ digits,
                                                    uint8_t* __restrict__ status,
                                                    uint32_t* __restrict__ bad) {
    constexpr int S = Split<A>::value;
    // the decoded points leave through LDS: a lane's Aff is ~100 bytes, and stored from its registers every dword of the
    // wave's 64 lands in a line of its own (50 stores x 64 lines per wave for BLS12-381 G1's two halves); staged, a wave
    // stores whole lines (222 -> 210 us per 2^20 points: the smaller part of what the light kernel gained)
    constexpr int W = (int)(sizeof(typename A::Aff) / 4);
    static_assert(sizeof(typename A::Aff) % 4 == 0, "Aff is staged word by word");
    __shared__ uint32_t stage[64 * W];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n_in;
    typename A::Aff a[S];
    uint32_t k[S][8];
    int st = 0;
    if (live) {
        if constexpr (LIGHT) {
            st = A::decode_split_light(a, k, points + A::wire_size(p.flags) * i, scalars + 32 * i);
        } else if constexpr (S == 1) {
            st = A::decode(a[0], points + A::wire_size(p.flags) * i, p.flags);
            A::scalar_words(k[0], scalars + 32 * i);
            Effective<A>::apply(k[0], a[0], p.bits);
        } else {
            st = A::decode_split(a, k, points + A::wire_size(p.flags) * i, scalars + 32 * i, p.flags);
        }
        if (status) status[i] = (uint8_t)st;
        if (st) atomicAdd(bad, 1u);
    }
    const size_t i0 = (size_t)blockIdx.x * blockDim.x;
    const size_t nvalid = i0 < n_in ? (n_in - i0 < 64 ? n_in - i0 : 64) : 0;
#pragma unroll
    for (int h = 0; h < S; h++) {
        const size_t e = (size_t)h * n_in + i;
        {
            uint32_t words[W];
            __builtin_memcpy(words, &a[h], sizeof(typename A::Aff));
#pragma unroll
            for (int j = 0; j < W; j++) stage[threadIdx.x * W + j] = words[j];
        }
        __syncthreads();
        uint32_t* dst = reinterpret_cast<uint32_t*>(aff + (size_t)h * n_in + i0);
#pragma unroll
        for (int j = 0; j < W; j++) {
            const size_t idx = (size_t)j * 64 + threadIdx.x;
            if (idx < nvalid * W) dst[idx] = stage[idx];
        }
        __syncthreads();
        // the digits go straight to memory as the recoding produces them (a digit array indexed by the window would
        // live in scratch: 516 B per lane and half); histogrammed per (window, tile) out of LDS: hist_lds_kernel
        if (live) recode_each(k[h], p.c, p.nwin, p.bits, [&](int w, int d) { digits[(size_t)w * p.n + e] = st ? 0 : d; });
    }
}

// exclusive scan of hist[0..m) into offs[0..m] in three launches: per-tile sums, a scan of the (<= a few hundred)
// tile sums, and a per-tile rescan that adds the tile's offset.  All global accesses are coalesced or 64-B vectors.
constexpr int SCAN_T = 256, SCAN_E = 16, SCAN_TILE = SCAN_T * SCAN_E;
static __global__ __launch_bounds__(SCAN_T) void scan_tilesum_kernel(const uint32_t* __restrict__ hist,
                                                                     uint32_t* __restrict__ tile, size_t m) {
    __shared__ uint32_t red[SCAN_T];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_E; e++) {
        const size_t j = base + (size_t)e * SCAN_T + threadIdx.x;
        if (j < m) s += hist[j];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = SCAN_T / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile[blockIdx.x] = red[0];
}
// in-place exclusive scan of tile[0..ntiles) (+ the total in tile[ntiles]) by ONE workgroup: every thread sums a
// contiguous chunk, the chunk sums are scanned through LDS, the chunks are rewritten.  (A single lane walking the
// array was fine for 64 tile sums; the LDS-staged sort scans nwin x buckets x tiles counters = thousands of tiles.)
static __global__ __launch_bounds__(SCAN_T) void scan_tiles_kernel(uint32_t* __restrict__ tile, size_t ntiles) {
    __shared__ uint32_t sh[SCAN_T];
    const size_t per = (ntiles + SCAN_T - 1) / SCAN_T, lo = per * threadIdx.x, hi = lo + per < ntiles ? lo + per : ntiles;
    uint32_t s = 0;
    for (size_t t = lo; t < hi; t++) s += tile[t];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < SCAN_T; off <<= 1) {
        const uint32_t add = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - s;
    for (size_t t = lo; t < hi; t++) {
        const uint32_t v = tile[t];
        tile[t] = run;
        run += v;
    }
    if (threadIdx.x == SCAN_T - 1) tile[ntiles] = sh[SCAN_T - 1];
}
static __global__ __launch_bounds__(SCAN_T) void scan_apply_kernel(const uint32_t* __restrict__ hist,
                                                                   const uint32_t* __restrict__ tile,
                                                                   uint32_t* __restrict__ offs, size_t m) {
    __shared__ uint32_t sh[SCAN_T];
    const size_t lo = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_E;
    uint32_t v[SCAN_E];
    uint32_t s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_E; e++) {
        v[e] = lo + e < m ? hist[lo + e] : 0u;
        s += v[e];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < SCAN_T; off <<= 1) {  // Hillis-Steele inclusive scan of the thread sums
        const uint32_t add = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = tile[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int e = 0; e < SCAN_E; e++) {
        if (lo + e < m) offs[lo + e] = run;
        run += v[e];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) offs[m] = tile[gridDim.x];
}
static inline void launch_scan(const uint32_t* hist, uint32_t* offs, size_t m, uint32_t* tile, hipStream_t st) {
    const unsigned ntiles = (unsigned)((m + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(scan_tilesum_kernel, dim3(ntiles), dim3(SCAN_T), 0, st, hist, tile, m);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(SCAN_T), 0, st, tile, (size_t)ntiles);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(ntiles), dim3(SCAN_T), 0, st, hist, (const uint32_t*)tile, offs, m);
}

// Counting sort of the (point, window) digits by bucket, staged in LDS -- no global atomics.  A workgroup owns one
// (window, tile of points) pair and a counter per bucket of that window in LDS (2^15 buckets x 4 B = 128 KB of the
// CU's 160 KB):
//   hist_lds_kernel     counts its tile with ds_add_u32 and stores the counters tile-major (hist2[(w T + tile) nb + b]:
//                       whole lines -- rounds 1-5 stored them bucket-major, one 4-byte store per 128-byte line, and scanned
//                       all nwin x nb x T of them: item 58 of DESIGN.md section 5)
//   tile_scan_kernel    one lane per (window, bucket) walks its T counters (lanes side by side: coalesced), leaves in each
//                       the number of the bucket's points in earlier tiles, and stores the bucket's total
//   launch_scan         of the nwin x nb totals: offs[b], what the rest of the pipeline indexes buckets by
//   scatter_lds_kernel  loads offs[b] + its tile's counter -- for every (bucket, tile) the first output slot of the
//                       tile's points of that bucket -- into the same LDS array and every point takes its slot with one
//                       returning ds_add (rank within the bucket = arrival order, any order is a valid sort).
// One workgroup per CU at a time (128 KB of LDS), so the grid is cut to a whole round of the chip (sort_tiles below).
// (Round 1 paid one global atomicAdd per digit in the decode kernel and another in the scatter: 2 x 16.8 M for
// 2^20 BLS12-381 G1 points; with 2^15 buckets per window two lanes of a wave rarely meet in a bucket, so wave-level
// ballot aggregation would save nothing on top of the LDS counters.)
constexpr int HIST_T = 1024;
#ifndef KYB_MSM_SORT_U
#define KYB_MSM_SORT_U 8
#endif
constexpr int SORT_U = KYB_MSM_SORT_U;  // digits in flight per thread in the two sort kernels
constexpr int HIST_MAX_NB = 1 << 15;
// 128 KB of static LDS: gfx950's 160 KB per CU, nothing smaller (the Makefile's ARCH is overridable; this says why not)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "msm.cuh: the LDS-staged counting sort needs gfx950's 160 KB of LDS per workgroup (HIST_MAX_NB counters = 128 KB)"
#endif
static __global__ __launch_bounds__(HIST_T) void hist_lds_kernel(Plan p, int tiles, const int32_t* __restrict__ digits,
                                                                 uint32_t* __restrict__ hist2) {
    __shared__ uint32_t h[HIST_MAX_NB];
    const int w = blockIdx.x % p.nwin, tile = blockIdx.x / p.nwin;  // as in scatter_lds_kernel
    for (int b = threadIdx.x; b < p.nb; b += HIST_T) h[b] = 0;
    __syncthreads();
    const size_t per = (p.n + tiles - 1) / tiles, lo = per * tile, hi = lo + per < p.n ? lo + per : p.n;
    const int32_t* dw = digits + (size_t)w * p.n;
    // SORT_U digits per thread and trip, loaded together (one digit per trip waits out a global load before each LDS
    // atomic).  Measured: hist 143 -> 125 us, scatter 283 -> 271 us, the MSM unchanged within noise for U = 1 / 4 / 8 / 16
    // (profiles/r05_msm_sort_batching_ab.jsonl) -- the two kernels are bound by the LDS atomics on 2^15 random counters
    // (16.8 M of them in 125 us: one lane-atomic per four cycles and CU), not by the loads
    for (size_t i0 = lo + threadIdx.x; i0 < hi; i0 += (size_t)HIST_T * SORT_U) {
        int d[SORT_U];
#pragma unroll
        for (int u = 0; u < SORT_U; u++) {
            const size_t i = i0 + (size_t)u * HIST_T;
            d[u] = i < hi ? dw[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < SORT_U; u++)
            if (d[u]) atomicAdd(&h[(d[u] < 0 ? -d[u] : d[u]) - 1], 1u);
    }
    __syncthreads();
    uint32_t* row = hist2 + ((size_t)w * tiles + tile) * p.nb;
    for (int b = threadIdx.x; b < p.nb; b += HIST_T) row[b] = h[b];
}
static __global__ __launch_bounds__(256) void tile_scan_kernel(size_t nbk, int nb, int tiles, uint32_t* __restrict__ hist2,
                                                               uint32_t* __restrict__ total) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nbk) return;
    const size_t w = g / (size_t)nb, b = g % (size_t)nb;
    uint32_t* col = hist2 + w * tiles * (size_t)nb + b;
    uint32_t run = 0;
    for (int t = 0; t < tiles; t++) {
        const uint32_t v = col[(size_t)t * nb];
        col[(size_t)t * nb] = run;
        run += v;
    }
    total[g] = run;
}
static __global__ __launch_bounds__(HIST_T) void scatter_lds_kernel(Plan p, int tiles, const int32_t* __restrict__ digits,
                                                                    const uint32_t* __restrict__ before,
                                                                    const uint32_t* __restrict__ offs,
                                                                    uint32_t* __restrict__ sorted, int xcd_major) {
    __shared__ uint32_t cur[HIST_MAX_NB];
    // window-minor: workgroups go round-robin over the 8 XCDs, so with 8 windows (the 2^20-point BLS12-381 G1 MSM) every
    // window's slice of `sorted` is written through ONE L2, where the 4-byte stores of a line can meet (by itself no
    // different, 223 against 227 us; it is what lets a scatter in sweeps gain anything: profiles/r06_msm_scatter_sweeps.txt)
    const int w = xcd_major ? blockIdx.x % p.nwin : blockIdx.x / tiles, tile = xcd_major ? blockIdx.x / p.nwin : blockIdx.x % tiles;
    const uint32_t* row = before + ((size_t)w * tiles + tile) * p.nb;
    const uint32_t* first = offs + (size_t)w * p.nb;
    for (int b = threadIdx.x; b < p.nb; b += HIST_T) cur[b] = first[b] + row[b];
    __syncthreads();
    const size_t per = (p.n + tiles - 1) / tiles, lo = per * tile, hi = lo + per < p.n ? lo + per : p.n;
    const int32_t* dw = digits + (size_t)w * p.n;
    for (size_t i0 = lo + threadIdx.x; i0 < hi; i0 += (size_t)HIST_T * SORT_U) {  // batched like the histogram's loop
        int d[SORT_U];
        uint32_t pos[SORT_U];
#pragma unroll
        for (int u = 0; u < SORT_U; u++) {
            const size_t i = i0 + (size_t)u * HIST_T;
            d[u] = i < hi ? dw[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < SORT_U; u++) pos[u] = d[u] ? atomicAdd(&cur[(d[u] < 0 ? -d[u] : d[u]) - 1], 1u) : 0u;
#pragma unroll
        for (int u = 0; u < SORT_U; u++)
            if (d[u]) sorted[pos[u]] = (uint32_t)(i0 + (size_t)u * HIST_T) | (d[u] < 0 ? 0x80000000u : 0u);
    }
}
// ---- The sort in two passes (round 6; plans of at least 2^13 buckets per window and at most 2^23 points) ----------------
// scatter_lds_kernel above stores every sorted entry on its own: 16.8 M four-byte stores into 16.8 M different lines for a
// 2^20-point BLS12-381 G1 MSM -- 220 us at the rate the L2 channels take partial-line writes, whatever order the
// workgroups run in and however many sweeps they make (profiles/r06_msm_scatter_sweeps.txt).  Two passes whose stores are
// RUNS instead:
//   1  by coarse bin (bucket / 256; <= 128 bins per window): coarse_hist_kernel counts every (window, bin, tile of 2^13
//      entries), one scan of the counters gives each (bin, tile) its first slot, coarse_scatter_kernel orders its tile by bin
//      in LDS and writes each bin's run of the tile as one piece -- entry = index | sign << 23 | (bucket mod 256) << 24;
//   2  fine_sort_kernel, one workgroup per (window, bin): counts the 256 buckets of its bin, writes their offsets (the
//      pipeline's offs[]), orders the bin's ~2^14 entries in LDS and writes them as ONE run of `sorted` (a bin too long
//      for LDS -- skewed digits -- is scattered directly).
#ifndef KYB_MSM_P2_T1
#define KYB_MSM_P2_T1 8192
#endif
#ifndef KYB_MSM_P2_LMAX
#define KYB_MSM_P2_LMAX 18432
#endif
// (tile and staging sizes that leave two workgroups per CU: fine_sort 76 -> 65 us, coarse_scatter 45 -> 38 us against 2^14 / 24 576;
// a bin of the halves' top window where the density doubles goes the direct way)
constexpr int P2_GIANT = 131072, P2_GS = 64;  // giant bins (giant_*_kernel below): entries from which, slices per bin
constexpr int P2_FB = 256, P2_T1 = KYB_MSM_P2_T1, P2_MAXCB = 128, P2_LMAX = KYB_MSM_P2_LMAX, P2_T = 1024;
static __global__ __launch_bounds__(P2_T) void coarse_hist_kernel(Plan p, int tiles1, int cb, const int32_t* __restrict__ digits,
                                                                  uint32_t* __restrict__ ch) {
    __shared__ uint32_t h[P2_T / 64][P2_MAXCB];  // a histogram per wave: the lanes of ONE wave meet in a counter often enough
    const int w = blockIdx.x % p.nwin, tile = blockIdx.x / p.nwin, wave = threadIdx.x >> 6;
    for (int j = threadIdx.x; j < (P2_T / 64) * P2_MAXCB; j += P2_T) (&h[0][0])[j] = 0;
    __syncthreads();
    const size_t lo = (size_t)tile * P2_T1, hi = lo + P2_T1 < p.n ? lo + P2_T1 : p.n;
    const int32_t* dw = digits + (size_t)w * p.n;
#pragma unroll 4
    for (size_t i = lo + threadIdx.x; i < hi; i += P2_T) {
        const int d = dw[i];
        if (d) atomicAdd(&h[wave][((d < 0 ? -d : d) - 1) / P2_FB], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < cb) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < P2_T / 64; k++) v += h[k][threadIdx.x];
        ch[((size_t)w * cb + threadIdx.x) * tiles1 + tile] = v;  // bin-major, tile-minor: the scan order
    }
}
static __global__ __launch_bounds__(P2_T) void coarse_scatter_kernel(Plan p, int tiles1, int cb, const int32_t* __restrict__ digits,
                                                                     const uint32_t* __restrict__ ch,
                                                                     const uint32_t* __restrict__ offs1,
                                                                     uint32_t* __restrict__ mid) {
    __shared__ uint32_t stage[P2_T1], dest[P2_T1];
    __shared__ uint32_t lpre[P2_MAXCB], gstart[P2_MAXCB], cur[P2_MAXCB], tot;
    const int w = blockIdx.x % p.nwin, tile = blockIdx.x / p.nwin;
    if ((int)threadIdx.x < cb) {
        const size_t j = ((size_t)w * cb + threadIdx.x) * tiles1 + tile;
        lpre[threadIdx.x] = ch[j];
        gstart[threadIdx.x] = offs1[j];
        cur[threadIdx.x] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // <= 128 counters
        uint32_t run = 0;
        for (int b = 0; b < cb; b++) {
            const uint32_t v = lpre[b];
            lpre[b] = run;
            run += v;
        }
        tot = run;
    }
    __syncthreads();
    const size_t lo = (size_t)tile * P2_T1, hi = lo + P2_T1 < p.n ? lo + P2_T1 : p.n;
    const int32_t* dw = digits + (size_t)w * p.n;
#pragma unroll 4
    for (size_t i = lo + threadIdx.x; i < hi; i += P2_T) {
        const int d = dw[i];
        if (d) {
            const uint32_t bk = (uint32_t)((d < 0 ? -d : d) - 1), bin = bk / P2_FB;
            const uint32_t r = atomicAdd(&cur[bin], 1u), pos = lpre[bin] + r;
            stage[pos] = (uint32_t)i | (d < 0 ? 1u << 23 : 0u) | (bk % P2_FB) << 24;
            dest[pos] = gstart[bin] + r;
        }
    }
    __syncthreads();
    const uint32_t m = tot;
    for (uint32_t q = threadIdx.x; q < m; q += P2_T) mid[dest[q]] = stage[q];  // neighbours in a bin are neighbours in mid
}
static __global__ __launch_bounds__(P2_T) void fine_sort_kernel(int tiles1, const uint32_t* __restrict__ offs1,
                                                                const uint32_t* __restrict__ mid, uint32_t* __restrict__ offs,
                                                                uint32_t* __restrict__ sorted, uint32_t* __restrict__ ngiant,
                                                                uint32_t* __restrict__ giant) {
    __shared__ uint32_t stage[P2_LMAX];
    __shared__ uint32_t cnt[P2_FB], pre[P2_FB];
    const size_t g = blockIdx.x;  // window * bins + bin
    const uint32_t lo = offs1[g * tiles1], hi = offs1[(g + 1) * tiles1], len = hi - lo;
    if (g == gridDim.x - 1 && threadIdx.x == 0) offs[(size_t)gridDim.x * P2_FB] = hi;
    if (len > (uint32_t)P2_GIANT) {  // a giant bin (one bucket with a large share of all points): giant_*_kernel, several workgroups
        if (threadIdx.x == 0) giant[atomicAdd(ngiant, 1u)] = (uint32_t)g;
        return;
    }
    if (threadIdx.x < P2_FB) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += P2_T) atomicAdd(&cnt[mid[i] >> 24], 1u);
    __syncthreads();
    if (threadIdx.x < P2_FB) pre[threadIdx.x] = cnt[threadIdx.x];
    __syncthreads();
    for (int off = 1; off < P2_FB; off <<= 1) {  // Hillis-Steele over the 256 counters
        uint32_t add = 0;
        if (threadIdx.x < P2_FB && (int)threadIdx.x >= off) add = pre[threadIdx.x - off];
        __syncthreads();
        if (threadIdx.x < P2_FB) pre[threadIdx.x] += add;
        __syncthreads();
    }
    if (threadIdx.x < P2_FB) {
        const uint32_t first = pre[threadIdx.x] - cnt[threadIdx.x];  // exclusive
        offs[g * P2_FB + threadIdx.x] = lo + first;
        pre[threadIdx.x] = first;
        cnt[threadIdx.x] = 0;  // the cursors now
    }
    __syncthreads();
    if (len <= (uint32_t)P2_LMAX) {
        for (uint32_t i = lo + threadIdx.x; i < hi; i += P2_T) {
            const uint32_t x = mid[i], f = x >> 24;
            stage[pre[f] + atomicAdd(&cnt[f], 1u)] = (x & 0x7fffffu) | ((x >> 23) & 1u) << 31;
        }
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < len; q += P2_T) sorted[lo + q] = stage[q];
    } else {
        for (uint32_t i = lo + threadIdx.x; i < hi; i += P2_T) {
            const uint32_t x = mid[i], f = x >> 24;
            sorted[lo + pre[f] + atomicAdd(&cnt[f], 1u)] = (x & 0x7fffffu) | ((x >> 23) & 1u) << 31;
        }
    }
}
// A GIANT bin (more than P2_GIANT entries: one bucket holding a large share of all points -- the carry-only top window of
// short scalars, equal scalars) would keep ONE workgroup of fine_sort_kernel busy for a millisecond.  It is cut into
// P2_GS slices: giant_count_kernel counts every slice's 256 buckets, giant_offs_kernel turns the counts into the bin's
// offs[] and every (slice, bucket)'s first slot, giant_scatter_kernel places the slices.  All three read the list of giant
// bins fine_sort_kernel left on the device and are a launch when it is empty.
static __global__ __launch_bounds__(P2_T) void giant_count_kernel(int tiles1, const uint32_t* __restrict__ offs1,
                                                                  const uint32_t* __restrict__ mid,
                                                                  const uint32_t* __restrict__ ngiant,
                                                                  const uint32_t* __restrict__ giant, uint32_t* __restrict__ gcnt) {
    __shared__ uint32_t cnt[P2_FB];
    const uint32_t ng = ngiant[0];
#pragma unroll 1
    for (uint32_t gi = blockIdx.y; gi < ng; gi += gridDim.y) {
        const size_t g = giant[gi];
        const uint32_t lo = offs1[g * tiles1], hi = offs1[(g + 1) * tiles1], per = (hi - lo + P2_GS - 1) / P2_GS;
        const uint32_t a = lo + blockIdx.x * per < hi ? lo + blockIdx.x * per : hi, b = a + per < hi ? a + per : hi;
        if (threadIdx.x < P2_FB) cnt[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t i = a + threadIdx.x; i < b; i += P2_T) atomicAdd(&cnt[mid[i] >> 24], 1u);
        __syncthreads();
        if (threadIdx.x < P2_FB) gcnt[((size_t)gi * P2_GS + blockIdx.x) * P2_FB + threadIdx.x] = cnt[threadIdx.x];
        __syncthreads();
    }
}
static __global__ __launch_bounds__(P2_FB) void giant_offs_kernel(int tiles1, const uint32_t* __restrict__ offs1,
                                                                  const uint32_t* __restrict__ ngiant,
                                                                  const uint32_t* __restrict__ giant, uint32_t* __restrict__ gcnt,
                                                                  uint32_t* __restrict__ offs) {
    __shared__ uint32_t pre[P2_FB];
    const uint32_t ng = ngiant[0];
#pragma unroll 1
    for (uint32_t gi = blockIdx.x; gi < ng; gi += gridDim.x) {
        const size_t g = giant[gi];
        const uint32_t lo = offs1[g * tiles1];
        uint32_t* col = gcnt + (size_t)gi * P2_GS * P2_FB + threadIdx.x;  // bucket threadIdx.x, slice-major
        uint32_t tot = 0;
        for (int sl = 0; sl < P2_GS; sl++) {  // within the bucket: the slices in order
            const uint32_t v = col[(size_t)sl * P2_FB];
            col[(size_t)sl * P2_FB] = tot;
            tot += v;
        }
        pre[threadIdx.x] = tot;
        __syncthreads();
        for (int off = 1; off < P2_FB; off <<= 1) {
            const uint32_t add = (int)threadIdx.x >= off ? pre[threadIdx.x - off] : 0u;
            __syncthreads();
            pre[threadIdx.x] += add;
            __syncthreads();
        }
        const uint32_t first = lo + pre[threadIdx.x] - tot;
        offs[g * P2_FB + threadIdx.x] = first;
        for (int sl = 0; sl < P2_GS; sl++) col[(size_t)sl * P2_FB] += first;  // first slot of (slice, bucket)
        __syncthreads();
    }
}
static __global__ __launch_bounds__(P2_T) void giant_scatter_kernel(int tiles1, const uint32_t* __restrict__ offs1,
                                                                    const uint32_t* __restrict__ mid,
                                                                    const uint32_t* __restrict__ ngiant,
                                                                    const uint32_t* __restrict__ giant,
                                                                    const uint32_t* __restrict__ gcnt, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t cur[P2_FB];
    const uint32_t ng = ngiant[0];
#pragma unroll 1
    for (uint32_t gi = blockIdx.y; gi < ng; gi += gridDim.y) {
        const size_t g = giant[gi];
        const uint32_t lo = offs1[g * tiles1], hi = offs1[(g + 1) * tiles1], per = (hi - lo + P2_GS - 1) / P2_GS;
        const uint32_t a = lo + blockIdx.x * per < hi ? lo + blockIdx.x * per : hi, b = a + per < hi ? a + per : hi;
        if (threadIdx.x < P2_FB) cur[threadIdx.x] = gcnt[((size_t)gi * P2_GS + blockIdx.x) * P2_FB + threadIdx.x];
        __syncthreads();
        for (uint32_t i = a + threadIdx.x; i < b; i += P2_T) {
            const uint32_t x = mid[i];
            sorted[atomicAdd(&cur[x >> 24], 1u)] = (x & 0x7fffffu) | ((x >> 23) & 1u) << 31;
        }
        __syncthreads();
    }
}
// the plan takes the two passes: whole bins of 256 buckets, an index that fits 23 bits.  KYB_MSM_SORT=single: never (A/B)
inline bool sort_two_pass(const Plan& p, size_t ne) {
    static const bool off = [] {
        const char* e = getenv("KYB_MSM_SORT");
        return e && e[0] == 's';
    }();
    // (below 2^19 entries per window the one pass is ahead: 1.70 against 1.85 ms for 2^16 points, 1.78 against 2.00 for 2^17)
    return !off && p.nb >= 8192 && p.nb / P2_FB <= P2_MAXCB && ne >= (size_t(1) << 19) && ne <= (size_t(1) << 23);
}

// Tiles per window of the one-pass sort: at most ONE round of the chip's CUs in all (a workgroup holds 128 KB of LDS;
// rounds 1-5 aimed at "about two per CU" and, for 17 windows of Ed25519 scalars, got 272 workgroups: a full round and a
// sixteenth of one, i.e. two), a tile of at least two points per bucket (the per-tile flush is per bucket).
// KYB_MSM_SORT_TILES forces a count (A/B runs: 14 .. 56 tiles all within 1 % for the 2^20-point BLS12-381 G1 MSM).
inline int sort_tiles(int num_cu, int nwin, size_t ne, int nb) {
    static const int forced = [] {
        const char* e = getenv("KYB_MSM_SORT_TILES");
        return e ? atoi(e) : 0;
    }();
    int tiles = forced > 0 ? forced : num_cu / nwin;
    if (tiles < 1) tiles = 1;
    while (tiles > 1 && (ne ? ne : 1) / tiles < 2 * (size_t)nb) tiles--;
    return tiles;
}

constexpr int MAXSUB = 256;
// Points per accumulate lane: a longer bucket is cut into pieces that are joined afterwards (one full addition per
// extra piece, bucket_kernel).  Twice the mean bucket length, so that only skewed digits split a bucket -- with a fixed
// 64 every second bucket of the 2^20-point BLS12-381 G1 MSM (mean 64) had a second, tiny piece: 6.10 -> 5.94 ms --
// between 64 and MAXSUB.  KYB_MSM_SUB overrides (experiments: profiles/r03_msm_knobs.json).
inline uint32_t piece_len(size_t ne, int nb) {
    static const int forced = [] {
        const char* e = getenv("KYB_MSM_SUB");
        return e ? atoi(e) : 0;
    }();
    size_t v = forced > 0 ? (size_t)forced : 2 * (ne / (size_t)nb + 1);
    v = (v + 31) / 32 * 32;
    return (uint32_t)(v < 64 ? 64 : (v > MAXSUB ? MAXSUB : v));
}

constexpr uint32_t LONG_PIECES = 4;  // a bucket of more pieces is joined by a workgroup (tree), not by one lane

// nsub[b] = number of SUB-sized pieces of bucket b; buckets of more than LONG_PIECES pieces (skewed digits: a short top
// window, equal or small scalars) are appended to longlist (counter in nlong[0]), buckets of 2 .. LONG_PIECES pieces to
// joinlist (counter in nlong[1], one atomic per wave): bucket_kernel visits those only.  An empty bucket is written here.
template <class A>
__global__ __launch_bounds__(256) void subcount_kernel(size_t nbk, uint32_t SUB, const uint32_t* __restrict__ offs,
                                                       uint32_t* __restrict__ nsub, uint32_t* __restrict__ nlong,
                                                       uint32_t* __restrict__ longlist, uint32_t* __restrict__ joinlist,
                                                       typename A::Acc* __restrict__ buckets) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t k = 1;
    if (b < nbk) {
        k = (offs[b + 1] - offs[b] + SUB - 1) / SUB;
        nsub[b] = k;
        if (k > LONG_PIECES) longlist[atomicAdd(nlong, 1u)] = (uint32_t)b;
        if (k == 0) {
            typename A::Acc id;
            A::identity(id);
            buckets[b] = id;
        }
    }
    const bool join = k >= 2 && k <= LONG_PIECES;
    const unsigned long long mask = __ballot(join);
    if (mask) {
        const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)mask) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(nlong + 1, (uint32_t)__popcll(mask));
        base = __shfl(base, leader);
        if (join) joinlist[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)b;
    }
}

// Piece t -> (first index into `sorted`, length), plus a histogram of the lengths.  The piece -> bucket map is a
// binary search in the scanned piece counts.
static __global__ __launch_bounds__(256) void piece_kernel(size_t nbk, size_t max_pieces, uint32_t SUB, const uint32_t* __restrict__ offs,
                                                           const uint32_t* __restrict__ suboffs,
                                                           uint32_t* __restrict__ plo, uint32_t* __restrict__ plen,
                                                           uint32_t* __restrict__ pdst, uint32_t* __restrict__ lenhist) {
    __shared__ uint32_t h[MAXSUB + 1];
    for (uint32_t i = threadIdx.x; i <= SUB; i += 256) h[i] = 0;
    __syncthreads();
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < max_pieces && t < suboffs[nbk]) {
        size_t lo_b = 0, hi_b = nbk;  // largest b with suboffs[b] <= t
        while (hi_b - lo_b > 1) {
            const size_t mid = (lo_b + hi_b) >> 1;
            if (suboffs[mid] <= t) lo_b = mid; else hi_b = mid;
        }
        const size_t b = lo_b;
        const uint32_t j = (uint32_t)(t - suboffs[b]);
        const uint32_t lo = offs[b] + j * SUB;
        const uint32_t end = offs[b + 1];
        const uint32_t len = (lo + SUB < end ? lo + SUB : end) - lo;
        plo[t] = lo;
        plen[t] = len;
        // where the piece's sum goes: a bucket's only piece (all but skewed buckets) is the bucket -- accumulate_kernel
        // stores it there and bucket_kernel has nothing to copy (74 us of the 2^20-point BLS12-381 G1 MSM)
        pdst[t] = suboffs[b + 1] - suboffs[b] == 1 ? (uint32_t)b | 0x80000000u : (uint32_t)t;
        atomicAdd(&h[SUB - len], 1u);  // bin 0 = longest
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= SUB; i += 256)
        if (h[i]) atomicAdd(&lenhist[i], h[i]);
}
// order[] = the pieces sorted by decreasing length (counting sort; ranks within a workgroup come from LDS atomics, one
// global atomic per length per workgroup reserves the range).
static __global__ __launch_bounds__(256) void piece_order_kernel(size_t nbk, size_t max_pieces, uint32_t SUB,
                                                                 const uint32_t* __restrict__ suboffs,
                                                                 const uint32_t* __restrict__ plen,
                                                                 const uint32_t* __restrict__ lenoffs,
                                                                 uint32_t* __restrict__ lencursor,
                                                                 uint32_t* __restrict__ order) {
    __shared__ uint32_t h[MAXSUB + 1], base[MAXSUB + 1];
    for (uint32_t i = threadIdx.x; i <= SUB; i += 256) h[i] = 0;
    __syncthreads();
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = t < max_pieces && t < suboffs[nbk];
    uint32_t bin = 0, rank = 0;
    if (valid) {
        bin = SUB - plen[t];
        rank = atomicAdd(&h[bin], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= SUB; i += 256)
        if (h[i]) base[i] = lenoffs[i] + atomicAdd(&lencursor[i], h[i]);
    __syncthreads();
    if (valid) order[base[bin] + rank] = (uint32_t)t;
}

// PieceOps<A>: the accumulator of a bucket piece -- a run of mixed additions -- may have a form of its own
// (A::Piece with piece_identity / piece_madd / piece_finish: XYZZ on the Weierstrass curves); Acc otherwise.
template <class A, class = void>
struct PieceOps {
    using type = typename A::Acc;
    __device__ static void identity(type& a) { A::identity(a); }
    __device__ static void madd(type& a, const typename A::Aff& p, bool neg) { A::madd(a, p, neg); }
    __device__ static void finish(typename A::Acc& r, const type& a) { r = a; }
};
template <class A>
struct PieceOps<A, decltype((void)sizeof(typename A::Piece))> {
    using type = typename A::Piece;
    __device__ static void identity(type& a) { A::piece_identity(a); }
    __device__ static void madd(type& a, const typename A::Aff& p, bool neg) { A::piece_madd(a, p, neg); }
    __device__ static void finish(typename A::Acc& r, const type& a) { A::piece_finish(r, a); }
};

// One lane per bucket piece: sums up to SUB points (mixed additions).  Lanes take the pieces in order of decreasing
// length, so the lanes of a wave run the same number of additions (bucket sizes are Poisson-spread: in bucket order a
// wave would wait for its longest piece, ~40 % above the mean at 32 points per bucket).
#ifndef KYB_MSM_ACC_WAVES
#define KYB_MSM_ACC_WAVES 2
#endif
template <class A>
__global__ __launch_bounds__(64, KYB_MSM_ACC_WAVES) void accumulate_kernel(size_t nbk, size_t max_pieces,
                                                        const typename A::Aff* __restrict__ aff,
                                                        const uint32_t* __restrict__ suboffs,
                                                        const uint32_t* __restrict__ order,
                                                        const uint32_t* __restrict__ plo,
                                                        const uint32_t* __restrict__ plen,
                                                        const uint32_t* __restrict__ pdst,
                                                        const uint32_t* __restrict__ sorted,
                                                        typename A::Acc* __restrict__ pieces,
                                                        typename A::Acc* __restrict__ buckets) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= max_pieces || i >= suboffs[nbk]) return;
    const uint32_t t = order[i];
    const uint32_t lo = plo[t], hi = lo + plen[t];
    typename PieceOps<A>::type acc;
    PieceOps<A>::identity(acc);
    // (fetching the next point during the current addition -- index and point are two dependent gathers -- measured no
    // different, 4.96 against 4.98 ms per 2^20-point MSM, profiles/r05_msm_prefetch_ab.jsonl: the second wave of the SIMD
    // already covers them; the loop is issue-bound)
#pragma unroll 1
    for (uint32_t q = lo; q < hi; q++) {
        const uint32_t e = sorted[q];
        const typename A::Aff pt = aff[e & 0x7fffffffu];
        PieceOps<A>::madd(acc, pt, (e >> 31) != 0);
    }
    typename A::Acc res;
    PieceOps<A>::finish(res, acc);
    const uint32_t d = pdst[t];
    if (d >> 31) buckets[d & 0x7fffffffu] = res;
    else pieces[t] = res;
}

// bucket b = sum of its 2 .. LONG_PIECES pieces, for the buckets of joinlist (a lone piece was stored as the bucket by
// accumulate_kernel, an empty bucket by subcount_kernel, long ones are left to bucket_long_kernel).  Grid-stride over the
// list.  One lane per join: the Weierstrass adapters take bucket_coop_kernel below instead (a one-lane addition is 45 us
// of latency, and that -- not the copying of 262 144 lone pieces -- was most of this stage's 74 us).
template <class A>
__global__ __launch_bounds__(64, 2) void bucket_kernel(const uint32_t* __restrict__ nlong, const uint32_t* __restrict__ joinlist,
                                                    const uint32_t* __restrict__ suboffs,
                                                    const typename A::Acc* __restrict__ pieces,
                                                    typename A::Acc* __restrict__ buckets) {
    const uint32_t cnt = nlong[1];
#pragma unroll 1
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += (size_t)gridDim.x * blockDim.x) {
        const uint32_t b = joinlist[j];
        const uint32_t lo = suboffs[b], hi = suboffs[b + 1];
        typename A::Acc acc = pieces[lo];
#pragma unroll 1
        for (uint32_t q = lo + 1; q < hi; q++) {
            const typename A::Acc v = pieces[q];
            A::add(acc, acc, v);
        }
        buckets[b] = acc;
    }
}

// Long buckets: one workgroup each (grid-stride over longlist); every thread sums a strided share of the pieces, then
// the workgroup adds its partial sums as a tree through LDS.  A bucket holding all 2^20 points (equal scalars) is
// 16 384 pieces: 64 + 8 dependent additions instead of 16 383.
template <class A>
constexpr int long_threads() {
    return sizeof(typename A::Acc) > 192 ? 128 : 256;  // the LDS tree must fit 64 KB
}
template <class A>
__global__ __launch_bounds__(256) void bucket_long_kernel(const uint32_t* __restrict__ nlong,
                                                          const uint32_t* __restrict__ longlist,
                                                          const uint32_t* __restrict__ suboffs,
                                                          const typename A::Acc* __restrict__ pieces,
                                                          typename A::Acc* __restrict__ buckets) {
    constexpr int T = long_threads<A>();
    __shared__ typename A::Acc sh[T];
    const uint32_t cnt = nlong[0];
#pragma unroll 1
    for (uint32_t j = blockIdx.x; j < cnt; j += gridDim.x) {
        const uint32_t b = longlist[j];
        const uint32_t lo = suboffs[b], hi = suboffs[b + 1];
        typename A::Acc acc;
        A::identity(acc);
#pragma unroll 1
        for (uint32_t q = lo + threadIdx.x; q < hi; q += T) {
            const typename A::Acc v = pieces[q];
            A::add(acc, acc, v);
        }
#pragma unroll 1
        for (int off = T / 2; off >= 1; off >>= 1) {
            sh[threadIdx.x] = acc;
            __syncthreads();
            if ((int)threadIdx.x < off) {
                const typename A::Acc v = sh[threadIdx.x + off];
                A::add(acc, acc, v);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) buckets[b] = acc;
    }
}

// partial[w][ch] = sum_{b in chunk} (b + 1) * B_b   (b = bucket index from 0, digit value b + 1)
template <class A>
__global__ __launch_bounds__(64) void reduce_kernel(Plan p, const typename A::Acc* __restrict__ buckets,
                                                    typename A::Acc* __restrict__ partial) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)p.nwin * p.nchunks) return;
    const size_t w = t / p.nchunks, ch = t - w * p.nchunks;
    const int lo = (int)ch * p.chunk;
    typename A::Acc run, tot;
    A::identity(run);
    A::identity(tot);
#pragma unroll 1
    for (int b = lo + p.chunk - 1; b >= lo; b--) {
        const typename A::Acc bk = buckets[w * p.nb + b];
        A::add(run, run, bk);
        A::add(tot, tot, run);
    }
    // tot = sum (b - lo + 1) B_b ; add lo * run   (lo < nb = 2^(c-1))
    typename A::Acc m;
    A::identity(m);
#pragma unroll 1
    for (int bit = p.c - 2; bit >= 0; bit--) {
        A::dbl(m, m);
        if ((lo >> bit) & 1) A::add(m, m, run);
    }
    A::add(tot, tot, m);
    partial[t] = tot;
}

// out[w][g] = sum of in[w][64 g .. 64 g + 64): one wave per group, the 64 partials summed as a tree through LDS
// (depth 6 additions instead of a 64-long dependent chain in one lane).  Applied until one partial per window is left.
template <class A>
__global__ __launch_bounds__(64) void tree_fold_kernel(int nwin, int nin, const typename A::Acc* __restrict__ in,
                                                       typename A::Acc* __restrict__ out) {
    __shared__ typename A::Acc sh[64];
    const int nout = (nin + 63) / 64;
    const int w = blockIdx.x / nout, g = blockIdx.x - w * nout;
    const int k = g * 64 + (int)threadIdx.x;
    typename A::Acc s;
    A::identity(s);
    if (k < nin) s = in[(size_t)w * nin + k];
#pragma unroll 1
    for (int off = 32; off >= 1; off >>= 1) {
        sh[threadIdx.x] = s;
        __syncthreads();
        if ((int)threadIdx.x < off) {
            const typename A::Acc v = sh[threadIdx.x + off];
            A::add(s, s, v);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)w * nout + g] = s;
}

// ---- The tail on cooperating lanes (adapters with COOP_SLOTS: the Weierstrass curves, msm_ws.cuh) ------------------
// Four lanes per point, coordinates and temporaries in LDS slots, a barrier between the product levels of a formula:
// an addition of the running sums is 5 multiplications deep instead of 16, a doubling 3 instead of 7.  The two
// kernels below replace reduce_kernel / tree_fold_kernel for those adapters (0.82 -> 0.55 ms and 0.41 -> 0.16 ms at
// 2^20 points; the final kernel keeps its three-lane register formulation, which is faster than slots for a lone
// chain of doublings).
template <class A, class = void>
struct HasCoopSlots {
    static constexpr bool value = false;
};
template <class A>
struct HasCoopSlots<A, decltype((void)A::COOP_SLOTS)> {
    static constexpr bool value = A::COOP_SLOTS != 0;
};

constexpr int REDUCE_FUSED_BITS = 4;  // log2 of reduce_coop_kernel's 16 groups
// partial[w][ch] = sum_{b in chunk} (b + 1) * B_b, one GROUP of four lanes per chunk, 16 groups per workgroup.
// runs != null (round 6, the split tail below): the chunk's own part only -- partial[w][ch] = sum (b - lo + 1) B_b and
// runs[w][ch] = sum B_b; the lo * run term of every chunk -- a double-and-add of 2 log2(nchunks) + log2(chunk) steps, two
// thirds of this kernel's 42 for the 2^20-point BLS12-381 G1 MSM, half of its additions computed and dropped -- is left
// to tree_fold_bits_coop_kernel.
template <class A>
__global__ __launch_bounds__(64, 2) void reduce_coop_kernel(Plan p, const typename A::Acc* __restrict__ buckets,
                                                         typename A::Acc* __restrict__ partial,
                                                         typename A::Acc* __restrict__ runs, int fuse) {
    constexpr int G = 16, RUN = 0, TOT = 3, BK = 6, M = 9, T = 12, NS = T + A::COOP_TEMPS;
    __shared__ typename A::Slot slots[G * NS];
    __shared__ uint32_t flags[G * 2];
    const int gi = (int)threadIdx.x >> 2, r = (int)threadIdx.x & 3;
    const size_t total = (size_t)p.nwin * p.nchunks;
    const size_t t = (size_t)blockIdx.x * G + gi;
    const bool live = t < total;
    const size_t tc = live ? t : total - 1;  // a dead group walks the last chunk again: every lane meets every barrier
    const size_t w = tc / p.nchunks, ch = tc - w * p.nchunks;
    const int lo = (int)ch * p.chunk;
    typename A::Slot* S = slots + gi * NS;
    uint32_t* fl = flags + gi * 2;
    const typename A::Acc* bw = buckets + w * p.nb;
    A::slot_load(S, RUN, bw + lo + p.chunk - 1, r);
    A::slot_load(S, TOT, bw + lo + p.chunk - 1, r);
    A::slot_identity(S, M, r);
    __syncthreads();
    // One loop over the chunk's whole schedule, so that the addition and the doubling are inlined once each:
    //   steps [0, 2 (chunk - 1)):  run += B_b ; tot += run            (b from the chunk's top bucket down)
    //   then per bit of ch, from the top: m = 2 m ; m += run if the bit is set   (lo * run, lo = ch * chunk: the
    //     additions a group does not want are computed and dropped; bit positions no group of the workgroup wants
    //     are skipped), log2(chunk) more doublings, and tot += m.
    int chbits = 0, tz = 0;
    while ((1 << chbits) < p.nchunks) chbits++;
    while ((1 << tz) < p.chunk) tz++;
    const bool split = runs != nullptr;
    // fuse (split, nchunks a multiple of 16): the first four levels of the split tail's tree (tree_fold_bits_coop_kernel)
    // right here, over the workgroup's 16 chunks -- RUN slots: the bit tree (A in group 0, D_m in group 2^m); TOT slots: the
    // plain sum, carried by the LAST group of every node, which the bit tree leaves idle.  One addition per group and level.
    const int nsum = 2 * (p.chunk - 1), nmul = split ? 0 : 2 * chbits + tz, ntree = fuse ? REDUCE_FUSED_BITS : 0;
    const int nsteps = nsum + nmul + (split ? 0 : 1) + ntree;
#pragma unroll 1
    for (int s = 0; s < nsteps; s++) {
        bool is_dbl = false, commit = true, skip = false;
        int P = TOT, Q = M;
        if (s < nsum) {
            if ((s & 1) == 0) {
                A::slot_load(S, BK, bw + (lo + p.chunk - 2 - (s >> 1)), r);
                __syncthreads();
                P = RUN;
                Q = BK;
            } else {
                P = TOT;
                Q = RUN;
            }
        } else if (!split && s < nsum + 2 * chbits) {
            const int k = s - nsum, bit = chbits - 1 - (k >> 1);
            if ((k & 1) == 0) {
                is_dbl = true;
            } else {
                commit = ((ch >> bit) & 1) != 0;
                skip = !__syncthreads_or(commit ? 1 : 0);
                P = M;
                Q = RUN;
            }
        } else if (!split && s < nsum + nmul) {
            is_dbl = true;
        } else if (fuse) {
            const int off = 1 << (s - nsum), low = gi & (2 * off - 1);
            const bool t_act = (low & (low - 1)) == 0 && low < off, w_act = low == 2 * off - 1;
            if (r < 3) {
                if (t_act) S[BK + r].f = slots[(gi + off) * NS + RUN + r].f;
                if (w_act) S[BK + r].f = slots[(gi - off) * NS + TOT + r].f;
            }
            __syncthreads();
            P = t_act ? RUN : TOT;
            Q = BK;
            commit = t_act || w_act;
        }
        if (is_dbl) A::coop_dbl_slots(S, r, M, T, true);
        else if (!skip) A::coop_add(S, fl, r, P, Q, T, commit);
    }
    if (fuse) {  // rows of nchunks / 16: W (nwin), D_m (nwin x 4), A (nwin) -- tree_fold_bits_coop_kernel's layout
        const size_t ncur = (size_t)p.nchunks / G, blk = blockIdx.x, w0 = blk / ncur, c0 = blk - w0 * ncur;
        if (gi == G - 1) A::slot_store(partial + w0 * ncur + c0, S, TOT, r);
        if (gi == 0) A::slot_store(partial + ((size_t)p.nwin * (1 + REDUCE_FUSED_BITS) + w0) * ncur + c0, S, RUN, r);
        if (gi && (gi & (gi - 1)) == 0) {
            const int m = gi == 1 ? 0 : (gi == 2 ? 1 : (gi == 4 ? 2 : 3));
            A::slot_store(partial + ((size_t)p.nwin + w0 * REDUCE_FUSED_BITS + m) * ncur + c0, S, RUN, r);
        }
    } else if (live) {
        A::slot_store(partial + t, S, TOT, r);
        if (split) A::slot_store(runs + t, S, RUN, r);
    }
}

// out[w][g] = sum of in[w][FG g .. FG g + FG): a group per partial, added as a tree (depth log2 FG)
template <class A>
constexpr int fold_groups() {
    return sizeof(typename A::Slot) <= 48 ? 64 : 32;  // 16 slots per group within 48 KB of LDS
}
template <class A>
__global__ __launch_bounds__(4 * fold_groups<A>(), 2) void tree_fold_coop_kernel(int nwin, int nin,
                                                                              const typename A::Acc* __restrict__ in,
                                                                              typename A::Acc* __restrict__ out, int per) {
    constexpr int FG = fold_groups<A>(), P = 0, Q = 3, T = 6, NS = T + A::COOP_TEMPS;
    __shared__ typename A::Slot slots[FG * NS];
    __shared__ uint32_t flags[FG * 2];
    const int gi = (int)threadIdx.x >> 2, r = (int)threadIdx.x & 3;
    const int nout = (nin + FG * per - 1) / (FG * per);  // per = 2: a group starts from the sum of two inputs (one more level)
    const int w = blockIdx.x / nout, g = blockIdx.x - w * nout;
    const int k = g * FG * per + gi;
    typename A::Slot* S = slots + gi * NS;
    if (k < nin) A::slot_load(S, P, in + (size_t)w * nin + k, r);
    else A::slot_identity(S, P, r);
    if (per == 2) {  // uniform
        if (k + FG < nin) A::slot_load(S, Q, in + (size_t)w * nin + k + FG, r);
        else A::slot_identity(S, Q, r);
        __syncthreads();
        A::coop_add(S, flags + gi * 2, r, P, Q, T, true);
    }
    __syncthreads();
#pragma unroll 1
    for (int off = FG / 2; off >= 1; off >>= 1) {
        if (gi < off && r < 3) S[Q + r].f = slots[(gi + off) * NS + P + r].f;
        __syncthreads();
        A::coop_add(S, flags + gi * 2, r, P, Q, T, gi < off);
    }
    if (gi == 0) A::slot_store(out + (size_t)w * nout + g, S, P, r);
}

// bucket_kernel on cooperating lanes: a group of four per bucket of joinlist.  The one-lane join is a 45 us addition, and
// a few hundred buckets of a 2^20-point BLS12-381 G1 MSM do have a second piece (the halves' top window is not uniform:
// the fold of k1 into (-z^2 / 2, z^2 / 2] doubles the density of part of its range).
template <class A>
__global__ __launch_bounds__(64, 2) void bucket_coop_kernel(const uint32_t* __restrict__ nlong, const uint32_t* __restrict__ joinlist,
                                                         const uint32_t* __restrict__ suboffs,
                                                         const typename A::Acc* __restrict__ pieces,
                                                         typename A::Acc* __restrict__ buckets) {
    constexpr int G = 16, P = 0, Q = 3, T = 6, NS = T + A::COOP_TEMPS;
    __shared__ typename A::Slot slots[G * NS];
    __shared__ uint32_t flags[G * 2];
    const int gi = (int)threadIdx.x >> 2, r = (int)threadIdx.x & 3;
    typename A::Slot* S = slots + gi * NS;
    const uint32_t cnt = nlong[1];
#pragma unroll 1
    for (size_t j0 = (size_t)blockIdx.x * G; j0 < cnt; j0 += (size_t)gridDim.x * G) {
        const size_t j = j0 + gi;
        const bool live = j < cnt;
        uint32_t b = 0, lo = 0, hi = 0;
        if (live) {
            b = joinlist[j];
            lo = suboffs[b];
            hi = suboffs[b + 1];
            A::slot_load(S, P, pieces + lo, r);
        } else {
            A::slot_identity(S, P, r);
        }
        __syncthreads();
#pragma unroll 1
        for (uint32_t s = 1; s < LONG_PIECES; s++) {
            const bool more = live && lo + s < hi;
            if (!__syncthreads_or(more ? 1 : 0)) break;
            if (more) A::slot_load(S, Q, pieces + lo + s, r);
            else A::slot_identity(S, Q, r);
            __syncthreads();
            A::coop_add(S, flags + gi * 2, r, P, Q, T, more);
        }
        if (live) A::slot_store(buckets + b, S, P, r);
        __syncthreads();
    }
}

// Long buckets (more than LONG_PIECES pieces: skewed digits, a carry-only top window, equal scalars) on cooperating lanes,
// in two launches: bucket_long_kernel above is ONE workgroup per bucket whose threads first add a strided share of the
// pieces each with the one-lane addition (45 us apiece) -- 840 us for the half-of-all-points bucket of 2^20 short scalars.
// Launch 1 gives every 2 FG pieces of a long bucket a workgroup of the cooperative fold (the sum of a slice; a bucket of
// one slice is finished there); launch 2 folds a bucket's slice sums, 2 FG at a time with a running total.
template <class A>
__device__ __forceinline__ void coop_fold_range(typename A::Slot* slots, uint32_t* flags, const typename A::Acc* __restrict__ src,
                                                int count, int gi, int r) {
    // count <= 2 FG inputs: group gi starts from src[gi] + src[gi + FG]; tree over the groups; the sum ends in group 0's P
    constexpr int FG = fold_groups<A>(), P = 0, Q = 3, T = 6, NS = T + A::COOP_TEMPS + 3;
    typename A::Slot* S = slots + gi * NS;
    if (gi < count) A::slot_load(S, P, src + gi, r);
    else A::slot_identity(S, P, r);
    if (gi + FG < count) A::slot_load(S, Q, src + gi + FG, r);
    else A::slot_identity(S, Q, r);
    __syncthreads();
    if (count > FG) A::coop_add(S, flags + gi * 2, r, P, Q, T, true);  // (uniform)
    int top = FG / 2;
    while (top >= 1 && top >= count) top >>= 1;  // levels whose partners would all be identities are skipped
#pragma unroll 1
    for (int off = top; off >= 1; off >>= 1) {
        if (gi < off && r < 3) S[Q + r].f = slots[(gi + off) * NS + P + r].f;
        __syncthreads();
        A::coop_add(S, flags + gi * 2, r, P, Q, T, gi < off);
    }
}
template <class A>
__global__ __launch_bounds__(4 * fold_groups<A>(), 2) void bucket_long_coop1_kernel(const uint32_t* __restrict__ nlong,
                                                                                 const uint32_t* __restrict__ longlist,
                                                                                 const uint32_t* __restrict__ suboffs,
                                                                                 const typename A::Acc* __restrict__ pieces,
                                                                                 typename A::Acc* __restrict__ lpart,
                                                                                 typename A::Acc* __restrict__ buckets) {
    constexpr int FG = fold_groups<A>(), P = 0, T = 6, NS = T + A::COOP_TEMPS + 3, SL = 2 * FG;
    __shared__ typename A::Slot slots[FG * NS];
    __shared__ uint32_t flags[FG * 2];
    const int gi = (int)threadIdx.x >> 2, r = (int)threadIdx.x & 3;
    const uint32_t cnt = nlong[0];
    // work item t = (long bucket t mod cnt, slices t / cnt, + 32, ...): many buckets of one slice and one bucket of many slices
    // both spread over the grid (a grid of (slices, buckets) left 512 one-slice buckets to 64 rows, eight in a row each)
    constexpr uint32_t VS = 32;
#pragma unroll 1
    for (size_t t = blockIdx.x; t < (size_t)cnt * VS; t += gridDim.x) {
        const uint32_t j = (uint32_t)(t % cnt), s0 = (uint32_t)(t / cnt);  // first slices first: consecutive workgroups, consecutive buckets
        const uint32_t b = longlist[j], lo = suboffs[b], np = suboffs[b + 1] - lo, ns = (np + SL - 1) / SL;
#pragma unroll 1
        for (uint32_t sl = s0; sl < ns; sl += VS) {
            const uint32_t left = np - sl * SL;
            coop_fold_range<A>(slots, flags, pieces + lo + sl * SL, (int)(left < (uint32_t)SL ? left : (uint32_t)SL), gi, r);
            if (gi == 0) A::slot_store(ns == 1 ? buckets + b : lpart + lo + sl, slots, P, r);
            __syncthreads();
        }
    }
}
template <class A>
__global__ __launch_bounds__(4 * fold_groups<A>(), 2) void bucket_long_coop2_kernel(const uint32_t* __restrict__ nlong,
                                                                                 const uint32_t* __restrict__ longlist,
                                                                                 const uint32_t* __restrict__ suboffs,
                                                                                 const typename A::Acc* __restrict__ lpart,
                                                                                 typename A::Acc* __restrict__ buckets) {
    constexpr int FG = fold_groups<A>(), P = 0, Q = 3, T = 6, TOT = T + A::COOP_TEMPS, NS = TOT + 3, SL = 2 * FG;
    __shared__ typename A::Slot slots[FG * NS];
    __shared__ uint32_t flags[FG * 2];
    const int gi = (int)threadIdx.x >> 2, r = (int)threadIdx.x & 3;
    typename A::Slot* S = slots + gi * NS;
    const uint32_t cnt = nlong[0];
#pragma unroll 1
    for (uint32_t j = blockIdx.x; j < cnt; j += gridDim.x) {
        const uint32_t b = longlist[j], lo = suboffs[b], np = suboffs[b + 1] - lo, ns = (np + SL - 1) / SL;
        if (ns == 1) continue;  // finished by the first launch (uniform)
#pragma unroll 1
        for (uint32_t c0 = 0; c0 < ns; c0 += SL) {  // 2 FG slice sums at a time; group 0 keeps the running total
            const uint32_t left = ns - c0;
            coop_fold_range<A>(slots, flags, lpart + lo + c0, (int)(left < (uint32_t)SL ? left : (uint32_t)SL), gi, r);
            if (c0) {
                if (r < 3) S[Q + r].f = S[TOT + r].f;
                __syncthreads();
                A::coop_add(S, flags + gi * 2, r, P, Q, T, gi == 0);
            }
            if (r < 3) S[TOT + r].f = S[P + r].f;
            __syncthreads();
        }
        if (gi == 0) A::slot_store(buckets + b, slots, P, r);
        __syncthreads();
    }
}

// The split tail (round 6).  With W_j = sum_{b in chunk j} (b - lo_j + 1) B_b and T_j = sum_{b in chunk j} B_b from
// reduce_coop_kernel, a window's sum is  sum_j W_j + chunk * sum_j j T_j,  and over the bits of the chunk number
//     sum_j j T_j = sum_k 2^k D_k,   D_k = sum of T_j over the j whose bit k is set.
// The D_k fall out of the SAME tree that sums the T_j, at the same depth: a node of level k (2^k leaves) carries its sum A
// and D_0 .. D_(k-1) of its leaves; joining L and R, A = A_L + A_R, D_m = D_m(L) + D_m(R), and the new D_k is A_R as it
// stands.  With the node's A in its first group and D_m in group 2^m of its span, every join is "group g += group g + 2^k"
// for the groups whose low k + 1 bits are zero or a single bit below 2^k -- at most 32 additions side by side per level,
// one addition deep like the plain tree.  A launch consumes log2(FG) bits of j; its D rows join the plain rows (summed
// as they are by the next launch) and its A row is the bits row of the next launch.  Rows: [0, nplain) plain,
// [nplain, nplain + nbits) bits; out rows: the plain ones in place, D of bits row w and bit m < lb_out (the bits the chunk
// numbers still have) at nplain + w lb_out + m, the A rows behind them.  What is left for the doubling chains of
// final_rows_kernel is 2^(c w) W_w and 2^(c w + tz + k) D_k(w): one wave each, all at once -- no addition waits for a
// doubling any more.
template <class A>
constexpr int fold_bits() {
    return fold_groups<A>() == 64 ? 6 : 5;
}
template <class A>
__global__ __launch_bounds__(4 * fold_groups<A>(), 2) void tree_fold_bits_coop_kernel(int nplain, int nbits, int nin, int lb_out,
                                                                                   const typename A::Acc* __restrict__ in,
                                                                                   typename A::Acc* __restrict__ out) {
    constexpr int FG = fold_groups<A>(), P = 0, Q = 3, T = 6, NS = T + A::COOP_TEMPS;
    __shared__ typename A::Slot slots[FG * NS];
    __shared__ uint32_t flags[FG * 2];
    const int gi = (int)threadIdx.x >> 2, r = (int)threadIdx.x & 3;
    const int nout = (nin + FG - 1) / FG;
    const int row = blockIdx.x / nout, g = blockIdx.x - row * nout;
    const bool bits = row >= nplain;
    const int k0 = g * FG + gi;
    typename A::Slot* S = slots + gi * NS;
    if (k0 < nin) A::slot_load(S, P, in + (size_t)row * nin + k0, r);
    else A::slot_identity(S, P, r);
    __syncthreads();
    const int have = nin - g * FG;  // inputs of this block: the levels above them would add identities (a last launch of 4
                                    // inputs per row: 2 levels instead of 6, 64 -> 25 us)
#pragma unroll 1
    for (int off = 1; off < FG && off < have; off <<= 1) {
        const int low = gi & (2 * off - 1);
        const bool active = bits ? ((low & (low - 1)) == 0 && low < off) : low == 0;
        if (active && r < 3) S[Q + r].f = slots[(gi + off) * NS + P + r].f;
        __syncthreads();
        A::coop_add(S, flags + gi * 2, r, P, Q, T, active);
    }
    if (!bits) {
        if (gi == 0) A::slot_store(out + (size_t)row * nout + g, S, P, r);
    } else {
        const int w = row - nplain;
        if (gi == 0) A::slot_store(out + (size_t)(nplain + nbits * lb_out + w) * nout + g, S, P, r);
        if (gi && (gi & (gi - 1)) == 0) {
            int m = 0;
            while ((1 << m) < gi) m++;
            if (m < lb_out) A::slot_store(out + (size_t)(nplain + w * lb_out + m) * nout + g, S, P, r);
        }
    }
}

// Coop<A>::value: the adapter offers dbl_coop (COOP lanes per point)
template <class A, class = void>
struct Coop {
    static constexpr int value = 1;
};
template <class A>
struct Coop<A, decltype((void)A::COOP)> {
    static constexpr int value = A::COOP;
};

constexpr int FINAL_T = 512;

// One workgroup: window w's sum is doubled w*c times, the nwin results are added as a tree and encoded.  This tail is a
// pure dependency chain ((nwin-1)*c doublings), so each point is held by COOP lanes that split the independent field
// products of a doubling between them when the adapter offers dbl_coop.
template <class A>
__global__ __launch_bounds__(FINAL_T) void final_kernel(Plan p, const typename A::Acc* __restrict__ wsum,
                                                        typename A::Acc* __restrict__ winsum,
                                                        const uint32_t* __restrict__ bad, uint8_t* __restrict__ out) {
    constexpr int L = Coop<A>::value;
    const int t = threadIdx.x, w = t / L, r = t - w * L;
    typename A::Acc s;
    A::identity(s);
    if (w < p.nwin) s = wsum[w];
    if constexpr (L > 1) {
        __shared__ typename A::Field sh[FINAL_T + L];
        const int total = (p.nwin - 1) * p.c;
#pragma unroll 1
        for (int k = 0; k < total; k++) {
            typename A::Acc d = s;
            A::dbl_coop(d, r, sh + w * L);
            if (k < w * p.c) s = d;
        }
    } else {
#pragma unroll 1
        for (int k = 0; k < w * p.c && w < p.nwin; k++) A::dbl(s, s);
    }
    if (w < p.nwin && r == 0) winsum[w] = s;
    __threadfence_block();
    __syncthreads();
#pragma unroll 1
    for (int off = FINAL_T / 2; off >= 1; off >>= 1) {
        if (off < p.nwin) {  // uniform
            if (t < off && t + off < p.nwin) {
                const typename A::Acc a = winsum[t], b = winsum[t + off];
                A::add(s, a, b);
                winsum[t] = s;
            }
            __threadfence_block();
            __syncthreads();
        }
    }
    if (t == 0) {
        if (*bad) {
            for (int k = 0; k < A::OUT; k++) out[k] = 0;
        } else {
            const typename A::Acc v = winsum[0];
            A::encode(out, v);
        }
    }
}

// Doublings of chain b of the split tail.  Chain b < nwin: window b's sum, c b doublings.  The rows behind them are D_k of
// window w in the order the tree produced them (lb0 bits per window from the reduce kernel, then per fold launch the bits
// that are left, lb at most), doubled c w + tz + k times.
__device__ __forceinline__ int tail_chain_doublings(const Plan& p, int b, int lb0, int lb, int tz, int chbits) {
    if (b < p.nwin) return b * p.c;
    int q = b - p.nwin, ww, k;
    if (q < p.nwin * lb0) {
        ww = q / lb0;
        k = q - ww * lb0;
    } else {
        q -= p.nwin * lb0;
        int k0 = lb0, lbl = chbits - k0 < lb ? chbits - k0 : lb;  // a launch emits the bits that are left, lb at most
        while (q >= p.nwin * lbl) {
            q -= p.nwin * lbl;
            k0 += lbl;
            lbl = chbits - k0 < lb ? chbits - k0 : lb;
        }
        ww = q / lbl;
        k = k0 + (q - ww * lbl);
    }
    return ww * p.c + tz + k;
}

// The split tail's doubling chains for adapters without the limb-per-lane rows (BLS12-381 G2, the BN curves): final_kernel's
// cooperative doubling (COOP lanes per chain), one WAVE per workgroup (21 chains of three lanes: its barriers stay inside
// the wave -- three waves per workgroup measured 3.69 against 2.59 ms for the 269 doublings of a 2^18-point G2 MSM), as
// many workgroups as the terms need.  A chain is as long as final_kernel's longest was -- what the split buys these
// curves is the reduce kernel's 24 - 28 steps.
constexpr int CHAINS_T = 64;
template <class A>
__global__ __launch_bounds__(CHAINS_T) void final_chains_kernel(Plan p, const typename A::Acc* __restrict__ src,
                                                           typename A::Acc* __restrict__ shifted, int nchains, int lb0, int lb,
                                                           int tz, int chbits) {
    constexpr int L = Coop<A>::value, T = CHAINS_T, CPB = T / L;
    __shared__ typename A::Field sh[T + L];
    __shared__ int longest;
    const int t = threadIdx.x, ci = t / L, r = t - ci * L;
    const int b = blockIdx.x * CPB + ci;
    const bool live = ci < CPB && b < nchains;
    typename A::Acc s;
    A::identity(s);
    int ndbl = 0;
    if (live) {
        s = src[b];
        ndbl = tail_chain_doublings(p, b, lb0, lb, tz, chbits);
    }
    if (t == 0) longest = 0;
    __syncthreads();
    atomicMax(&longest, ndbl);
    __syncthreads();
    const int total = longest;
#pragma unroll 1
    for (int k = 0; k < total; k++) {
        typename A::Acc d = s;
        A::dbl_coop(d, r, sh + ci * L);  // (a thread past the last whole chain has slots of its own: sh[T + L])
        if (k < ndbl) s = d;
    }
    if (live && r == 0) shifted[b] = s;
}

// ---- The doubling chains of the final stage with ONE LIMB PER LANE (rowfp.cuh; adapters with ROW_FINAL: a base field of
// 13 x 30-bit limbs) -- round 6.  final_kernel above holds all windows in one wave, three lanes per window, and every
// doubling costs it three 539-instruction field products in a row: 770 us for the 128 doublings of a 2^20-point G1 MSM,
// on ONE of the chip's 1 024 SIMDs.  Here every window has a wave of its own; the wave's four rows hold the window's sum
// and share the seven products of a doubling (three levels of a 57-multiply-add product).  The sums leave in the packed
// form, 2^(c w) times what they were, and final_kernel -- called with c = 0 -- adds and encodes them.
template <class A, class = void>
struct HasRowFinal {
    static constexpr bool value = false;
};
template <class A>
struct HasRowFinal<A, decltype((void)A::ROW_FINAL)> {
    static constexpr bool value = A::ROW_FINAL != 0;
};
#if defined(KYB_ROWFP_INCLUDED)
template <class A>
__global__ __launch_bounds__(64) void final_rows_kernel(Plan p, const typename A::Acc* __restrict__ wsum,
                                                        typename A::Acc* __restrict__ shifted, int lb0, int lb, int tz, int chbits) {
    using C = typename A::RowC;
    using namespace rowfp;
    __shared__ uint32_t limbs[3][ROW];
    const int w = blockIdx.x;
    const int ndbl = tail_chain_doublings(p, w, lb0, lb, tz, chbits);
    const auto cx = make_ctx<C>();
    const auto dc = make_dbl_consts<C>();
    const V32 row = row_of_lane();
    const typename A::Acc* src = wsum + w;
    JacRow<C> pt{load_packed<C>(src->X.v), load_packed<C>(src->Y.v), load_packed<C>(src->Z.v)};
#pragma unroll 1
    for (int k = 0; k < ndbl; k++) jac_dbl_wave<C>(cx, dc, row, pt);
    const V32 X = below_2p<C>(cx, pt.X), Y = below_2p<C>(cx, pt.Y), Z = below_2p<C>(cx, pt.Z);
    if (threadIdx.x < ROW) {
        limbs[0][threadIdx.x] = X;
        limbs[1][threadIdx.x] = Y;
        limbs[2][threadIdx.x] = Z;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        typename A::Field f;
        finish_limbs<C>(f, limbs[threadIdx.x]);
        typename A::Field* dst = reinterpret_cast<typename A::Field*>(shifted + w);
        dst[threadIdx.x] = f;
    }
}
#endif

// the split tail (reduce_coop_kernel's runs, tree_fold_bits_coop_kernel, one doubling chain per term) wants the cooperative
// slots and a doubling chain that is not one lane's: the limb-per-lane rows (BLS12-381 G1) or the adapter's dbl_coop
template <class A>
constexpr bool split_tail() {
    return HasCoopSlots<A>::value && (HasRowFinal<A>::value || Coop<A>::value > 1);
}

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

// Enqueue the whole MSM on `st`.  d_status may be null.  n == 0 writes the identity encoding.
template <class A>
int run(DeviceCtx* ctx, size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
        hipStream_t st, uint32_t flags = 0) {
    if (n * Split<A>::value >= (size_t(1) << 31)) {
        set_error("msm: n too large");
        return KYB_E_ARG;
    }
    if ((n && (!d_scalars || !d_points)) || !d_out) {
        set_error("msm: bad argument");
        return KYB_E_ARG;
    }
    if (int frc = check_flags(flags & ~KYB_F_SCALAR_BITS_MASK, 1, false, "msm")) return frc;
    std::lock_guard<std::recursive_mutex> enq_lock(ctx->enq_mu);  // context.h: one whole pipeline at a time per device
    const size_t ne = n * Split<A>::value;  // points after the adapter's endomorphism split
    // KYB_F_SCALAR_BITS(b): only the low b bits of every scalar count (proportionally fewer windows); adapters that
    // split their scalars through an endomorphism already work on halves and ignore it
    int bits = Split<A>::bits;
    const int want = (int)((flags >> 16) & 0x1ffu);
    if (Split<A>::value == 1 && want && want < bits) bits = want;
    const Plan p = make_plan(ne ? ne : 1, bits, Split<A>::cmax);
    Plan pr = p;
    pr.n = ne;
    pr.flags = flags;
    const size_t nbk = (size_t)p.nwin * p.nb;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align256(bytes);
        return o;
    };
    const size_t o_aff = take(sizeof(typename A::Aff) * (ne ? ne : 1));
    const size_t o_dig = take(sizeof(int32_t) * (ne ? ne : 1) * p.nwin);
    const size_t o_sorted = take(sizeof(uint32_t) * (ne ? ne : 1) * p.nwin);
    const int tiles = sort_tiles(ctx->num_cu, p.nwin, ne, p.nb);
    const size_t m2 = nbk * (size_t)tiles;
    const size_t o_hist = take(sizeof(uint32_t) * m2);
    const size_t o_total = take(sizeof(uint32_t) * nbk);  // points per bucket
    const bool two_pass = sort_two_pass(p, ne);
    const int cb = p.nb / P2_FB, tiles1 = (int)(((ne ? ne : 1) + P2_T1 - 1) / P2_T1);
    const size_t m1 = two_pass ? (size_t)p.nwin * cb * tiles1 : 0;
    const size_t o_mid = take(two_pass ? sizeof(uint32_t) * (ne ? ne : 1) * p.nwin : 0);
    const size_t o_ch = take(sizeof(uint32_t) * m1);
    const size_t o_offs1 = take(sizeof(uint32_t) * (m1 + 1));
    const size_t max_giant = two_pass ? (ne ? ne : 1) * (size_t)p.nwin / P2_GIANT + 1 : 0;  // bins of more than P2_GIANT entries
    const size_t o_giant = take(sizeof(uint32_t) * max_giant);
    const size_t o_gcnt = take(sizeof(uint32_t) * max_giant * P2_GS * P2_FB);
    const uint32_t SUB = piece_len(ne ? ne : 1, p.nb);
    const size_t o_lenhist = take(sizeof(uint32_t) * (MAXSUB + 2));
    const size_t o_lencursor = take(sizeof(uint32_t) * (MAXSUB + 2));
    const size_t o_nlong = take(256);
    const size_t o_bad = take(256);
    const size_t zero_end = off;  // lenhist, lencursor, nlong, bad are zeroed together
    const size_t o_offs = take(sizeof(uint32_t) * (nbk + 1));
    const size_t o_nsub = take(sizeof(uint32_t) * nbk);
    const size_t o_suboffs = take(sizeof(uint32_t) * (nbk + 1));
    const size_t max_pieces = nbk + ((ne ? ne : 1) * (size_t)p.nwin + SUB - 1) / SUB;
    const size_t o_pieces = take(sizeof(typename A::Acc) * max_pieces);
    const size_t o_plo = take(sizeof(uint32_t) * max_pieces);
    const size_t o_plen = take(sizeof(uint32_t) * max_pieces);
    const size_t o_order = take(sizeof(uint32_t) * max_pieces);
    const size_t o_pdst = take(sizeof(uint32_t) * max_pieces);
    const size_t o_longlist = take(sizeof(uint32_t) * nbk);
    const size_t o_joinlist = take(sizeof(uint32_t) * nbk);
    const size_t o_lpart = take(HasCoopSlots<A>::value ? sizeof(typename A::Acc) * max_pieces : 0);  // slice sums of long buckets
    const size_t o_buckets = take(sizeof(typename A::Acc) * nbk);
    const int nfold = (p.nchunks + 31) / 32;  // first fold level: 64 (one-lane tail) or 32 / 64 (cooperative tail) partials per output
    size_t n_partial = (size_t)p.nwin * p.nchunks, n_fold = (size_t)p.nwin * nfold, n_chains = (size_t)p.nwin;
    if constexpr (split_tail<A>()) {
        // the split tail's rows (tree_fold_bits_coop_kernel): W and T rows side by side, then ping-pong between the two
        // buffers with nwin * fold_bits more plain rows per launch
        constexpr int FG = fold_groups<A>(), LB = fold_bits<A>();
        n_partial *= 2;
        auto rows = [&](size_t nplain, int ncur) {  // for either start: with and without the reduce kernel's own levels
            bool to_fold = true;
            while (ncur > 1) {
                const int nout = (ncur + FG - 1) / FG;
                const size_t need = (nplain + (size_t)p.nwin * (LB + 1)) * nout;
                size_t& dst = to_fold ? n_fold : n_partial;
                if (need > dst) dst = need;
                nplain += (size_t)p.nwin * LB;
                ncur = nout;
                to_fold = !to_fold;
            }
            if (nplain > n_chains) n_chains = nplain;
        };
        rows((size_t)p.nwin, p.nchunks);
        if (p.nchunks >= 16) rows((size_t)p.nwin * (1 + REDUCE_FUSED_BITS), p.nchunks / 16);
    }
    const size_t o_partial = take(sizeof(typename A::Acc) * n_partial);
    const size_t o_fold = take(sizeof(typename A::Acc) * n_fold);
    const size_t o_shift = take(sizeof(typename A::Acc) * n_chains);
    const size_t o_shift2 = take(sizeof(typename A::Acc) * ((n_chains + 63) / 64));
    const size_t o_tile = take(sizeof(uint32_t) * (((nbk > m1 ? nbk : m1) + SCAN_TILE - 1) / SCAN_TILE + 2));
    const size_t o_winsum = take(sizeof(typename A::Acc) * p.nwin);
    void* ws;
    int rc = ctx_workspace(ctx, WS_MSM, st, off, &ws);
    if (rc) return rc;
    uint8_t* base = (uint8_t*)ws;
    auto* aff = (typename A::Aff*)(base + o_aff);
    auto* digits = (int32_t*)(base + o_dig);
    auto* sorted = (uint32_t*)(base + o_sorted);
    auto* hist = (uint32_t*)(base + o_hist);
    auto* total = (uint32_t*)(base + o_total);
    auto* mid = (uint32_t*)(base + o_mid);
    auto* ch = (uint32_t*)(base + o_ch);
    auto* offs1 = (uint32_t*)(base + o_offs1);
    auto* giant = (uint32_t*)(base + o_giant);
    auto* gcnt = (uint32_t*)(base + o_gcnt);
    auto* bad = (uint32_t*)(base + o_bad);
    auto* offs = (uint32_t*)(base + o_offs);
    auto* nsub = (uint32_t*)(base + o_nsub);
    auto* suboffs = (uint32_t*)(base + o_suboffs);
    auto* pieces = (typename A::Acc*)(base + o_pieces);
    auto* plo = (uint32_t*)(base + o_plo);
    auto* plen = (uint32_t*)(base + o_plen);
    auto* order = (uint32_t*)(base + o_order);
    auto* pdst = (uint32_t*)(base + o_pdst);
    auto* lenhist = (uint32_t*)(base + o_lenhist);
    auto* lencursor = (uint32_t*)(base + o_lencursor);
    auto* nlong = (uint32_t*)(base + o_nlong);
    auto* longlist = (uint32_t*)(base + o_longlist);
    auto* joinlist = (uint32_t*)(base + o_joinlist);
    auto* lpart = (typename A::Acc*)(base + o_lpart);
    auto* buckets = (typename A::Acc*)(base + o_buckets);
    auto* partial = (typename A::Acc*)(base + o_partial);
    auto* winsum = (typename A::Acc*)(base + o_winsum);
    auto* folded = (typename A::Acc*)(base + o_fold);
    auto* shift = (typename A::Acc*)(base + o_shift);
    auto* shift2 = (typename A::Acc*)(base + o_shift2);
    auto* tile = (uint32_t*)(base + o_tile);
    KYB_HIP_CHECK(hipMemsetAsync(base + o_lenhist, 0, zero_end - o_lenhist, st));
    if (n) {
        bool light = false;
        if constexpr (LightDecode<A>::value) {
            static const bool off = [] {  // KYB_MSM_DECODE=full: the one decode kernel for every calling convention (A/B)
                const char* e = getenv("KYB_MSM_DECODE");
                return e && e[0] == 'f';
            }();
            light = !off && (flags & FLAG_UNCOMPRESSED) && flag_trusted(flags, 0);
            if (light)
                hipLaunchKernelGGL((decode_kernel<A, true>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, pr, n, (const uint8_t*)d_scalars,
                                   (const uint8_t*)d_points, aff, digits, (uint8_t*)d_status, bad);
        }
        if (!light)
            hipLaunchKernelGGL(decode_kernel<A>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, pr, n, (const uint8_t*)d_scalars,
                               (const uint8_t*)d_points, aff, digits, (uint8_t*)d_status, bad);
    }
    if (p.nb > HIST_MAX_NB) {
        set_error("msm: window too wide for the LDS-staged sort");
        return KYB_E_ARG;
    }
    if (two_pass) {
        hipLaunchKernelGGL(coarse_hist_kernel, dim3((unsigned)(tiles1 * p.nwin)), dim3(P2_T), 0, st, pr, tiles1, cb, (const int32_t*)digits, ch);
        launch_scan(ch, offs1, m1, tile, st);
        hipLaunchKernelGGL(coarse_scatter_kernel, dim3((unsigned)(tiles1 * p.nwin)), dim3(P2_T), 0, st, pr, tiles1, cb, (const int32_t*)digits,
                           (const uint32_t*)ch, (const uint32_t*)offs1, mid);
        hipLaunchKernelGGL(fine_sort_kernel, dim3((unsigned)(p.nwin * cb)), dim3(P2_T), 0, st, tiles1, (const uint32_t*)offs1,
                           (const uint32_t*)mid, offs, sorted, nlong + 2, giant);
        hipLaunchKernelGGL(giant_count_kernel, dim3(P2_GS, 16), dim3(P2_T), 0, st, tiles1, (const uint32_t*)offs1, (const uint32_t*)mid,
                           (const uint32_t*)(nlong + 2), (const uint32_t*)giant, gcnt);
        hipLaunchKernelGGL(giant_offs_kernel, dim3(64), dim3(P2_FB), 0, st, tiles1, (const uint32_t*)offs1, (const uint32_t*)(nlong + 2),
                           (const uint32_t*)giant, gcnt, offs);
        hipLaunchKernelGGL(giant_scatter_kernel, dim3(P2_GS, 16), dim3(P2_T), 0, st, tiles1, (const uint32_t*)offs1, (const uint32_t*)mid,
                           (const uint32_t*)(nlong + 2), (const uint32_t*)giant, (const uint32_t*)gcnt, sorted);
    } else {
    hipLaunchKernelGGL(hist_lds_kernel, dim3(tiles * p.nwin), dim3(HIST_T), 0, st, pr, tiles, (const int32_t*)digits, hist);
    hipLaunchKernelGGL(tile_scan_kernel, dim3((unsigned)((nbk + 255) / 256)), dim3(256), 0, st, nbk, p.nb, tiles, hist, total);
    launch_scan(total, offs, nbk, tile, st);
    static const int xcd_major = [] {  // KYB_MSM_SORT_XCD=0: tile-minor workgroup order (A/B)
        const char* e = getenv("KYB_MSM_SORT_XCD");
        return e && e[0] == '0' ? 0 : 1;
    }();
    hipLaunchKernelGGL(scatter_lds_kernel, dim3(tiles * p.nwin), dim3(HIST_T), 0, st, pr, tiles, (const int32_t*)digits,
                       (const uint32_t*)hist, (const uint32_t*)offs, sorted, xcd_major);
    }
    hipLaunchKernelGGL(subcount_kernel<A>, dim3((unsigned)((nbk + 255) / 256)), dim3(256), 0, st, nbk, SUB, (const uint32_t*)offs, nsub,
                       nlong, longlist, joinlist, buckets);
    launch_scan(nsub, suboffs, nbk, tile, st);
    const unsigned pgrid = (unsigned)((max_pieces + 255) / 256);
    hipLaunchKernelGGL(piece_kernel, dim3(pgrid), dim3(256), 0, st, nbk, max_pieces, SUB, (const uint32_t*)offs,
                       (const uint32_t*)suboffs, plo, plen, pdst, lenhist);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(SCAN_T), 0, st, lenhist, (size_t)(SUB + 1));
    hipLaunchKernelGGL(piece_order_kernel, dim3(pgrid), dim3(256), 0, st, nbk, max_pieces, SUB, (const uint32_t*)suboffs,
                       (const uint32_t*)plen, (const uint32_t*)lenhist, lencursor, order);
    hipLaunchKernelGGL(accumulate_kernel<A>, dim3((unsigned)((max_pieces + 63) / 64)), dim3(64), 0, st, nbk, max_pieces, aff,
                       (const uint32_t*)suboffs, (const uint32_t*)order, (const uint32_t*)plo, (const uint32_t*)plen,
                       (const uint32_t*)pdst, sorted, pieces, buckets);
    bool coop_join = false;
    if constexpr (HasCoopSlots<A>::value) {
        static const bool lane_join = [] {  // KYB_MSM_JOIN=lane: the one-lane bucket_kernel (A/B)
            const char* e = getenv("KYB_MSM_JOIN");
            return e && e[0] == 'l';
        }();
        coop_join = !lane_join;
        if (coop_join)
            hipLaunchKernelGGL(bucket_coop_kernel<A>, dim3((unsigned)(nbk / 16 < 1024 ? nbk / 16 + 1 : 1024)), dim3(64), 0, st,
                               (const uint32_t*)nlong, (const uint32_t*)joinlist, (const uint32_t*)suboffs, (const typename A::Acc*)pieces,
                               buckets);
    }
    if (!coop_join)
    hipLaunchKernelGGL(bucket_kernel<A>, dim3((unsigned)(nbk / 64 < 1024 ? nbk / 64 + 1 : 1024)), dim3(64), 0, st, (const uint32_t*)nlong,
                       (const uint32_t*)joinlist, (const uint32_t*)suboffs, (const typename A::Acc*)pieces, buckets);
    if (coop_join) {
        if constexpr (HasCoopSlots<A>::value) {
            constexpr int FG = fold_groups<A>();
            hipLaunchKernelGGL(bucket_long_coop1_kernel<A>, dim3(4096), dim3(4 * FG), 0, st, (const uint32_t*)nlong, (const uint32_t*)longlist,
                               (const uint32_t*)suboffs, (const typename A::Acc*)pieces, lpart, buckets);
            hipLaunchKernelGGL(bucket_long_coop2_kernel<A>, dim3(1024), dim3(4 * FG), 0, st, (const uint32_t*)nlong, (const uint32_t*)longlist,
                               (const uint32_t*)suboffs, (const typename A::Acc*)lpart, buckets);
        }
    } else {
    hipLaunchKernelGGL(bucket_long_kernel<A>, dim3(1024), dim3(long_threads<A>()), 0, st, (const uint32_t*)nlong,
                       (const uint32_t*)longlist, (const uint32_t*)suboffs, pieces, buckets);
    }
    const size_t nred = (size_t)p.nwin * p.nchunks;
    if constexpr (HasCoopSlots<A>::value) {
        // the tail on cooperating lanes (four per point); KYB_MSM_TAIL=lane keeps the one-lane kernels (A/B)
        static const bool lane_tail = [] {
            const char* e = getenv("KYB_MSM_TAIL");
            return e && e[0] == 'l';
        }();
        if (!lane_tail) {
            constexpr int FG = fold_groups<A>();
            if constexpr (split_tail<A>()) {
                // KYB_MSM_REDUCE=mul: every chunk multiplies its own lo * run (rounds 3-5); KYB_MSM_FINAL=lanes implies it
                static const bool split = [] {
                    const char* e = getenv("KYB_MSM_REDUCE");
                    const char* f = getenv("KYB_MSM_FINAL");
                    return !(e && e[0] == 'm') && !(f && f[0] == 'l');
                }();
                if (split && p.nwin > 1) {
                    constexpr int LB = fold_bits<A>();
                    int chbits = 0, tz = 0;
                    while ((1 << chbits) < p.nchunks) chbits++;
                    while ((1 << tz) < p.chunk) tz++;
                    static const bool fuse_ok = [] {  // KYB_MSM_REDUCE=nofuse: the whole tree in the fold launches (A/B)
                        const char* e = getenv("KYB_MSM_REDUCE");
                        return !(e && e[0] == 'n');
                    }();
                    const int fuse = fuse_ok && p.nchunks >= 16 ? 1 : 0, lb0 = fuse ? REDUCE_FUSED_BITS : 0;
                    hipLaunchKernelGGL(reduce_coop_kernel<A>, dim3((unsigned)((nred + 15) / 16)), dim3(64), 0, st, pr, buckets, partial,
                                       partial + nred, fuse);
                    typename A::Acc* cur = partial;
                    typename A::Acc* nxt = folded;
                    int nplain = p.nwin * (1 + lb0), done = lb0;
                    for (int ncur = fuse ? p.nchunks / 16 : p.nchunks; ncur > 1;) {
                        const int nout = (ncur + FG - 1) / FG, lb_out = chbits - done < LB ? chbits - done : LB;
                        hipLaunchKernelGGL(tree_fold_bits_coop_kernel<A>, dim3((unsigned)((nplain + p.nwin) * nout)), dim3(4 * FG), 0, st,
                                           nplain, p.nwin, ncur, lb_out, (const typename A::Acc*)cur, nxt);
                        typename A::Acc* t = cur;
                        cur = nxt;
                        nxt = t;
                        nplain += p.nwin * lb_out;
                        done += lb_out;
                        ncur = nout;
                    }
                    // nplain terms now: the W sums and every D_k, each with a doubling chain of its own
                    bool rows_done = false;
#if defined(KYB_ROWFP_INCLUDED)
                    if constexpr (HasRowFinal<A>::value) {
                        hipLaunchKernelGGL(final_rows_kernel<A>, dim3((unsigned)nplain), dim3(64), 0, st, pr, (const typename A::Acc*)cur, shift,
                                           lb0, LB, tz, chbits);
                        rows_done = true;
                    }
#endif
                    if (!rows_done) {
                        constexpr int CPB = CHAINS_T / Coop<A>::value;
                        hipLaunchKernelGGL(final_chains_kernel<A>, dim3((unsigned)((nplain + CPB - 1) / CPB)), dim3(CHAINS_T), 0, st, pr,
                                           (const typename A::Acc*)cur, shift, nplain, lb0, LB, tz, chbits);
                    }
                    typename A::Acc* a = shift;
                    typename A::Acc* b = shift2;
                    int m = nplain;
                    while (m > 2) {
                        const int nout = (m + 2 * FG - 1) / (2 * FG);
                        hipLaunchKernelGGL(tree_fold_coop_kernel<A>, dim3((unsigned)nout), dim3(4 * FG), 0, st, 1, m, (const typename A::Acc*)a, b, 2);
                        typename A::Acc* t = a;
                        a = b;
                        b = t;
                        m = nout;
                    }
                    Plan p0 = pr;
                    p0.c = 0;     // nothing left to double
                    p0.nwin = m;  // one or two terms for final_kernel's own tree
                    hipLaunchKernelGGL(final_kernel<A>, dim3(1), dim3(64), 0, st, p0, (const typename A::Acc*)a, winsum, bad, (uint8_t*)d_out);
                    KYB_HIP_CHECK(hipGetLastError());
                    return KYB_OK;
                }
            }
            hipLaunchKernelGGL(reduce_coop_kernel<A>, dim3((unsigned)((nred + 15) / 16)), dim3(64), 0, st, pr, buckets, partial,
                               (typename A::Acc*)nullptr, 0);
            typename A::Acc* cur = partial;
            typename A::Acc* nxt = folded;
            int ncur = p.nchunks;
            while (ncur > 1) {
                const int nout = (ncur + FG - 1) / FG;
                hipLaunchKernelGGL(tree_fold_coop_kernel<A>, dim3((unsigned)(p.nwin * nout)), dim3(4 * FG), 0, st, p.nwin, ncur,
                                   cur, nxt, 1);
                typename A::Acc* t = cur;
                cur = nxt;
                nxt = t;
                ncur = nout;
            }
            // (the final kernel stays the three-lane register version: its slot-based counterpart measured 0.91 ms
            // against 0.78 for the 112 doublings of the 2^20-point G1 MSM)
            // (the final kernel stays the three-lane register version: a slot-based one measured 5.48 against 5.39 ms
            // for the whole 2^20-point MSM -- a lone chain of doublings gains nothing from slots)
            const unsigned final_t = (unsigned)((p.nwin * Coop<A>::value + 63) / 64 * 64);
#if defined(KYB_ROWFP_INCLUDED)
            if constexpr (HasRowFinal<A>::value) {
                // the doubling chains one wave per window on the limb-per-lane arithmetic (KYB_MSM_FINAL=lanes: the old kernel, A/B)
                static const bool lanes_final = [] {
                    const char* e = getenv("KYB_MSM_FINAL");
                    return e && e[0] == 'l';
                }();
                if (!lanes_final && p.nwin > 1 && p.nwin <= FG) {
                    typename A::Acc* shifted = cur == partial ? folded : partial;  // the fold buffer not holding the sums
                    hipLaunchKernelGGL(final_rows_kernel<A>, dim3((unsigned)p.nwin), dim3(64), 0, st, pr, (const typename A::Acc*)cur, shifted, 0, 1, 0, 0);
                    // the nwin shifted sums are one more row of partials for the cooperative fold (a tree of four-lane additions:
                    // final_kernel's own tree is one lane per addition), and final_kernel is left with the encoding
                    typename A::Acc* total = cur;  // the sums were read by the kernel above: their buffer is free again
                    hipLaunchKernelGGL(tree_fold_coop_kernel<A>, dim3(1), dim3(4 * FG), 0, st, 1, p.nwin, (const typename A::Acc*)shifted, total, 1);
                    Plan p0 = pr;
                    p0.c = 0;     // nothing left to double
                    p0.nwin = 1;  // nor to add
                    hipLaunchKernelGGL(final_kernel<A>, dim3(1), dim3(64), 0, st, p0, (const typename A::Acc*)total, winsum, bad, (uint8_t*)d_out);
                    KYB_HIP_CHECK(hipGetLastError());
                    return KYB_OK;
                }
            }
#endif
            hipLaunchKernelGGL(final_kernel<A>, dim3(1), dim3(final_t), 0, st, pr, cur, winsum, bad, (uint8_t*)d_out);
            KYB_HIP_CHECK(hipGetLastError());
            return KYB_OK;
        }
    }
    hipLaunchKernelGGL(reduce_kernel<A>, dim3((unsigned)((nred + 63) / 64)), dim3(64), 0, st, pr, buckets, partial);
    // fold the per-window chunk partials 64 at a time (ping-pong between `partial` and `folded`) down to one each
    typename A::Acc* cur = partial;
    typename A::Acc* nxt = folded;
    int ncur = p.nchunks;
    while (ncur > 1) {
        const int nout = (ncur + 63) / 64;
        hipLaunchKernelGGL(tree_fold_kernel<A>, dim3((unsigned)(p.nwin * nout)), dim3(64), 0, st, p.nwin, ncur, cur, nxt);
        typename A::Acc* t = cur;
        cur = nxt;
        nxt = t;
        ncur = nout;
    }
    // only as many waves as hold windows: an idle wave sharing a SIMD with a working one would halve its issue rate
    const unsigned final_t = (unsigned)((p.nwin * Coop<A>::value + 63) / 64 * 64);
    hipLaunchKernelGGL(final_kernel<A>, dim3(1), dim3(final_t), 0, st, pr, cur, winsum, bad, (uint8_t*)d_out);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}

// Host-buffer wrapper on the calling thread's device: copy in, run, copy out, synchronise.
template <class A>
int run_host_single(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out, uint8_t* status,
                    uint32_t flags = 0) {
    if ((n && (!scalars || !points)) || !out) {
        set_error("msm: bad argument");
        return KYB_E_ARG;
    }
    if (int frc = check_flags(flags & ~KYB_F_SCALAR_BITS_MASK, 1, false, "msm")) return frc;
    DeviceCtx* ctx;
    int rc = get_ctx(&ctx);
    if (rc) return rc;
    // the pipeline lives in the per-device workspace: host-buffer MSM calls on one device are serialised.  Lock order
    // as in every other host-buffer call -- the staging pool first, the enqueue mutex second (taken the other way round
    // here until the concurrency test of tests/test_gpu_soak.py ran this against a host-buffer multiplication on a
    // second thread: each held what the other waited for)
    // (round 6: the pipeline's workspace belongs to the pool's stream, so the enqueue mutex is held for the enqueue only --
    // run() takes it -- and a second thread's upload overlaps this call's kernels)
    StageScope sc_(ctx);
    StageBuf d_s, d_p, d_o, d_st;
    rc = d_s.upload(scalars, n * 32);
    if (rc == KYB_OK) rc = d_p.upload(points, n * A::wire_size(flags));
    if (rc == KYB_OK) rc = d_o.alloc(A::OUT);
    if (rc == KYB_OK) rc = d_st.alloc(n + 1);
    if (rc == KYB_OK) rc = run<A>(ctx, n, d_s.p, d_p.p, d_o.p, d_st.p, sc_.stream(), flags);
    if (rc == KYB_OK) rc = d_o.download(out, A::OUT);
    if (rc == KYB_OK && status && n) rc = d_st.download(status, n);
    return rc;
}

// Host-buffer entry: with several devices configured (context.h md_*), the POINTS are sharded -- every device runs the
// whole pipeline on its slice and yields one encoded partial point; the calling thread then adds the partials (an
// MSM of `devices` points with unit scalars on its own device).  Bucket arrays never leave a device; what crosses the
// host is devices x (point bytes) -- the exchange SURVEY.md section 8e prescribes, without a collective because one
// process owns all the devices here (kyber_amd/dist.py is the one-process-per-GPU variant with an RCCL all-gather).
template <class A>
int run_host(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out, uint8_t* status,
             uint32_t flags = 0) {
    if (!md_active(n)) return run_host_single<A>(n, scalars, points, out, status, flags);
    if (!scalars || !points || !out) {
        set_error("msm: bad argument");
        return KYB_E_ARG;
    }
    if (int frc = check_flags(flags & ~KYB_F_SCALAR_BITS_MASK, 1, false, "msm")) return frc;
    // the width the partial buffer is sized with and the width the shards run at are the same snapshot (a concurrent
    // kyb_set_devices() must not make shard s write past partial[w])
    const std::vector<int> devs = md_devices();
    const int w = (int)devs.size();
    if (w < 2 || n < (size_t)w) return run_host_single<A>(n, scalars, points, out, status, flags);
    std::vector<uint8_t> partial((size_t)w * A::OUT), st_tmp;
    uint8_t* stp = status;
    if (!stp) {
        st_tmp.assign(n, 0);
        stp = st_tmp.data();
    }
    const size_t wire = A::wire_size(flags);
    int rc = md_run(n, [&](int s, size_t lo, size_t hi) {
        return run_host_single<A>(hi - lo, scalars + 32 * lo, points + wire * lo, partial.data() + (size_t)s * A::OUT, stp + lo,
                                  flags);
    }, &devs);
    if (rc) return rc;
    bool bad = false;
    for (size_t i = 0; i < n; i++) bad |= stp[i] != 0;
    if (bad) {  // the single-device contract: any rejected point -> all-zero output, status names it
        for (int k = 0; k < A::OUT; k++) out[k] = 0;
        return KYB_OK;
    }
    std::vector<uint8_t> unit((size_t)w * 32, 0);
    for (int s = 0; s < w; s++) unit[(size_t)s * 32 + (A::SCALAR_BE ? 31 : 0)] = 1;
    // the partials are this library's own encodings: validated by construction
    return run_host_single<A>((size_t)w, unit.data(), partial.data(), out, nullptr, A::COMBINE_FLAGS);
}

// ----------------------------------------------------------------------------------------- batched PubPoly.Eval
// out[i] = sum_j commits[j] x_i^j with x_i = idx[i] + 1, by Horner from the top coefficient -- share.PubPoly.Eval
// (share/poly.go:340-348: xi = 1 + i; v = v * xi + commits[j]) for many indices at once: what PubPoly.Shares / Check
// and the DKG / VSS verification loops evaluate once per participant.  One launch decodes the t commitments (one lane
// each, the adapter's UnmarshalBinary checks), one launch evaluates: a lane per index, t - 1 steps of
// [x] acc (33-bit double-and-add: x <= 2^32) + one mixed addition.  If any commitment is rejected every output is
// all-zero bytes and status[j] names it.
// (same register budget as decode_kernel: the two share their out-of-line callees, and a callee reachable from a
// kernel without the budget is compiled without it -- which silently cost decode_kernel its second wave)
template <class A>
__global__ __launch_bounds__(64, DecodeWaves<A>::value) void poly_decode_kernel(size_t t, const uint8_t* __restrict__ commits,
                                                         typename A::Aff* __restrict__ aff, uint8_t* __restrict__ status,
                                                         uint32_t* __restrict__ bad, uint32_t flags) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= t) return;
    typename A::Aff a;
    const int st = A::decode(a, commits + A::wire_size(flags) * j, flags);
    aff[j] = a;
    if (status) status[j] = (uint8_t)st;
    if (st) atomicAdd(bad, 1u);
}
template <class A>
__global__ __launch_bounds__(64) void poly_eval_kernel(size_t n, const uint32_t* __restrict__ idx, size_t t,
                                                       const typename A::Aff* __restrict__ aff,
                                                       const uint32_t* __restrict__ bad, uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t* o = out + (size_t)A::OUT * i;
    if (*bad) {
        for (int k = 0; k < A::OUT; k++) o[k] = 0;
        return;
    }
    const uint64_t x = (uint64_t)idx[i] + 1;
    typename A::Acc acc;
    A::identity(acc);
    if (t) {
        const typename A::Aff top = aff[t - 1];
        A::madd(acc, top, false);
    }
#pragma unroll 1
    for (size_t j = t; j-- > 1;) {
        typename A::Acc r;  // r = x * acc, MSB first
        A::identity(r);
#pragma unroll 1
        for (int b = 32; b >= 0; b--) {
            A::dbl(r, r);
            if ((x >> b) & 1) A::add(r, r, acc);
        }
        const typename A::Aff c = aff[j - 1];
        A::madd(r, c, false);
        acc = r;
    }
    A::encode(o, acc);
}
// Enqueue on `st`; d_status may be null.
template <class A>
int poly_eval_run(DeviceCtx* ctx, size_t n, const void* d_idx, size_t t, const void* d_commits, void* d_out,
                  void* d_status, uint32_t flags, hipStream_t st) {
    if (t >= (size_t(1) << 24)) {
        set_error("poly_eval: threshold too large");
        return KYB_E_ARG;
    }
    if ((n && (!d_idx || !d_out)) || (t && !d_commits)) {
        set_error("poly_eval: bad argument");
        return KYB_E_ARG;
    }
    if (int frc = check_flags(flags, 1, false, "poly_eval")) return frc;
    std::lock_guard<std::recursive_mutex> enq_lock(ctx->enq_mu);
    void* ws;
    int rc = ctx_workspace(ctx, WS_MSM, st, sizeof(typename A::Aff) * (t ? t : 1) + 512, &ws);
    if (rc) return rc;
    uint32_t* bad = (uint32_t*)ws;
    auto* aff = (typename A::Aff*)((uint8_t*)ws + 256);
    KYB_HIP_CHECK(hipMemsetAsync(bad, 0, 256, st));
    if (t)
        hipLaunchKernelGGL(poly_decode_kernel<A>, dim3((unsigned)((t + 63) / 64)), dim3(64), 0, st, t,
                           (const uint8_t*)d_commits, aff, (uint8_t*)d_status, bad, flags);
    if (n)
        hipLaunchKernelGGL(poly_eval_kernel<A>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, (const uint32_t*)d_idx, t,
                           aff, (const uint32_t*)bad, (uint8_t*)d_out);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
template <class A>
int poly_eval_host(size_t n, const uint32_t* idx, size_t t, const uint8_t* commits, uint8_t* out, uint8_t* status,
                   uint32_t flags) {
    if ((n && (!idx || !out)) || (t && !commits)) {
        set_error("poly_eval: bad argument");
        return KYB_E_ARG;
    }
    if (int frc = check_flags(flags, 1, false, "poly_eval")) return frc;
    DeviceCtx* ctx;
    int rc = get_ctx(&ctx);
    if (rc) return rc;
    StageScope sc_(ctx);
    StageBuf d_i, d_c, d_o, d_st;
    rc = d_i.upload(idx, n * 4);
    if (rc == KYB_OK) rc = d_c.upload(commits, t * A::wire_size(flags));
    if (rc == KYB_OK) rc = d_o.alloc(n * A::OUT);
    if (rc == KYB_OK) rc = d_st.alloc(t + 1);
    if (rc == KYB_OK) rc = poly_eval_run<A>(ctx, n, d_i.p, t, d_c.p, d_o.p, d_st.p, flags, sc_.stream());
    if (rc == KYB_OK) rc = d_o.download(out, n * A::OUT);
    if (rc == KYB_OK && status && t) rc = d_st.download(status, t);
    return rc;
}

}  // namespace msm
}  // namespace kyb
