// Cooperative tower machine: the pairing (Miller loop + final exponentiation) of 64 (pairs of) points per workgroup,
// computed by TWELVE waves -- one per Fp coefficient of an Fp12 element -- out of the CU's LDS.
//
// Replaces the per-lane pairing of round 1 (pairing/bn256 optate.go:126-274 miller / finalExponentiation / optimalAte,
// and the Pair / ValidatePairing of the BLS12-381 backends behind pairing/bls12381/kilic/suite.go:57-75), which held a
// whole Fp12 accumulator per lane: 512 registers, one wave per SIMD, every Fp12 operand through scratch (0.27 of the
// integer-MAD peak, 29 GB of scratch traffic per 65 536 pairings).  Here:
//
//   * lane l of every wave works on pairing l of the workgroup; wave w owns ONE Fp coefficient of whatever Fp12 /
//     Fp2-tuple value an instruction produces.  All live values (f, T, the line, temporaries) sit in LDS as `slots`
//     [slot][limb quad][lane] -- 16-byte ds_read_b128 per lane, consecutive lanes consecutive, conflict-free.
//   * one instruction = for every wave, out = MontgomeryReduce( sum_t X_t * Y_t  +  R * sum_t Z_t ) with up to 24 terms,
//     X, Y, Z = small-integer combinations c1 S[a] + c2 S[b] of slots (Karatsuba-free schoolbook over the tower: the
//     products accumulate UNREDUCED in 27 64-bit columns, one reduction per output coefficient instead of one per
//     multiplication), then a workgroup barrier.  An Fp12 multiplication is ONE instruction of 12 products per wave;
//     the code a wave executes is ~6 KB -- instruction-cache resident -- and about 110 VGPRs: three waves per SIMD.
//   * Fp elements are 14 signed 28-bit limbs (balanced digits, |limb| <= 2^27, Montgomery radix R = 2^392) so that
//     v_mad_i64_i32 is the whole inner loop, subtraction is limb-wise, and there is no conditional subtraction anywhere:
//     values are only bounded (|v| < B p, B tracked by the program generator), canonical form is produced once, at the
//     output.  (bn256: 10 limbs, R = 2^280; bn254: 10 limbs of 27 bits, R = 2^270.)
//   * the program (which slots, which coefficients) is data: gen_tower_vm.py emits it, tests/test_tower_vm_program.py
//     replays it in exact integer arithmetic against the oracle's pairing, and proves the column / limb bounds.
#pragma once
#include "hd.h"

namespace kyb {
namespace tvm {

constexpr int WAVES = 12;
constexpr int LANES = 64;
constexpr int THREADS = WAVES * LANES;
constexpr int REC_WORDS = 64;  // one (instruction, wave) record: header + up to 31 two-word terms
constexpr int MAX_CONSTS = 32;

// header word
//   0-5 out slot | 6-11 n terms | 12 barrier before the store (an input slot of some wave is overwritten) |
//   13 raw: no products, no Montgomery reduction -- the linear terms are normalised as they are |
//   14-18 post scale (0 = 1) | 19-20 store mask (0 none, 1 lanes live in pair A, 2 pair B, 3 lanes whose exponent has
//   bit (last word & 0xffff) - repetition * (last word >> 16) set: run<.., EXPO = true> only) | 21-24 opcode
enum Op : uint32_t { OP_DOT = 0, OP_IDLE = 1, OP_GLOAD = 2, OP_INV = 3, OP_GT_STORE = 4, OP_IS_ONE = 5, OP_CLOAD = 6, OP_SPILL = 7, OP_FILL = 8, OP_CMP_EQ = 9, OP_GCLOAD = 10, OP_CMP_NZ2 = 11 };
//   GLOAD: slot <- input[word 1] of this pairing (packed words of the per-lane field code, taken as an integer)
//   CLOAD: slot <- constant[word 1];  INV: slot <- Inv(slot[word 1]);  SPILL / FILL: slot <-> global scratch (word 1, wave)
//   GT_STORE: canonical big-endian bytes of the slot at byte offset (word 1 & 0xffff) of the pairing's output
//   IS_ONE: slot != (word 1 >> 16), CMP_EQ: slot != slot[word 1], CMP_NZ2: slot == 0 and slot[word 1] == 0  raise the bits
//   of word 2 in the lane's result flags: bit 0 the verdict (a check program's boolean, a GT element's rejection), bit 1 /
//   bit 2 the G2 operand of pair A / B is outside the order-r subgroup (honoured for the operands Args::g2_member names)
//   GCLOAD: constant area entries [out field .. + 3] <- TABLE[(word 1 & 0xffff) + (word 1 >> 16) * r .. + 3], r = the
//   running repetition of the instruction's block: the constants of a loop body that change from one pass to the next
//   (the lines of a fixed point).  Product terms then read them like any constant -- the multiply-add block has no
//   third operand path (a term kind that read the table directly cost the OTHER programs 6 %: DESIGN.md 4a)
// term words
//   w0: 0-5 x1 | 6-11 x2 | 12-17 y1 | 18-23 y2 | 24-25 kind (0 product, 1 linear: x only, 2 product with y = CONST[y1 + 64 y2])
//   w1: int8 cx1 | cx2 | cy1 | cy2          operand = c1 S[s1] + c2 S[s2]  (c2 = 0: one slot)
enum Kind : uint32_t { K_PROD = 0, K_LIN = 1, K_PROD_CONST = 2 };

struct Sched {
    uint32_t start, len, repeat, pad;
};

// Arguments of one launch.
struct Args {
    const uint32_t* prog;    // [instructions][WAVES][REC_WORDS]
    const Sched* sched;      // blocks of instructions and their repeat counts, executed in order
    uint32_t nsched;
    const int32_t* consts;   // [index][16] balanced limbs (Montgomery form unless the program says otherwise); the kernel
                             // copies them into LDS (MAX_CONSTS)
    uint32_t nconsts;
    const uint32_t* in;      // inputs: [input index][pairing][WORDS_IN] packed Montgomery words of the per-lane field code
    const uint8_t* pst;      // [operand][pairing] status bytes of the operand kernel: bits 0-6 UnmarshalBinary status, bit 7
                             // the point is at infinity; operands 0, 1 = pair A, 2, 3 = pair B
    uint32_t npst;
    uint8_t* status;         // out, may be null: first non-zero operand status per pairing
    uint8_t* out;            // GT bytes or result booleans
    size_t n;                // pairings
    uint32_t out_stride;     // bytes per pairing in `out`
    uint32_t check;          // 1: the program ends in IS_ONE and `out` takes one boolean per pairing
    uint32_t* gspill;        // global scratch: [workgroup][global slot][wave][slot image]
    uint32_t ngslots;
    const int32_t* gconsts;  // [index][16] the program's table of per-repetition constants (OP_GCLOAD), global memory
    const uint32_t* expo;    // [pairing][F::NW] programs with store mask 3 (GT exponentiation): the element's exponent, 8
                             // little-endian words at the head of its row
    uint32_t g2_member;      // bit k: operand k is a G2 point that was decoded without its r-torsion test; the program's
                             // verdict on it (result-flag bit 1 for operands 0-1, bit 2 for operands 2-3) stands in:
                             // status 2 and a rejected pairing, exactly as if the operand kernel had reported it
    uint8_t* redo_out;       // may be null.  [pairing] <- 1 when the program left result-flag bit 8 CLEAR (a product-form
                             // check whose joint Miller value was zero: gen_tower_vm.py FLAG_MILLER_NONZERO), else 0
    const uint8_t* only;     // may be null.  The launch computes and stores ONLY the pairings marked here (the redo_out
                             // of an earlier launch); a batch of 64 without a marked pairing is skipped whole
};

template <class F>
struct Lds {
    static constexpr int N = F::N;
    static constexpr int NQ = N / 4, TW = N - 4 * NQ;  // 16-byte quads + a tail of 0 or 2 words
    static_assert(TW == 0 || TW == 2, "limb count must be 0 or 2 mod 4");
    static constexpr int SLOT_WORDS = N * LANES;
    __device__ static void load(int32_t (&v)[N], const uint32_t* lds, uint32_t slot, int lane) {
        const uint32_t* b = lds + slot * SLOT_WORDS;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int4 t = *reinterpret_cast<const int4*>(b + (q * LANES + lane) * 4);
            v[4 * q] = t.x;
            v[4 * q + 1] = t.y;
            v[4 * q + 2] = t.z;
            v[4 * q + 3] = t.w;
        }
        if constexpr (TW == 2) {
            const int2 t = *reinterpret_cast<const int2*>(b + NQ * LANES * 4 + lane * 2);
            v[4 * NQ] = t.x;
            v[4 * NQ + 1] = t.y;
        }
    }
    __device__ static void store(uint32_t* lds, uint32_t slot, int lane, const int32_t (&v)[N]) {
        uint32_t* b = lds + slot * SLOT_WORDS;
#pragma unroll
        for (int q = 0; q < NQ; q++)
            *reinterpret_cast<int4*>(b + (q * LANES + lane) * 4) = make_int4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        if constexpr (TW == 2) *reinterpret_cast<int2*>(b + NQ * LANES * 4 + lane * 2) = make_int2(v[4 * NQ], v[4 * NQ + 1]);
    }
};

// sign extension from a W-bit digit (W = F::W: 28 bits, 27 for bn254 -- gen_tower_vm.py says why)
template <int W>
KYB_HD int32_t sext(uint32_t x) { return (int32_t)(x << (32 - W)) >> (32 - W); }

// v = c1 S[s1] + c2 S[s2]; the coefficients are wave-uniform, so the branches are scalar
template <class F>
__device__ __forceinline__ void operand(int32_t (&v)[F::N], const uint32_t* lds, uint32_t s1, int c1, uint32_t s2, int c2,
                                        int lane) {
    constexpr int N = F::N;
    Lds<F>::load(v, lds, s1, lane);
    if (c2 == 0) {
        if (c1 != 1) {
#pragma unroll
            for (int i = 0; i < N; i++) v[i] *= c1;
        }
        return;
    }
    int32_t b[N];
    Lds<F>::load(b, lds, s2, lane);
    if (c1 == 1 && c2 == 1) {
#pragma unroll
        for (int i = 0; i < N; i++) v[i] += b[i];
    } else if (c1 == 1 && c2 == -1) {
#pragma unroll
        for (int i = 0; i < N; i++) v[i] -= b[i];
    } else {
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = v[i] * c1 + b[i] * c2;
    }
}

// Columns t[N .. 2N-1] (after the reduction) or any N columns -> balanced W-bit limbs.  The top limb absorbs the last
// carry (the generator bounds the value so that it fits).
template <int N, int W>
__device__ __forceinline__ void normalise(int32_t (&r)[N], const int64_t* t) {
    int64_t carry = 0;
#pragma unroll
    for (int c = 0; c < N - 1; c++) {
        const int64_t v = t[c] + carry;
        carry = (v + (int64_t(1) << (W - 1))) >> W;
        r[c] = sext<W>((uint32_t)v);
    }
    r[N - 1] = (int32_t)(t[N - 1] + carry);
}

// t[0 .. 2N-2] (+ a spare t[2N-1] = 0): signed 64-bit columns of a double-width value T, |T| < R p * 2^8.  Leaves
// T R^-1 mod p (|.| < |T| / R + p/2 + small) in t[N .. 2N-1].
template <class F>
__device__ __forceinline__ void mont_reduce(int64_t (&t)[2 * F::N]) {
    constexpr int N = F::N, W = F::W;
    // F::OPAQUE_P: the digits of p as opaque scalars.  Left as literals, a digit that happens to be +-2^k (BLS12-381
    // has one) is strength-reduced into a 64-bit shift and a subtract-with-borrow pair -- five instructions where one
    // multiply-add with an SGPR operand does (-1.5 % per pairing); a field without such a digit is better off with
    // the literals (+1.1 % on the BN fields, same-box A/B)
    int32_t pd[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        pd[j] = F::P[j];
        if constexpr (F::OPAQUE_P) asm volatile("" : "+s"(pd[j]));
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int32_t m = sext<W>((uint32_t)t[i] * F::NINV);
#pragma unroll
        for (int j = 0; j < N; j++) t[i + j] += (int64_t)m * pd[j];
        t[i + 1] += t[i] >> W;  // exact: the low W bits are zero
    }
}

// Canonical words of a bounded balanced value: v + 4p in (0, 8p), as N + 1 unsigned W-bit digits, brought into [0, p)
// by three conditional subtractions (4p, 2p, p), then packed into NW 32-bit words.  Requires |v| < 4p.
template <class F>
__device__ void canon_words(uint32_t (&w)[F::NW], const int32_t (&l)[F::N]) {
    constexpr int N = F::N, NW = F::NW, ND = N + 1, W = F::W;
    constexpr uint32_t MASK = (1u << W) - 1;
    uint32_t d[ND];
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int64_t x = (int64_t)l[i] + (int64_t)F::P4[i] + carry;
        d[i] = (uint32_t)x & MASK;
        carry = x >> W;
    }
    d[N] = (uint32_t)(carry + (int64_t)F::P4[N]);
#pragma unroll
    for (int s = 2; s >= 0; s--) {
        uint32_t y[ND];
        int32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < ND; i++) {
            const uint32_t k = s == 2 ? F::P4[i] : (s == 1 ? F::P2[i] : F::P1[i]);
            const int32_t z = (int32_t)d[i] - (int32_t)k - borrow;
            y[i] = (uint32_t)z & MASK;
            borrow = (z >> 31) & 1;
        }
        const uint32_t keep = 0u - (uint32_t)borrow;  // all ones: d < k p, keep d
#pragma unroll
        for (int i = 0; i < ND; i++) d[i] = (d[i] & keep) | (y[i] & ~keep);
    }
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const int bit = 32 * k, jj = bit / W, o = bit - W * jj;
        uint32_t v = d[jj] >> o;
        if (jj + 1 <= N) v |= d[jj + 1] << (W - o);
        if (2 * W - o < 32 && jj + 2 <= N) v |= d[jj + 2] << (2 * W - o);
        w[k] = v;
    }
}

// NW packed words (a value in [0, 2^(32 NW))) -> N unsigned W-bit digits, normalised to balanced ones
template <class F>
__device__ void words_to_limbs(int32_t (&r)[F::N], const uint32_t (&w)[F::NW]) {
    constexpr int N = F::N, NW = F::NW, W = F::W;
    int64_t t[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        const int bit = W * j, idx = bit >> 5, sh = bit & 31;
        uint32_t x = idx < NW ? (w[idx] >> sh) : 0u;
        if (sh + W > 32 && idx + 1 < NW) x |= w[idx + 1] << (32 - sh);
        t[j] = (int64_t)(x & ((1u << W) - 1));
    }
    normalise<N, W>(r, t);
}

// The interpreter.  `Inv` supplies the base-field inversion on packed words (the per-lane field code's Kaliski inverse).
// Workgroups are persistent: workgroup b takes the batches b, b + gridDim.x, ... of 64 pairings.
// EXPO: the GT exponentiation programs -- store mask 3 is decoded, a failed comparison (OP_CMP_EQ: the membership test)
// rejects the element (zero output, status 2) instead of feeding a boolean.  A separate instantiation, so that the
// pairing kernels' code is what it was.
// REDO: the launch honours Args::redo_out / Args::only (a product-form check and its two-pairing fallback: bn_pair.inc);
// its own instantiation too, for the same reason.
template <class F, class Inv, bool EXPO = false, bool REDO = false>
__device__ void run(const Args& a, uint32_t* lds, uint32_t* misc, int32_t* clds) {
    constexpr int N = F::N;
    constexpr int SW = Lds<F>::SLOT_WORDS;
    const int lane = threadIdx.x & (LANES - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* gs = a.gspill + (size_t)blockIdx.x * a.ngslots * WAVES * SW;
    for (size_t batch = blockIdx.x; batch * LANES < a.n; batch += gridDim.x) {
        const size_t pairing = batch * LANES + lane;
        const bool valid = pairing < a.n;
        const size_t pidx = valid ? pairing : a.n - 1;  // out-of-range lanes recompute the last pairing, store nothing
        if constexpr (REDO) {
            if (a.only && !__syncthreads_or((valid && a.only[pidx]) ? 1 : 0)) continue;  // (uniform: nothing marked in this batch)
        }
        // lane flags: bit 0 pair A dead (an operand at infinity), bit 1 pair B dead, bit 7 an operand was rejected
        // mchk: the result-flag bits that reject this lane (an operand named by g2_member that decoded fine and is not
        // the point at infinity: its membership verdict is the program's)
        uint32_t fl = 0, first_st = 0, mchk = 0;
        for (uint32_t k = 0; k < a.npst; k++) {
            const uint32_t b = a.pst[(size_t)k * a.n + pidx];
            if ((b & 0x7fu) && !first_st) first_st = b & 0x7fu;
            if (b & 0x80u) fl |= k < 2 ? 1u : 2u;
            if (!b && ((a.g2_member >> k) & 1u)) mchk |= k < 2 ? 2u : 4u;
        }
        if (first_st) fl |= 0x80u;
        if (wave == 0) misc[lane] = 0;
        __syncthreads();
        // The schedule is walked one instruction AHEAD: the record of the next instruction is requested before the
        // current one executes, so its L2 latency (all twelve waves would otherwise sit on it after every barrier)
        // hides behind a few thousand multiply-adds.
        uint32_t e = 0, rep = 0;
        Sched sc = a.sched[0];
        uint32_t ins = sc.start;
        bool more = a.nsched > 0;
        uint32_t recw = a.prog[((size_t)ins * WAVES + wave) * REC_WORDS + lane];
        while (more) {
            uint32_t e2 = e, rep2 = rep, ins2 = ins + 1;
            Sched sc2 = sc;
            if (ins2 == sc.start + sc.len) {
                ins2 = sc.start;
                if (++rep2 == sc.repeat) {
                    rep2 = 0;
                    if (++e2 < a.nsched) {
                        sc2 = a.sched[e2];
                        ins2 = sc2.start;
                    }
                }
            }
            const bool more2 = e2 < a.nsched;
            const uint32_t rec_next = more2 ? a.prog[((size_t)ins2 * WAVES + wave) * REC_WORDS + lane] : 0u;
            {
                {
                    const uint32_t hdr = __builtin_amdgcn_readlane(recw, 0);
                    const uint32_t arg = __builtin_amdgcn_readlane(recw, 1);
                    const uint32_t out_slot = hdr & 63u, nterm = (hdr >> 6) & 63u, op = (hdr >> 21) & 15u;
                    int32_t r[N];
                    bool have = false;
                    if (op == OP_DOT) {
                        uint32_t ebit_word = 0, ebit = 0;  // (requested before the products, consumed after them)
                        if constexpr (EXPO) {
                            if (((hdr >> 19) & 3u) == 3u) {
                                const uint32_t desc = __builtin_amdgcn_readlane(recw, REC_WORDS - 1);
                                ebit = (desc & 0xffffu) - rep * (desc >> 16);
                                ebit_word = a.expo[pidx * F::NW + (ebit >> 5)];
                            }
                        }
                        int64_t t[2 * N];
#pragma unroll
                        for (int i = 0; i < 2 * N; i++) t[i] = 0;
#pragma unroll 1
                        for (uint32_t k = 0; k < nterm; k++) {
                            const uint32_t w0 = __builtin_amdgcn_readlane(recw, 1 + 2 * k);
                            const uint32_t w1 = __builtin_amdgcn_readlane(recw, 2 + 2 * k);
                            const uint32_t kind = (w0 >> 24) & 3u;
                            const int cx1 = (int8_t)(w1 & 0xff), cx2 = (int8_t)((w1 >> 8) & 0xff);
                            int32_t x[N];
                            operand<F>(x, lds, w0 & 63u, cx1, (w0 >> 6) & 63u, cx2, lane);
                            if (kind == K_LIN) {
#pragma unroll
                                for (int i = 0; i < N; i++) t[N + i] += (int64_t)x[i];
                                continue;
                            }
                            // ONE multiply-add block for both kinds of product: the second operand is assembled first
                            // (slots, or a constant's wave-uniform limbs), so the 2N column accumulators keep their
                            // registers whatever path the operands took
                            int32_t y[N];
                            if (kind == K_PROD) {
                                const int cy1 = (int8_t)((w1 >> 16) & 0xff), cy2 = (int8_t)(w1 >> 24);
                                operand<F>(y, lds, (w0 >> 12) & 63u, cy1, (w0 >> 18) & 63u, cy2, lane);
                            } else {
                                const int32_t* c = clds + 16 * ((w0 >> 12) & 0xfffu);
#pragma unroll
                                for (int j = 0; j < N; j++) y[j] = c[j];
                            }
#pragma unroll
                            for (int i = 0; i < N; i++) {
                                asm volatile("" : "+v"(x[i]), "+v"(y[i]));  // operands are final here: one block below
                            }
#pragma unroll
                            for (int i = 0; i < N; i++)
#pragma unroll
                                for (int j = 0; j < N; j++) t[i + j] += (int64_t)x[i] * y[j];
                        }
                        if (!((hdr >> 13) & 1u)) mont_reduce<F>(t);
                        normalise<N, F::W>(r, t + N);
                        const int scale = (hdr >> 14) & 31u;
                        if (scale > 1) {  // small factor (<= 15) on the normalised limbs, normalised again
#pragma unroll
                            for (int i = 0; i < N; i++) t[i] = (int64_t)r[i] * scale;
                            normalise<N, F::W>(r, t);
                        }
                        const uint32_t mask = (hdr >> 19) & 3u;
                        if (mask) {  // lanes whose pair is dead keep the old value of the output slot
                            int32_t old[N];
                            Lds<F>::load(old, lds, out_slot, lane);
                            bool dead = (fl >> (mask - 1)) & 1u;
                            if constexpr (EXPO) {
                                if (mask == 3u) dead = !((ebit_word >> (ebit & 31u)) & 1u);
                            }
#pragma unroll
                            for (int i = 0; i < N; i++) r[i] = dead ? old[i] : r[i];
                        }
                        have = true;
                    } else if (op == OP_GLOAD) {
                        const uint32_t* src = a.in + ((size_t)arg * a.n + pidx) * F::NW;
                        uint32_t w[F::NW];
#pragma unroll
                        for (int k = 0; k < F::NW; k++) w[k] = src[k];
                        words_to_limbs<F>(r, w);
                        have = true;
                    } else if (op == OP_GCLOAD) {  // 64 lanes x one word = four entries
                        clds[16 * out_slot + lane] = a.gconsts[16 * ((arg & 0xffffu) + rep * (arg >> 16)) + lane];
                    } else if (op == OP_CLOAD) {
                        const int32_t* c = clds + 16 * arg;
#pragma unroll
                        for (int i = 0; i < N; i++) r[i] = c[i];
                        have = true;
                    } else if (op == OP_INV) {
                        int32_t v[N];
                        Lds<F>::load(v, lds, arg & 63u, lane);
                        uint32_t w[F::NW];
                        canon_words<F>(w, v);
                        Inv::inv(w);
                        words_to_limbs<F>(r, w);
                        have = true;
                    } else if (op == OP_SPILL || op == OP_FILL) {
                        // the slot image [quad][lane][4 words] goes to / comes from global scratch as it is: 16 bytes per
                        // lane, consecutive lanes consecutive
                        uint32_t* g = gs + ((size_t)arg * WAVES + wave) * SW;
                        if (op == OP_SPILL) {
                            int32_t v[N];
                            Lds<F>::load(v, lds, out_slot, lane);
                            Lds<F>::store(g, 0, lane, v);
                        } else {
                            Lds<F>::load(r, g, 0, lane);
                            have = true;
                        }
                    } else if (op == OP_GT_STORE || op == OP_IS_ONE) {
                        // the slot holds the PLAIN coefficient (the program multiplied by the constant 1): canonical form
                        int32_t v[N];
                        Lds<F>::load(v, lds, out_slot, lane);
                        uint32_t w[F::NW];
                        canon_words<F>(w, v);
                        if (op == OP_GT_STORE) {
                            const bool one = (fl & 3u) != 0;  // an operand at infinity: e = 1
                            const bool rejected = (fl >> 7) || (misc[lane] & (mchk | (EXPO ? 1u : 0u)));
                            if (valid) {
                                uint32_t* q = reinterpret_cast<uint32_t*>(a.out + pairing * a.out_stride + (arg & 0xffffu));
                                const uint32_t is_c0 = arg >> 16;  // this coefficient is the 1 of the identity
#pragma unroll
                                for (int k = 0; k < F::NW; k++) {
                                    uint32_t x = w[F::NW - 1 - k];
                                    if (one) x = (is_c0 && k == F::NW - 1) ? 1u : 0u;
                                    if (rejected) x = 0;
                                    q[k] = __builtin_bswap32(x);
                                }
                            }
                        } else {
                            uint32_t diff = 0;
#pragma unroll
                            for (int k = 0; k < F::NW; k++) diff |= w[k] ^ (((arg >> 16) && k == 0) ? 1u : 0u);
                            if (diff) atomicOr(&misc[lane], __builtin_amdgcn_readlane(recw, 2));
                        }
                    }
                    else if (op == OP_CMP_EQ || op == OP_CMP_NZ2) {
                        int32_t v[N];
                        uint32_t wa[F::NW], wb[F::NW];
                        Lds<F>::load(v, lds, out_slot, lane);
                        canon_words<F>(wa, v);
                        Lds<F>::load(v, lds, arg & 63u, lane);
                        canon_words<F>(wb, v);
                        uint32_t diff = 0, any = 0;
#pragma unroll
                        for (int k = 0; k < F::NW; k++) {
                            diff |= wa[k] ^ wb[k];
                            any |= wa[k] | wb[k];
                        }
                        if (op == OP_CMP_EQ ? diff != 0 : any == 0) atomicOr(&misc[lane], __builtin_amdgcn_readlane(recw, 2));
                    }
                    // (Timing experiment, round 2: with every barrier compiled out -- wrong results, same instruction
                    // stream -- the BLS12-381 kernel runs 1 % faster and the bn256 one 7 %: the barriers and the waiting
                    // at them are not what separates the machine from its issue bound.)
                    if ((hdr >> 12) & 1u) __syncthreads();
                    if (have) Lds<F>::store(lds, out_slot, lane, r);
                    __syncthreads();
                }
            }
            e = e2;
            rep = rep2;
            ins = ins2;
            sc = sc2;
            recw = rec_next;
            more = more2;
        }
        bool store = wave == 0 && valid;
        if constexpr (REDO) store = store && !(a.only && !a.only[pairing]);
        if (store) {
            const uint32_t mf = misc[lane];
            if constexpr (REDO) {
                if (a.redo_out) a.redo_out[pairing] = (mf & 8u) ? 0 : 1;
            }
            if (a.check) a.out[pairing * a.out_stride] = (!(mf & (1u | mchk)) && !(fl >> 7)) ? 1 : 0;
            if (a.status) {
                // the first operand, in argument order, that UnmarshalBinary would have refused: a decode status of the
                // operand kernel, or status 2 where the program found a G2 operand outside the subgroup
                uint32_t stv = 0;
                for (uint32_t k = 0; k < a.npst && !stv; k++) {
                    const uint32_t b = a.pst[(size_t)k * a.n + pidx];
                    stv = b & 0x7fu;
                    if (!b && ((a.g2_member >> k) & 1u) && (mf & (k < 2 ? 2u : 4u))) stv = 2u;
                }
                if (EXPO && !stv && (mf & 1u)) stv = 2u;
                a.status[pairing] = (uint8_t)stv;
            }
        }
        __syncthreads();
    }
}

}  // namespace tvm
}  // namespace kyb
