// Lane machine: G1 / G2 scalar multiplication of the pairing suites, one independent WAVE per 64 lanes.
//
// Replaces the per-lane formulation of round 1 (bls12381.cuh g1_mul_glv / g2_mul_gls behind kilic/g1.go:110-116
// G1Elt.Mul and kilic/g2.go G2Elt.Mul -> MulScalarBig): Jacobian routines out of line at 512 registers, window table
// and digits in private scratch indexed by the digit -- one wave per SIMD, 5-15 GB of scratch traffic per 65 536
// elements, 0.40 of the integer multiply-add peak.  Here:
//
//   * the instruction of the cooperative tower machine (tower_vm.cuh) -- out = MontReduce(sum X_t Y_t + R sum Z_t) over
//     slots of 14 (10) signed 28-bit limbs, operands c1 S[a] + c2 S[b], lazy sums, no conditional subtraction -- run
//     by every wave on its OWN 64 lanes: no barriers, the program is data (gen_lane_vm.py), ~10 KB of code;
//   * G1: a lane is a point.  G2: lanes 2l, 2l+1 hold the real / imaginary halves of every Fp2 coordinate of point l --
//     an Fp2 product is two base-field products per lane (the partner's halves come from its LDS row or by DPP), an
//     Fp2 square is ONE ((a0 + a1)(a0 - a1) | 2 a0 a1), so a G2 point is two G1-sized lanes and 65 536 G2 elements are
//     two waves per SIMD instead of one;
//   * six slots per lane: five in LDS ([slot][limb quad][lane], 17.9 KB per wave: eight waves per CU), one in
//     registers (a second register slot tipped the pair kernel over 256 registers: 140 spilled words per record); the window table (odd multiples, affine, all psi / phi images precomputed) in global memory in the
//     lane's own limb format, read with the lane's digit; the digits are bytes prepared by the prep kernel;
//   * regular signed odd digits: every lane executes the same records whatever its scalar; an exceptional addition
//     leaves Z = 0, which is flagged at the end and recomputed by the per-lane code (gen_lane_vm.py says why that is
//     complete).
#pragma once
#include "hd.h"
#include "tower_vm.cuh"

namespace kyb {
namespace lvm {

constexpr int LANES = 64;
constexpr int REC_WORDS = 64;
constexpr int NL = 5;       // slots 0 .. NL-1 live in LDS, slot NL in registers
constexpr int NSLOTS = 6;
constexpr int TAB_WORDS = 16;  // one table coordinate: N limbs padded to 64 bytes

enum Op : uint32_t { OP_DOT = 0, OP_IN = 1, OP_OUTW = 2, OP_INV = 3, OP_TSTORE = 4, OP_SELDIGIT = 5, OP_CTRSET = 6, OP_CTRADD = 7, OP_ZFLAG = 8 };
// header: 0-3 op | 4-6 out slot | 7-11 terms | 12 raw (no products, no reduction) | 13 negate the result on odd lanes
// term w0: 0-2 x1 | 3-5 x2 | 6-8 y1 | 9-11 y2 | 12-23 constant index / table coordinate | 24-27 kind | 28 table value
//          takes the digit's sign | 29 static table entry (w1 byte 2) instead of the digit's
// term w1: int8 cx1 | cx2 | cy1 | cy2
enum Kind : uint32_t { K_MUL = 0, K_LIN = 1, K_MULC = 2, K_LINC = 3, K_MULT = 4, K_LINT = 5, K2_MUL = 6, K2_SQR = 7, K2_MULT = 8, K2_MULC = 9, K2_NORM = 10, K_SQR = 11 };
// K_SQR: x^2 in the base field by the symmetric product (N (N + 1) / 2 multiply-adds) when it is the record's first term

struct Sched {
    uint32_t start, len, repeat, pad;
};

struct Args {
    const uint32_t* prog;    // [records][REC_WORDS]
    const Sched* sched;
    uint32_t nsched;
    const int32_t* consts;   // [index][16] balanced limbs
    const uint32_t* in;      // [input][lane][NW] plain canonical words (OP_IN)
    uint32_t* out;           // [output][lane][NW] plain canonical words (OP_OUTW)
    const uint8_t* digits;   // [lane][dstride]: table index | sign << 7
    uint32_t dstride;
    int32_t* table;          // [lane][nentry][ncoord][TAB_WORDS]
    uint32_t nentry, ncoord;
    uint8_t* zflag;          // [lane]: OP_ZFLAG found the slot zero
    size_t nlanes;
    int32_t* trace;          // debugging: lanes 0, 1 of the first wave store every result here ([record][2][16]); else null
};

template <class F>
struct Lds {
    static constexpr int N = F::N;
    static constexpr int NQ = N / 4, TW = N - 4 * NQ;
    static_assert(TW == 0 || TW == 2, "limb count must be 0 or 2 mod 4");
    static constexpr int SLOT_WORDS = N * LANES;
    __device__ static __forceinline__ void load(int32_t (&v)[N], const uint32_t* lds, uint32_t slot, int lane) {
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
    __device__ static __forceinline__ void store(uint32_t* lds, uint32_t slot, int lane, const int32_t (&v)[N]) {
        uint32_t* b = lds + slot * SLOT_WORDS;
#pragma unroll
        for (int q = 0; q < NQ; q++)
            *reinterpret_cast<int4*>(b + (q * LANES + lane) * 4) = make_int4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        if constexpr (TW == 2) *reinterpret_cast<int2*>(b + NQ * LANES * 4 + lane * 2) = make_int2(v[4 * NQ], v[4 * NQ + 1]);
    }
};

template <int N>
struct RegSlots {
    int32_t r0[N];
};

// the partner lane's copy of a register value (lanes 2l <-> 2l + 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ int32_t partner(int32_t v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true); }

// v = S[id] of this lane, or of its partner where `sw` (per lane) says so; `pairk` (uniform): the kind can ask for it
template <class F>
__device__ __forceinline__ void slot_get(int32_t (&v)[F::N], const uint32_t* lds, const RegSlots<F::N>& rs, uint32_t id, int lane, bool pairk,
                                         bool sw) {
    constexpr int N = F::N;
    if (id < NL) {
        Lds<F>::load(v, lds, id, pairk ? (lane ^ (sw ? 1 : 0)) : lane);
        return;
    }
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = rs.r0[i];
    if (pairk) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int32_t o = partner(v[i]);
            v[i] = sw ? o : v[i];
        }
    }
}

// v = c1 S[s1] + c2 S[s2]; the coefficients are wave-uniform, so the branches are scalar
template <class F>
__device__ __forceinline__ void operand(int32_t (&v)[F::N], const uint32_t* lds, const RegSlots<F::N>& rs, uint32_t s1, int c1, uint32_t s2, int c2,
                                        int lane, bool pairk, bool sw) {
    constexpr int N = F::N;
    slot_get<F>(v, lds, rs, s1, lane, pairk, sw);
    if (c2 == 0) {
        if (c1 != 1) {
#pragma unroll
            for (int i = 0; i < N; i++) v[i] *= c1;
        }
        return;
    }
    int32_t b[N];
    slot_get<F>(b, lds, rs, s2, lane, pairk, sw);
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

template <class F>
__device__ __forceinline__ void table_get(int32_t (&v)[F::N], const int32_t* base) {
    constexpr int N = F::N;
    const int4* q = reinterpret_cast<const int4*>(base);
#pragma unroll
    for (int k = 0; k < (N + 3) / 4; k++) {
        const int4 t = q[k];
        if (4 * k < N) v[4 * k] = t.x;
        if (4 * k + 1 < N) v[4 * k + 1] = t.y;
        if (4 * k + 2 < N) v[4 * k + 2] = t.z;
        if (4 * k + 3 < N) v[4 * k + 3] = t.w;
    }
}

// The interpreter: one wave, 64 lanes.  PAIR: lanes 2l / 2l+1 are the halves of one Fp2-coordinate point.
// `Inv` supplies the base-field inversion on packed words (mont.cuh fp_inv), as in the tower machine.
template <class F, class Inv, bool PAIR>
__device__ void run(const Args& a, uint32_t* lds) {
    constexpr int N = F::N;
    const int lane = threadIdx.x & (LANES - 1);
    const size_t gl0 = (size_t)blockIdx.x * LANES + lane;
    const bool valid = gl0 < a.nlanes;
    // an out-of-range lane recomputes an in-range one (of its own parity in pair mode) and stores nothing
    const size_t gl = valid ? gl0 : (PAIR ? a.nlanes - 2 + (gl0 & 1) : a.nlanes - 1);
    const bool odd = PAIR && (lane & 1);
    RegSlots<N> rs;
#pragma unroll
    for (int i = 0; i < N; i++) rs.r0[i] = 0;
    uint32_t tsel = 0;
    int32_t tneg = 0;
    int widx = 0;
    uint32_t ntrace = 0;
    const size_t lane_tab = (size_t)a.nentry * a.ncoord * TAB_WORDS;

    uint32_t e = 0, rep = 0;
    Sched sc = a.sched[0];
    uint32_t ins = sc.start;
    bool more = a.nsched > 0;
    uint32_t recw = a.prog[(size_t)ins * REC_WORDS + lane];
    while (more) {
        // The current record is consumed FIRST (its header read: the wait for its load lands here, with nothing else in
        // flight) and only then is the next record requested, so that request stays in flight behind the whole record.
        // Written the other way round -- request, then header -- the compiler's wait for the header covered the request
        // just issued as well (vmcnt(0) at the loop head): every record paid an L2 round trip, ~1000 cycles, which is
        // what an empty record cost in tools/lvm_microbench.py.
        const uint32_t hdr = __builtin_amdgcn_readlane(recw, 0);
        const uint32_t arg = __builtin_amdgcn_readlane(recw, 1);
        __builtin_amdgcn_sched_barrier(0);
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
        const uint32_t rec_next = more2 ? a.prog[(size_t)ins2 * REC_WORDS + lane] : 0u;
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t op = hdr & 15u, out_slot = (hdr >> 4) & 7u, nterm = (hdr >> 7) & 31u;
        int32_t r[N];
        bool have = false;
        if (op == OP_DOT) {
            // The columns are cleared where the record's first term is consumed, in the same block as its multiply-adds:
            // the first touch of every column then takes a literal zero addend instead of 2N cleared registers
            // (56 moves per record, 7 % of its instructions at one or two products per reduction).
            int64_t t[2 * N];
            bool fresh = true;
#pragma unroll 1
            for (uint32_t k = 0; k < nterm; k++) {
                const uint32_t w0 = __builtin_amdgcn_readlane(recw, 2 + 2 * k);
                const uint32_t w1 = __builtin_amdgcn_readlane(recw, 3 + 2 * k);
                const uint32_t kind = (w0 >> 24) & 15u, aux = (w0 >> 12) & 0xfffu;
                const int cx1 = (int8_t)(w1 & 0xff), cx2 = (int8_t)((w1 >> 8) & 0xff);
                const int cy1 = (int8_t)((w1 >> 16) & 0xff), cy2 = (int8_t)(w1 >> 24);
                // table operands: the entry is the digit's (per lane) or static (w1 byte 2)
                const uint32_t ent = ((w0 >> 29) & 1u) ? (uint32_t)(cy1 & 0xff) : tsel;
                const int32_t sgn = ((w0 >> 28) & 1u) ? tneg : 0;
                if (kind == K_LIN || kind == K_LINC || kind == K_LINT) {
                    int32_t x[N];
                    if (kind == K_LIN) {
                        operand<F>(x, lds, rs, w0 & 7u, cx1, (w0 >> 3) & 7u, cx2, lane, false, false);
                    } else if (kind == K_LINC) {
                        const int32_t* c0 = a.consts + 16 * aux;
#pragma unroll
                        for (int i = 0; i < N; i++) x[i] = (PAIR ? (odd ? c0[16 + i] : c0[i]) : c0[i]) * cx1;
                    } else {
                        table_get<F>(x, a.table + gl * lane_tab + ((size_t)ent * a.ncoord + aux) * TAB_WORDS);
#pragma unroll
                        for (int i = 0; i < N; i++) x[i] = (x[i] ^ sgn) - sgn;
                    }
                    if (fresh) {
#pragma unroll
                        for (int i = 0; i < N; i++) {
                            t[i] = 0;
                            t[N + i] = (int64_t)x[i];
                        }
                        fresh = false;
                    } else {
#pragma unroll
                        for (int i = 0; i < N; i++) t[N + i] += (int64_t)x[i];
                    }
                    continue;
                }
                const bool two = PAIR && (kind == K2_MUL || kind == K2_MULT || kind == K2_MULC || kind == K2_NORM);
                const uint32_t passes = two ? 2 : 1;
#pragma unroll 1
                for (uint32_t p = 0; p < passes; p++) {
                    int32_t X[N], Y[N];
                    // Fp2 product, pass 0 | 1:  even lanes xs ys | - xp yp,  odd lanes xp ys | xs yp
                    const bool pk = PAIR && kind >= K2_MUL;
                    const bool xsw = kind == K2_NORM ? (p != 0) : (odd != (p != 0));
                    operand<F>(X, lds, rs, w0 & 7u, cx1, (w0 >> 3) & 7u, cx2, lane, pk && kind != K2_SQR, xsw);
                    if (kind == K_MUL || kind == K2_MUL) {
                        operand<F>(Y, lds, rs, (w0 >> 6) & 7u, cy1, (w0 >> 9) & 7u, cy2, lane, pk, p != 0);
                    } else if (kind == K_SQR) {
#pragma unroll
                        for (int i = 0; i < N; i++) Y[i] = X[i];
                    } else if (kind == K_MULC) {
                        const int32_t* c = a.consts + 16 * aux;
#pragma unroll
                        for (int j = 0; j < N; j++) Y[j] = c[j];
                    } else if (kind == K2_MULC) {
                        const int32_t* c = a.consts + 16 * aux;
                        const bool second = odd != (p != 0);  // pass 0 takes the lane's own half of the constant
#pragma unroll
                        for (int j = 0; j < N; j++) Y[j] = second ? c[16 + j] : c[j];
                    } else if (kind == K_MULT || kind == K2_MULT) {
                        const size_t g = (pk && p) ? (gl ^ 1) : gl;
                        table_get<F>(Y, a.table + g * lane_tab + ((size_t)ent * a.ncoord + aux) * TAB_WORDS);
#pragma unroll
                        for (int j = 0; j < N; j++) Y[j] = (Y[j] ^ sgn) - sgn;
                    } else if (kind == K2_SQR) {  // even: (xs + xp)(xs - xp) ; odd: (2 xs) xp
                        operand<F>(Y, lds, rs, w0 & 7u, cx1, (w0 >> 3) & 7u, cx2, lane, true, true);  // the partner's half
#pragma unroll
                        for (int i = 0; i < N; i++) {
                            const int32_t xs = X[i], xq = Y[i];
                            Y[i] = odd ? xq : xs - xq;
                            X[i] = xs + (odd ? xs : xq);
                        }
                    } else {  // K2_NORM: xs^2 + xp^2
#pragma unroll
                        for (int i = 0; i < N; i++) Y[i] = X[i];
                    }
                    if (two && kind != K2_NORM && p) {  // the second product of an even lane enters negated
                        const int32_t m = odd ? 0 : -1;
#pragma unroll
                        for (int j = 0; j < N; j++) Y[j] = (Y[j] ^ m) - m;
                    }
#pragma unroll
                    for (int i = 0; i < N; i++) asm volatile("" : "+v"(X[i]), "+v"(Y[i]));  // operands final: the blocks below
                    if (fresh) {
#pragma unroll
                        for (int i = 0; i < 2 * N; i++) t[i] = 0;
                        if (!PAIR && kind == K_SQR) {  // x_i^2 on the diagonal, (2 x_i) x_j above it
#pragma unroll
                            for (int i = 0; i < N; i++) {
                                t[2 * i] += (int64_t)X[i] * X[i];
                                const int32_t d = 2 * X[i];
#pragma unroll
                                for (int j = i + 1; j < N; j++) t[i + j] += (int64_t)d * X[j];
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < N; i++)
#pragma unroll
                                for (int j = 0; j < N; j++) t[i + j] += (int64_t)X[i] * Y[j];
                        }
                        fresh = false;
                    } else {
#pragma unroll
                        for (int i = 0; i < N; i++)
#pragma unroll
                            for (int j = 0; j < N; j++) t[i + j] += (int64_t)X[i] * Y[j];
                    }
                }
            }
            if (!((hdr >> 12) & 1u)) tvm::mont_reduce<F>(t);
            tvm::normalise<N, F::W>(r, t + N);
            if (PAIR && ((hdr >> 13) & 1u)) {
                const int32_t m = odd ? -1 : 0;
#pragma unroll
                for (int i = 0; i < N; i++) r[i] = (r[i] ^ m) - m;
            }
            have = true;
        } else if (op == OP_IN) {
            const uint32_t* src = a.in + ((size_t)arg * a.nlanes + gl) * F::NW;
            uint32_t w[F::NW];
#pragma unroll
            for (int k = 0; k < F::NW; k++) w[k] = src[k];
            tvm::words_to_limbs<F>(r, w);
            have = true;
        } else if (op == OP_OUTW || op == OP_ZFLAG) {
            int32_t v[N];
            slot_get<F>(v, lds, rs, out_slot, lane, false, false);
            uint32_t w[F::NW];
            tvm::canon_words<F>(w, v);
            if (op == OP_OUTW) {
                if (valid) {
                    uint32_t* dst = a.out + ((size_t)arg * a.nlanes + gl) * F::NW;
#pragma unroll
                    for (int k = 0; k < F::NW; k++) dst[k] = w[k];
                }
            } else {
                uint32_t any = 0;
#pragma unroll
                for (int k = 0; k < F::NW; k++) any |= w[k];
                if (valid) a.zflag[gl] = any ? 0 : 1;
            }
        } else if (op == OP_INV) {
            int32_t v[N];
            slot_get<F>(v, lds, rs, arg & 7u, lane, false, false);
            uint32_t w[F::NW];
            tvm::canon_words<F>(w, v);
            Inv::inv(w);
            tvm::words_to_limbs<F>(r, w);
            have = true;
        } else if (op == OP_TSTORE) {
            int32_t v[N];
            slot_get<F>(v, lds, rs, out_slot, lane, false, false);
            if (valid) {
                int4* dst = reinterpret_cast<int4*>(a.table + gl * lane_tab + ((size_t)(arg >> 8) * a.ncoord + (arg & 0xffu)) * TAB_WORDS);
#pragma unroll
                for (int k = 0; k < (N + 3) / 4; k++)
                    dst[k] = make_int4(v[4 * k], 4 * k + 1 < N ? v[4 * k + 1] : 0, 4 * k + 2 < N ? v[4 * k + 2] : 0, 4 * k + 3 < N ? v[4 * k + 3] : 0);
            }
        } else if (op == OP_SELDIGIT) {
            const uint32_t b = a.digits[gl * a.dstride + arg + (uint32_t)widx];
            tsel = b & 15u;
            tneg = (b >> 7) ? -1 : 0;
        } else if (op == OP_CTRSET) {
            widx = (int)arg;
        } else if (op == OP_CTRADD) {
            widx += (int)arg;
        }
        if (have) {
            if (out_slot < NL) {
                Lds<F>::store(lds, out_slot, lane, r);
            } else {
#pragma unroll
                for (int i = 0; i < N; i++) rs.r0[i] = r[i];
            }
            if (a.trace) {
                if (blockIdx.x == 0 && lane < 2) {
#pragma unroll
                    for (int i = 0; i < N; i++) a.trace[((size_t)ntrace * 2 + lane) * 16 + i] = r[i];
                }
                ntrace++;
            }
        }
        e = e2;
        rep = rep2;
        ins = ins2;
        sc = sc2;
        recw = rec_next;
        more = more2;
    }
}

}  // namespace lvm
}  // namespace kyb
