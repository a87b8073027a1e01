// MSM curve adapter (see msm.cuh) for the short-Weierstrass groups: G1 / G2 of BLS12-381 and bn256.
// Codec supplies the suite's wire format: WIRE, decode(Aff<F>&, const uint8_t*) -> status,
// encode(uint8_t*, const Aff<F>&).
#pragma once
#include "coop_slots.cuh"
#include "curve.cuh"
#include "msm.cuh"

namespace kyb {
namespace msm {

template <class T>
struct IsFp2 {
    static constexpr bool value = false;
};
template <class T>
struct IsFp2<Fp2<T>> {
    static constexpr bool value = true;
};

template <class F, class Codec>
struct Weierstrass {
    struct Aff {
        F x, y;
        uint32_t inf;
    };
    using Acc = Jac<F>;
    static constexpr int WIRE = Codec::WIRE, OUT = Codec::WIRE;
    static constexpr bool SCALAR_BE = true;                    // mod.Int wire format (group/mod/int.go:334-350)
    static constexpr uint32_t COMBINE_FLAGS = KYB_F_TRUSTED(0);  // partial points of the multi-device MSM are our own
    __host__ __device__ static size_t wire_size(uint32_t flags) { return Codec::wire_size(flags); }
    __device__ static int decode(Aff& a, const uint8_t* wire, uint32_t flags) {
        kyb::Aff<F> t;
        const int st = Codec::decode(t, wire, flags);
        a.x = t.x;
        a.y = t.y;
        a.inf = t.inf ? 1u : 0u;
        return st;
    }
    __device__ static void scalar_words(uint32_t (&k)[8], const uint8_t* wire) { words_from_be<8>(k, wire); }
    __device__ static void identity(Acc& a) { jac_set_inf(a); }
    __device__ static void madd(Acc& acc, const Aff& p, bool neg) {
        F y = p.y, ny;
        f_neg(ny, p.y);
        f_cmov(y, ny, neg);
        jac_madd_inl(acc, acc, p.x, y, p.inf != 0);
    }
    // Bucket pieces are runs of mixed additions: they accumulate in XYZZ form (curve.cuh: 8M + 2S, no selects on the
    // common path) and leave it once, at the end of the run (accumulate_kernel: 2^24 additions of a 2^20-point MSM).
    // Base fields with headroom (BLS12-381 Fp) run the piece in LIMB form (XyzzL: no packing between the products,
    // the point's sign applied inside the formula; curve.cuh XyzzSel).
    using PS = XyzzSel<F>;
    using Piece = typename PS::type;
    __device__ static void piece_identity(Piece& a) { PS::identity(a); }
    __device__ static void piece_madd(Piece& acc, const Aff& p, bool neg) {
        if (p.inf) return;
        PS::madd(acc, p.x, p.y, neg);
    }
    __device__ static void piece_finish(Acc& r, const Piece& a) { PS::finish(r, a); }
    // inline bodies: the accumulators of the pipeline's loops (bucket pieces, running sums, folds) stay in registers
    // from one group operation to the next instead of crossing a call boundary (scratch) each time
    __device__ static void add(Acc& r, const Acc& a, const Acc& b) { jac_add_inl(r, a, b); }
    __device__ static void dbl(Acc& r, const Acc& a) { jac_dbl_inl(r, a); }
    // Cooperative doubling for the latency-bound tail of the MSM (final_kernel): COOP lanes hold the same point and
    // each computes one of the independent field products of a level, exchanged through LDS `sh` (COOP slots of this
    // group) -- a doubling is 3 dependent multiplications deep instead of 2M + 5S.  Every thread of the block must
    // call it (it contains barriers); `r` is the lane's index in its group.
    static constexpr int COOP = 3;
    using Field = F;
    __device__ static void dbl_coop(Acc& s, int r, F* sh) {
        F a = s.Y, b = s.Y, m, t, E;
        f_cmov(a, s.X, r == 0);
        f_cmov(b, s.X, r == 0);
        f_cmov(b, s.Z, r == 2);
        f_mul(m, a, b);  // X^2 | Y^2 | Y Z
        sh[r] = m;
        __syncthreads();
        const F A = sh[0], B = sh[1], YZ = sh[2];
        __syncthreads();
        f_dbl(E, A);
        f_add(E, E, A);  // 3 X^2
        f_add(t, s.X, B);
        a = B;
        f_cmov(a, t, r == 1);
        f_cmov(a, E, r == 2);
        f_sqr(m, a);  // Y^4 | (X + Y^2)^2 | 9 X^4
        sh[r] = m;
        __syncthreads();
        F C = sh[0], T = sh[1];
        const F G = sh[2];
        __syncthreads();
        F D;
        f_sub(T, T, A);
        f_sub(T, T, C);
        f_dbl(D, T);  // 4 X Y^2
        f_dbl(s.Z, YZ);
        f_dbl(t, D);
        f_sub(s.X, G, t);
        f_sub(t, D, s.X);
        f_mul(t, E, t);  // the one product of the last level, computed by every lane
        f_dbl(C, C);
        f_dbl(C, C);
        f_dbl(C, C);
        f_sub(s.Y, t, C);
    }
    // ---- Cooperative group arithmetic on LDS slots (coop_slots.cuh): four lanes per point in the MSM's reduce / fold
    // kernels.  Thin wrappers so that msm.cuh stays adapter-generic.
    static constexpr int COOP_SLOTS = 1, COOP_TEMPS = coop::TEMPS;
    using Slot = coop::Slot<F>;
    __device__ static void slot_load(Slot* S, int P, const Acc* src, int r) {  // lanes 0..2 move one coordinate each
        if (r < 3) S[P + r].f = reinterpret_cast<const F*>(src)[r];
    }
    __device__ static void slot_store(Acc* dst, const Slot* S, int P, int r) {
        if (r < 3) reinterpret_cast<F*>(dst)[r] = S[P + r].f;
    }
    __device__ static void slot_identity(Slot* S, int P, int r) {
        if (r < 3) {
            F v;
            f_one(v);
            if (r == 2) f_zero(v);
            S[P + r].f = v;
        }
    }
    __device__ __forceinline__ static void coop_add(Slot* S, uint32_t* fl, int r, int P, int Q, int T, bool commit) {
        coop::add<F>(S, fl, r, P, Q, T, commit);
    }
    __device__ __forceinline__ static void coop_dbl_slots(Slot* S, int r, int P, int T, bool commit) {
        coop::dbl<F>(S, r, P, T, commit);
    }
    __device__ static void encode(uint8_t* out, const Acc& a) {
        kyb::Aff<F> t;
        jac_to_aff(t, a);
        Codec::encode(out, t);
    }
};

}  // namespace msm
}  // namespace kyb
