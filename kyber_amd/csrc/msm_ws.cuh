// MSM curve adapter (see msm.cuh) for the short-Weierstrass groups: G1 / G2 of BLS12-381 and bn256.
// Codec supplies the suite's wire format: WIRE, decode(Aff<F>&, const uint8_t*) -> status,
// encode(uint8_t*, const Aff<F>&).
#pragma once
#include "curve.cuh"
#include "msm.cuh"

namespace kyb {
namespace msm {

// the one-lane addition behind the cooperative routine's rare case (equal x), out of line under a name of its own: only
// kernels with the cooperative tail's two-wave budget reach it, so it is compiled for that budget
template <class F>
KYB_HD_NOINLINE void jac_add_rare(Jac<F>& r, const Jac<F>& p, const Jac<F>& q) {
    jac_add_inl<F, false>(r, p, q);
}

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
    using Piece = Xyzz<F>;
    __device__ static void piece_identity(Piece& a) { xyzz_set_inf(a); }
    __device__ static void piece_madd(Piece& acc, const Aff& p, bool neg) {
        if (p.inf) return;
        F y = p.y, ny;
        f_neg(ny, p.y);
        f_cmov(y, ny, neg);
        xyzz_madd(acc, p.x, y);
    }
    __device__ static void piece_finish(Acc& r, const Piece& a) { xyzz_to_jac(r, a); }
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
    // ---- Cooperative group arithmetic on LDS slots: the MSM's serial tail (reduce / fold / final kernels of msm.cuh).
    // The tail is chains of dependent point operations on a few thousand points -- a lane per point runs its 16
    // multiplications of an addition one after the other at a lone wave's pace.  Here FOUR lanes own a point: its
    // coordinates and the formula's temporaries live in LDS slots (one field element each), every lane fetches the
    // two operands of "its" product of the current level by slot number, multiplies, writes the product back, and a
    // workgroup barrier separates the levels.  A Jacobian addition is 5 product levels deep (17 products) instead of
    // 16 multiplications, a doubling 3 instead of 7.  All lanes of the block run the same levels; what a group keeps is
    // decided at the commit (`commit`: predicated steps of a double-and-add, dead groups of a fold).
    static constexpr int COOP_SLOTS = 1, COOP_TEMPS = 10;
    struct alignas(16) Slot {
        F f;
    };
    __device__ static int sel4(int r, int a, int b, int c, int d) { return r == 0 ? a : (r == 1 ? b : (r == 2 ? c : d)); }
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
    // P += Q (slots P..P+2, Q..Q+2 = X, Y, Z; temporaries T..T+9; fl = two flag words of the group).  add-2007-bl with
    // Z3 = 2 Z1 Z2 H as a product; infinity operands and equal x are settled at the commit (the latter by lane 0 running
    // the one-lane routine on the untouched operands).  Six barriers.
    // (Both routines are force-inlined into ONE call site per kernel: as out-of-line functions they took the slots
    // through a generic pointer and every LDS access became a flat_load / flat_store.)
    __device__ __forceinline__ static void coop_add(Slot* S, uint32_t* fl, int r, int P, int Q, int T, bool commit) {
        F a, b, m, d, d2;
        a = S[sel4(r, P + 2, Q + 2, P + 1, Q + 1)].f;
        b = S[sel4(r, P + 2, Q + 2, Q + 2, P + 2)].f;
        f_mul(m, a, b);
        S[T + r].f = m;  // Z1Z1 | Z2Z2 | Y1 Z2 | Y2 Z1
        __syncthreads();
        a = S[sel4(r, P, Q, T + 2, T + 3)].f;
        b = S[sel4(r, T + 1, T + 0, T + 1, T + 0)].f;
        f_mul(m, a, b);
        S[T + 4 + r].f = m;  // U1 | U2 | S1 | S2
        __syncthreads();
        a = S[sel4(r, T + 5, T + 7, P + 2, P + 2)].f;
        b = S[sel4(r, T + 4, T + 6, Q + 2, Q + 2)].f;
        f_sub(d, a, b);  // H | S2 - S1
        f_dbl(d2, d);    // 2H | rr
        {
            F x = a, y = b;
            f_cmov(x, d2, r < 2);
            f_cmov(y, d2, r < 2);
            f_mul(m, x, y);  // I = (2H)^2 | rr^2 | Z1 Z2
        }
        if (r == 0) {
            S[T + 0].f = d;
            S[T + 1].f = m;
            fl[0] = f_is_zero(d) ? 1u : 0u;
        }
        if (r == 1) {
            S[T + 2].f = d2;
            S[T + 3].f = m;
        }
        if (r == 2) {
            S[T + 8].f = m;
            fl[1] = (f_is_zero(a) ? 1u : 0u) | (f_is_zero(b) ? 2u : 0u);
        }
        __syncthreads();
        a = S[sel4(r, T + 0, T + 4, T + 8, T + 8)].f;
        b = S[sel4(r, T + 1, T + 1, T + 0, T + 0)].f;
        f_mul(m, a, b);  // J = H I | V = U1 I | Z1 Z2 H
        f_dbl(d, m);
        if (r == 0) S[T + 5].f = m;
        if (r == 1) S[T + 7].f = m;
        if (r == 2) S[T + 9].f = d;  // Z3
        __syncthreads();
        {
            const F RR = S[T + 3].f, J = S[T + 5].f, V = S[T + 7].f;
            F X3;
            f_sub(X3, RR, J);
            f_sub(X3, X3, V);
            f_sub(X3, X3, V);
            f_sub(d, V, X3);
            a = S[sel4(r, T + 2, T + 6, T + 2, T + 6)].f;  // rr | S1
            b = d;
            f_cmov(b, J, (r & 1) != 0);  // V - X3 | J
            f_mul(m, a, b);              // lane 0 keeps rr (V - X3) for the commit
            if (r == 1) S[T + 0].f = m;  // S1 J
            if (r == 2) S[T + 1].f = X3;
        }
        __syncthreads();
        const uint32_t f0 = fl[0], f1 = fl[1];
        if (commit) {
            if (f1 & 1u) {  // P at infinity: the sum is Q
                if (r < 3) S[P + r].f = S[Q + r].f;
            } else if (f1 & 2u) {  // Q at infinity: P stays
            } else if (f0) {       // same x: the point itself or its inverse
                if (r == 0) {
                    Acc p, q;
                    p.X = S[P].f; p.Y = S[P + 1].f; p.Z = S[P + 2].f;
                    q.X = S[Q].f; q.Y = S[Q + 1].f; q.Z = S[Q + 2].f;
                    jac_add_rare(p, p, q);
                    S[P].f = p.X; S[P + 1].f = p.Y; S[P + 2].f = p.Z;
                }
            } else {
                if (r == 0) {
                    F s1j = S[T + 0].f, y3;
                    f_dbl(s1j, s1j);
                    f_sub(y3, m, s1j);
                    S[P + 1].f = y3;
                }
                if (r == 1) S[P].f = S[T + 1].f;
                if (r == 2) S[P + 2].f = S[T + 9].f;
            }
        }
        __syncthreads();
    }
    // P = 2 P (dbl-2009-l, three barriers; infinity and Y = 0 give Z = 0 by the formulas)
    __device__ __forceinline__ static void coop_dbl_slots(Slot* S, int r, int P, int T, bool commit) {
        F a, b, x, t1, t2, m;
        a = S[sel4(r, P, P + 1, P + 1, P + 1)].f;
        b = S[sel4(r, P, P + 1, P + 2, P + 2)].f;
        f_mul(m, a, b);
        S[T + r].f = m;  // A = X^2 | B = Y^2 | Y Z | (again)
        __syncthreads();
        a = S[sel4(r, T + 1, P, T + 0, T + 0)].f;
        b = S[sel4(r, T + 1, T + 1, T + 0, T + 0)].f;
        x = a;
        f_add(t1, a, b);
        f_cmov(x, t1, r >= 1);  // B | X + B | 2A
        f_add(t2, t1, a);
        f_cmov(x, t2, r >= 2);  // . | . | 3A
        f_mul(m, x, x);
        if (r == 0) S[T + 4].f = m;  // C = B^2
        if (r == 1) S[T + 5].f = m;  // (X + B)^2
        if (r == 2) {
            S[T + 6].f = m;  // G = E^2
            S[T + 7].f = x;  // E = 3A
        }
        __syncthreads();
        {
            const F A_ = S[T + 0].f, C = S[T + 4].f, TT = S[T + 5].f, G = S[T + 6].f, E = S[T + 7].f;
            F D, X3, Y3, c8;
            f_sub(D, TT, A_);
            f_sub(D, D, C);
            f_dbl(D, D);  // 4 X Y^2
            f_dbl(t1, D);
            f_sub(X3, G, t1);
            f_sub(t1, D, X3);
            f_mul(t1, E, t1);
            f_dbl(c8, C);
            f_dbl(c8, c8);
            f_dbl(c8, c8);
            f_sub(Y3, t1, c8);
            F Z3 = S[T + 2].f;
            f_dbl(Z3, Z3);
            if (commit) {
                if (r == 0) S[P].f = X3;
                if (r == 1) S[P + 1].f = Y3;
                if (r == 2) S[P + 2].f = Z3;
            }
        }
        __syncthreads();
    }
    __device__ static void encode(uint8_t* out, const Acc& a) {
        kyb::Aff<F> t;
        jac_to_aff(t, a);
        Codec::encode(out, t);
    }
};

}  // namespace msm
}  // namespace kyb
