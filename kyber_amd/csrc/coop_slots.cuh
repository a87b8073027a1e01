// Cooperative group arithmetic on LDS slots: the MSM's serial tail (reduce / fold kernels of msm.cuh).
// The tail is chains of dependent point operations on a few thousand points -- a lane per point runs the 16
// multiplications of an addition one after the other at a lone wave's pace.  Here FOUR lanes own a point: its
// coordinates and the formula's temporaries live in LDS slots (one field element each), every lane fetches the two
// operands of "its" product of the current level by slot number, multiplies, writes the product back, and a workgroup
// barrier separates the levels.  A Jacobian addition is 5 product levels deep (17 products) instead of 16
// multiplications, a doubling 3 instead of 7.  All lanes of the block run the same levels; what a group keeps is decided
// at the commit (`commit`: predicated steps of a double-and-add, dead groups of a fold).
//
// Both routines are force-inlined into ONE call site per kernel: as out-of-line functions they took the slots through
// a generic pointer and every LDS access became a flat_load / flat_store (and the Fp2 instantiation hung).
//
// The host build (tests/host_harness.cpp) runs the same code with four THREADS as the four lanes and a pthread barrier
// as the workgroup barrier (KYB_COOP_SYNC -> kyb::coop_host_sync): the slot schedule -- which level reads and writes
// which slot -- is checked on the CPU against the one-lane routines, rare cases included.
#pragma once
#include "curve.cuh"

#if defined(__HIPCC__)
#define KYB_COOP_SYNC() __syncthreads()
#define KYB_COOP_FN __device__ __forceinline__
#else
namespace kyb {
void coop_host_sync();  // the harness's barrier between its four lane threads
}
#define KYB_COOP_SYNC() ::kyb::coop_host_sync()
#define KYB_COOP_FN inline
#endif

namespace kyb {
// the one-lane addition behind the cooperative routine's rare case (equal x), out of line under a name of its own: only
// kernels with the cooperative tail's two-wave budget reach it, so it is compiled for that budget
template <class F>
KYB_HD_NOINLINE void jac_add_rare(Jac<F>& r, const Jac<F>& p, const Jac<F>& q) {
    jac_add_inl<F, false>(r, p, q);
}

namespace coop {

constexpr int TEMPS = 10;  // temporaries an addition needs beside its two points
template <class F>
struct alignas(16) Slot {
    F f;
};
KYB_HD int sel4(int r, int a, int b, int c, int d) { return r == 0 ? a : (r == 1 ? b : (r == 2 ? c : d)); }

// P += Q (slots P..P+2, Q..Q+2 = X, Y, Z; temporaries T..T+9; fl = two flag words of the group).  add-2007-bl with
// Z3 = 2 Z1 Z2 H as a product; infinity operands and equal x are settled at the commit (the latter by lane 0 running
// the one-lane routine on the untouched operands).  Six barriers.
template <class F>
KYB_COOP_FN void add(Slot<F>* S, uint32_t* fl, int r, int P, int Q, int T, bool commit) {
    F a, b, m, d, d2;
    a = S[sel4(r, P + 2, Q + 2, P + 1, Q + 1)].f;
    b = S[sel4(r, P + 2, Q + 2, Q + 2, P + 2)].f;
    f_mul(m, a, b);
    S[T + r].f = m;  // Z1Z1 | Z2Z2 | Y1 Z2 | Y2 Z1
    KYB_COOP_SYNC();
    a = S[sel4(r, P, Q, T + 2, T + 3)].f;
    b = S[sel4(r, T + 1, T + 0, T + 1, T + 0)].f;
    f_mul(m, a, b);
    S[T + 4 + r].f = m;  // U1 | U2 | S1 | S2
    KYB_COOP_SYNC();
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
    KYB_COOP_SYNC();
    a = S[sel4(r, T + 0, T + 4, T + 8, T + 8)].f;
    b = S[sel4(r, T + 1, T + 1, T + 0, T + 0)].f;
    f_mul(m, a, b);  // J = H I | V = U1 I | Z1 Z2 H
    f_dbl(d, m);
    if (r == 0) S[T + 5].f = m;
    if (r == 1) S[T + 7].f = m;
    if (r == 2) S[T + 9].f = d;  // Z3
    KYB_COOP_SYNC();
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
    KYB_COOP_SYNC();
    const uint32_t f0 = fl[0], f1 = fl[1];
    if (commit) {
        if (f1 & 1u) {  // P at infinity: the sum is Q
            if (r < 3) S[P + r].f = S[Q + r].f;
        } else if (f1 & 2u) {  // Q at infinity: P stays
        } else if (f0) {       // same x: the point itself or its inverse
            if (r == 0) {
                Jac<F> p, q;
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
    KYB_COOP_SYNC();
}
// P = 2 P (dbl-2009-l, three barriers; infinity and Y = 0 give Z = 0 by the formulas)
template <class F>
KYB_COOP_FN void dbl(Slot<F>* S, int r, int P, int T, bool commit) {
    F a, b, x, t1, t2, m;
    a = S[sel4(r, P, P + 1, P + 1, P + 1)].f;
    b = S[sel4(r, P, P + 1, P + 2, P + 2)].f;
    f_mul(m, a, b);
    S[T + r].f = m;  // A = X^2 | B = Y^2 | Y Z | (again)
    KYB_COOP_SYNC();
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
    KYB_COOP_SYNC();
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
    KYB_COOP_SYNC();
}

}  // namespace coop
}  // namespace kyb
