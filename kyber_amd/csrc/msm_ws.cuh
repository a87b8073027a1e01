// MSM curve adapter (see msm.cuh) for the short-Weierstrass groups: G1 / G2 of BLS12-381 and bn256.
// Codec supplies the suite's wire format: WIRE, decode(Aff<F>&, const uint8_t*) -> status,
// encode(uint8_t*, const Aff<F>&).
#pragma once
#include "curve.cuh"
#include "msm.cuh"

namespace kyb {
namespace msm {

template <class F, class Codec>
struct Weierstrass {
    struct Aff {
        F x, y;
        uint32_t inf;
    };
    using Acc = Jac<F>;
    static constexpr int WIRE = Codec::WIRE, OUT = Codec::WIRE;
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
        jac_madd(acc, acc, p.x, y, p.inf != 0);
    }
    __device__ static void add(Acc& r, const Acc& a, const Acc& b) { jac_add(r, a, b); }
    __device__ static void dbl(Acc& r, const Acc& a) { jac_dbl(r, a); }
    __device__ static void encode(uint8_t* out, const Acc& a) {
        kyb::Aff<F> t;
        jac_to_aff(t, a);
        Codec::encode(out, t);
    }
};

}  // namespace msm
}  // namespace kyb
