// The wire formats of the BLS12-381 MSM adapters (bls12381_msm.hip, bls12381_msm_plain.hip, bls12381_msm_gls.hip).
#pragma once
#include "bls12381.cuh"
namespace kyb {
struct BlsG1Codec {
    static constexpr int WIRE = 48;
    __host__ __device__ static size_t wire_size(uint32_t flags) { return bls::g1_wire_size(flags); }
    __device__ static int decode(bls::g1_aff& a, const uint8_t* in, uint32_t flags) {
        return bls::g1_decode_f(a, in, flags, 0);
    }
    __device__ static void encode(uint8_t* out, const bls::g1_aff& a) { bls::g1_encode(out, a); }
};
struct BlsG2Codec {
    static constexpr int WIRE = 96;
    __host__ __device__ static size_t wire_size(uint32_t flags) { return bls::g2_wire_size(flags); }
    __device__ static int decode(bls::g2_aff& a, const uint8_t* in, uint32_t flags) {
        return bls::g2_decode_f(a, in, flags, 0);
    }
    __device__ static void encode(uint8_t* out, const bls::g2_aff& a) { bls::g2_encode(out, a); }
};
}  // namespace kyb
