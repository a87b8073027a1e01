// bls12381: multi-scalar multiplication entry points (msm.cuh pipeline over the Weierstrass adapter).
#include "bls12381.cuh"
#include "pairing_abi.cuh"
#include "msm_ws.cuh"
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
using BlsG1Msm = msm::Weierstrass<bls::fp, BlsG1Codec>;
using BlsG2Msm = msm::Weierstrass<bls::fp2, BlsG2Codec>;
}  // namespace kyb

extern "C" {
int kyb_bls12381_g1_msm(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out[48], uint8_t* status,
                         uint32_t flags) {
    return kyb::msm::run_host<kyb::BlsG1Msm>(n, scalars, points, out, status, flags);
}
int kyb_bls12381_g2_msm(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out[96], uint8_t* status,
                         uint32_t flags) {
    return kyb::msm::run_host<kyb::BlsG2Msm>(n, scalars, points, out, status, flags);
}
int kyb_bls12381_g1_msm_dev(size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
                            uint32_t flags, void* stream) {
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    return kyb::msm::run<kyb::BlsG1Msm>(ctx, n, d_scalars, d_points, d_out, d_status, (hipStream_t)stream, flags);
}
int kyb_bls12381_g2_msm_dev(size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
                            uint32_t flags, void* stream) {
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    return kyb::msm::run<kyb::BlsG2Msm>(ctx, n, d_scalars, d_points, d_out, d_status, (hipStream_t)stream, flags);
}
}
