// bn256: multi-scalar multiplication entry points (msm.cuh pipeline over the Weierstrass adapter).
#include "bn256.cuh"
#include "pairing_abi.cuh"
#include "msm_ws.cuh"
namespace kyb {
struct BnG1Codec {
    static constexpr int WIRE = 64;
    __host__ __device__ static size_t wire_size(uint32_t) { return 64; }
    __device__ static int decode(bn::g1_aff& a, const uint8_t* in, uint32_t) { return bn::g1_decode(a, in); }
    __device__ static void encode(uint8_t* out, const bn::g1_aff& a) { bn::g1_encode(out, a); }
};
struct BnG2Codec {
    static constexpr int WIRE = 128;
    __host__ __device__ static size_t wire_size(uint32_t) { return 128; }
    __device__ static int decode(bn::g2_aff& a, const uint8_t* in, uint32_t) { return bn::g2_decode(a, in); }
    __device__ static void encode(uint8_t* out, const bn::g2_aff& a) { bn::g2_encode(out, a); }
};
using BnG1Msm = msm::Weierstrass<bn::fp, BnG1Codec>;
using BnG2Msm = msm::Weierstrass<bn::fp2, BnG2Codec>;
}  // namespace kyb

extern "C" {
int kyb_bn256_g1_msm(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out[64], uint8_t* status,
                         uint32_t flags) {
    return kyb::msm::run_host<kyb::BnG1Msm>(n, scalars, points, out, status, flags);
}
int kyb_bn256_g2_msm(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out[128], uint8_t* status,
                         uint32_t flags) {
    return kyb::msm::run_host<kyb::BnG2Msm>(n, scalars, points, out, status, flags);
}
int kyb_bn256_g1_msm_dev(size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
                         uint32_t flags, void* stream) {
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    return kyb::msm::run<kyb::BnG1Msm>(ctx, n, d_scalars, d_points, d_out, d_status, (hipStream_t)stream, flags);
}
int kyb_bn256_g2_msm_dev(size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
                         uint32_t flags, void* stream) {
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    return kyb::msm::run<kyb::BnG2Msm>(ctx, n, d_scalars, d_points, d_out, d_status, (hipStream_t)stream, flags);
}
int kyb_bn256_g1_poly_eval(size_t n, const uint32_t* idx, size_t t, const uint8_t* commits, uint8_t* out, uint8_t* status,
                       uint32_t flags) {
    return kyb::msm::poly_eval_host<kyb::BnG1Msm>(n, idx, t, commits, out, status, flags);
}
int kyb_bn256_g1_poly_eval_dev(size_t n, const void* d_idx, size_t t, const void* d_commits, void* d_out, void* d_status,
                           uint32_t flags, void* stream) {
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    return kyb::msm::poly_eval_run<kyb::BnG1Msm>(ctx, n, d_idx, t, d_commits, d_out, d_status, flags, (hipStream_t)stream);
}
int kyb_bn256_g2_poly_eval(size_t n, const uint32_t* idx, size_t t, const uint8_t* commits, uint8_t* out, uint8_t* status,
                       uint32_t flags) {
    return kyb::msm::poly_eval_host<kyb::BnG2Msm>(n, idx, t, commits, out, status, flags);
}
int kyb_bn256_g2_poly_eval_dev(size_t n, const void* d_idx, size_t t, const void* d_commits, void* d_out, void* d_status,
                           uint32_t flags, void* stream) {
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    return kyb::msm::poly_eval_run<kyb::BnG2Msm>(ctx, n, d_idx, t, d_commits, d_out, d_status, flags, (hipStream_t)stream);
}
}
