// MSM adapters that live in translation units of their own (see bls12381_msm_plain.hip for why): what the C-ABI entry
// points of bls12381_msm.hip / bn_msm.inc call.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "context.h"
namespace kyb {
#define KYB_MSM_ADAPTER(name)                                                                                                          \
    int name##_host(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out, uint8_t* status, uint32_t flags);           \
    int name##_dev(DeviceCtx* ctx, size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status, hipStream_t st, \
                   uint32_t flags);
KYB_MSM_ADAPTER(bls12381_g1_msm_plain)
KYB_MSM_ADAPTER(bls12381_g2_msm_gls)
KYB_MSM_ADAPTER(bn256_g1_msm_glv)
KYB_MSM_ADAPTER(bn254_g1_msm_glv)
#undef KYB_MSM_ADAPTER
}  // namespace kyb
