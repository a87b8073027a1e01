// bls12381: the G1 MSM adapter for short scalars (kyb_bls12381_g1_msm with KYB_F_SCALAR_BITS(b), b <= 160).
// A translation unit of its own: two adapters over one field in one unit share their inlined helpers, the compiler stops
// inlining them, and BOTH accumulate kernels pay (accumulate_kernel<BlsG1Msm> went from 209 registers and no scratch to
// 248 and 320 B -- 2.63 -> 3.45 ms per 2^20 points -- when BlsG1MsmPlain sat beside it: DESIGN.md section 5 item 58 j).
#include "bls12381.cuh"
#include "pairing_abi.cuh"
#include "rowfp.cuh"
#include "msm_ws.cuh"
#include "bls12381_msm_codec.cuh"
#include "msm_adapters.h"
#ifndef KYB_BLS_G1_DECODE_WAVES
#define KYB_BLS_G1_DECODE_WAVES 2
#endif
namespace kyb {
// G1 WITHOUT the split, for calls that say their scalars are short (KYB_F_SCALAR_BITS(b), b <= 160: bdn's 128-bit
// coefficients, sign/bdn/bdn.go:126-161 on a G1 signature scheme).  On halves a 128-bit k is k0 + k1 z^2 with k1 in {0, +-1}:
// two fifths of all points in ONE bucket of the second half, and the call took LONGER than with full scalars (4.85
// against 3.83 ms for 2^20 points).  Plain windows: ceil(129 / 16) = 9 visits per point instead of 16, the same tail
// kernels (limb-per-lane chains, light decode).
struct BlsG1MsmPlain : msm::Weierstrass<bls::fp, BlsG1Codec> {
    using Base = msm::Weierstrass<bls::fp, BlsG1Codec>;
    static constexpr int ROW_FINAL = 1;
    using RowC = bls::FC;
    static constexpr int DECODE_WAVES = KYB_BLS_G1_DECODE_WAVES;
    static constexpr int LIGHT_DECODE_WAVES = 4;
    __device__ static int decode_split_light(Aff (&a)[1], uint32_t (&k)[1][8], const uint8_t* pt, const uint8_t* scalar) {
        bls::g1_aff t;
        const int st = bls::g1_decode_unc_trusted(t, pt);
        a[0].x = t.x;
        a[0].y = t.y;
        a[0].inf = t.inf ? 1u : 0u;
        Base::scalar_words(k[0], scalar);
        return st;
    }
};
int bls12381_g1_msm_plain_host(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out, uint8_t* status, uint32_t flags) {
    return msm::run_host<BlsG1MsmPlain>(n, scalars, points, out, status, flags);
}
int bls12381_g1_msm_plain_dev(DeviceCtx* ctx, size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
                              hipStream_t st, uint32_t flags) {
    return msm::run<BlsG1MsmPlain>(ctx, n, d_scalars, d_points, d_out, d_status, st, flags);
}
}  // namespace kyb
