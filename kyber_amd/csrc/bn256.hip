// bn256 batch kernels for gfx950 + their C-ABI entry points (stamped out by pairing_abi.cuh).
//
// Replaces pairing/bn256 (in-tree arithmetic):
//   pointG1.Mul / pointG2.Mul        point.go:154,405 -> curve.go:189 / twist.go:162  -> bn256_g1_mul_kernel / _g2_mul_kernel
//   Suite.Pair                       suite.go:97 -> optate.go:266                     -> bn256_pair_kernel
//   Suite.ValidatePairing            suite.go:105-107 (two pairings + Equal)          -> bn256_pair_check_kernel
//   (Un)MarshalBinary                point.go:170-238, 423-499, 630-662               -> fused into every kernel
// (this translation unit: G1 / G2 scalar multiplication; pairing kernels are in bn256_pair.hip, MSM in
//  bn256_msm.hip -- split only so that the three compile in parallel)
// Every kernel of this unit on a two-wave register budget (hd.h KYB_TU_WAVES): the out-of-line field and group code takes
// the loosest budget of the kernels that reach it, and with one kernel at 512 registers the G2 ladder ran at 374 (one wave
// per SIMD).  At 256 registers: 2^18 G2 multiplications 24.2 -> 18.9 ms with every operand re-validated, 18.0 -> 14.4 ms
// vouched for; G1 and the fixed-base kernels unchanged; a three-wave budget loses (21.1 ms) --
// profiles/r04_tu_wave_budgets.json.
#ifndef KYB_TU_WAVES
#define KYB_TU_WAVES 2
#define KYB_G1_MUL_WAVES 2
#define KYB_G2_MUL_WAVES 2
#endif
#include "bn256.cuh"
#include "pairing_abi.cuh"

namespace kyb {
namespace bn {
// (the lane machine of bls12381_lvm.cuh has no BN programs yet: every element goes to the per-lane kernels)
inline int lvm_mul(bool, size_t, const uint8_t*, const uint8_t*, size_t, uint8_t*, uint8_t*, uint32_t, hipStream_t, const uint8_t** only, bool* handled) {
    *only = nullptr;
    *handled = false;
    return KYB_OK;
}
inline int unmarshal_small(bool, size_t, const uint8_t*, uint8_t*, uint8_t*, uint32_t, hipStream_t, bool* handled) {
    *handled = false;
    return KYB_OK;
}
}  // namespace bn
}  // namespace kyb
KYB_DEFINE_MUL_ABI(bn256, bn, 64, 128)

// ---- pointG1.Hash (pairing/bn256/point.go:261-313): the step before the pairing check in sign/bls Verify
namespace kyb {
__global__ __launch_bounds__(64, KYB_TU_WAVES) void bn256_hash_g1_kernel(size_t n, const uint8_t* __restrict__ msgs, size_t msg_len,
                                                           uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = bn::hash_g1_wire(out + 64 * idx, msgs + msg_len * idx, msg_len);
    if (status) status[idx] = (uint8_t)st;
}
}  // namespace kyb
extern "C" {
int kyb_bn256_hash_g1_dev(size_t n, const void* d_msgs, size_t msg_len, void* d_out, void* d_status, void* stream) {
    if (n && ((!d_msgs && msg_len) || !d_out)) {
        kyb::set_error("kyb_bn256_hash_g1_dev: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    hipLaunchKernelGGL(kyb::bn256_hash_g1_kernel, dim3(kyb::grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n,
                       (const uint8_t*)d_msgs, msg_len, (uint8_t*)d_out, (uint8_t*)d_status);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
int kyb_bn256_hash_g1(size_t n, const uint8_t* msgs, size_t msg_len, uint8_t* out, uint8_t* status) {
    if (n && ((!msgs && msg_len) || !out)) {
        kyb::set_error("kyb_bn256_hash_g1: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    kyb::StageScope sc_(ctx);
    kyb::StageBuf m, o, st;
    KYB_TRY(m.upload(msgs, n * msg_len));
    KYB_TRY(o.alloc(n * 64));
    KYB_TRY(st.alloc(n));
    KYB_TRY(kyb_bn256_hash_g1_dev(n, m.p, msg_len, o.p, st.p, nullptr));
    KYB_TRY(o.download(out, n * 64));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
}
