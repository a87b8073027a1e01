// bn254 batch kernels for gfx950 + their C-ABI entry points (stamped out by pairing_abi.cuh).
//
// Replaces pairing/bn254 (in-tree arithmetic, the bn256 package over alt_bn128's constants):
//   pointG1.Mul / pointG2.Mul        point.go:103-111, 421-429 -> curve.go:196 / twist.go:170 -> bn254_g1_mul_kernel / _g2_mul_kernel
//   (Un)MarshalBinary                point.go:127-200, 431-520 (strict: coordinates < p, G2 in the subgroup) -> fused
//   pointG1.Hash                     point.go:207-285 (Keccak-256 expand_message_xmd + Shallue-van de Woestijne) -> bn254_hash_g1_kernel
// (this translation unit: G1 / G2 scalar multiplication and hashing; pairing kernels are in bn254_pair.hip, MSM in
//  bn254_msm.hip)
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
#include "bn254.cuh"
#include "pairing_abi.cuh"
#include <string.h>

namespace kyb {
namespace bn4 {
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
}  // namespace bn4
}  // namespace kyb
KYB_DEFINE_MUL_ABI(bn254, bn4, 64, 128)

namespace kyb {
__global__ __launch_bounds__(64, KYB_TU_WAVES) void bn254_hash_g1_kernel(size_t n, const uint8_t* __restrict__ msgs, size_t msg_len, DstArg dst,
                                                           uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = bn4::hash_g1_wire(out + 64 * idx, msgs + msg_len * idx, msg_len, dst);
    if (status) status[idx] = (uint8_t)st;
}
static int bn254_make_dst(DstArg& d, const uint8_t* dst, size_t dst_len) {
    if (dst_len > 255 || (dst_len && !dst)) {
        set_error("kyb_bn254_hash_g1: the domain separation tag must be at most 255 bytes");
        return KYB_E_ARG;
    }
    memset(&d, 0, sizeof d);
    if (dst_len) memcpy(d.b, dst, dst_len);
    d.len = (uint32_t)dst_len;
    return KYB_OK;
}
}  // namespace kyb
extern "C" {
int kyb_bn254_hash_g1_dev(size_t n, const void* d_msgs, size_t msg_len, const uint8_t* dst, size_t dst_len, void* d_out,
                          void* d_status, void* stream) {
    if (n && ((!d_msgs && msg_len) || !d_out)) {
        kyb::set_error("kyb_bn254_hash_g1_dev: bad argument");
        return KYB_E_ARG;
    }
    kyb::DstArg d;
    KYB_TRY(kyb::bn254_make_dst(d, dst, dst_len));
    if (!n) return KYB_OK;
    hipLaunchKernelGGL(kyb::bn254_hash_g1_kernel, dim3(kyb::grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n,
                       (const uint8_t*)d_msgs, msg_len, d, (uint8_t*)d_out, (uint8_t*)d_status);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
int kyb_bn254_hash_g1(size_t n, const uint8_t* msgs, size_t msg_len, const uint8_t* dst, size_t dst_len, uint8_t* out,
                      uint8_t* status) {
    if (n && ((!msgs && msg_len) || !out)) {
        kyb::set_error("kyb_bn254_hash_g1: bad argument");
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
    KYB_TRY(kyb_bn254_hash_g1_dev(n, m.p, msg_len, dst, dst_len, o.p, st.p, sc_.stream()));
    KYB_TRY(o.download(out, n * 64));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
}
