// G1Elt / G2Elt.UnmarshalBinary (kilic/g1.go:127-131, g2.go) and .Hash of LARGE batches: the per-lane kernels of bls12381.hip
// (pairing_abi.cuh) compiled once more, in a translation unit whose every kernel is on a two-wave register budget.
//
// Out-of-line device functions take the loosest budget of the kernels that reach them; bls12381.hip holds kernels that
// want 512 registers (one wave per SIMD by design: the cooperating-lane kernels, the ladders of a half-empty chip), so
// its unmarshal kernels come out at 293 / 505 registers -- right for a batch of one wave per SIMD, wrong for one that
// could have two in flight.  Same box, 2^20 points: G1 3.54 -> 4.33e7/s, G2 2.21 -> 2.57e7/s with the kernels below
// (256 registers); at 2^16 points the loose ones are 1-2 % ahead (profiles/r04_tu_wave_budgets.json), hence the
// threshold in bls12381_lvm.cuh unmarshal_small: two waves per SIMD.
#ifndef KYB_TU_WAVES
#define KYB_TU_WAVES 2
#endif
#include "bls12381_h2c.cuh"
#include "context.h"

namespace kyb {
namespace bls {

__global__ __launch_bounds__(64, 2) void bls12381_g1_unmarshal_w2_kernel(size_t n, const uint8_t* __restrict__ pts, uint8_t* __restrict__ out,
                                                                          uint8_t* __restrict__ status, uint32_t flags) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = g1_unmarshal_wire(out + g1_out_size(flags) * idx, pts + g1_wire_size(flags) * idx, flags);
    if (status) status[idx] = (uint8_t)st;
}
__global__ __launch_bounds__(64, 2) void bls12381_g2_unmarshal_w2_kernel(size_t n, const uint8_t* __restrict__ pts, uint8_t* __restrict__ out,
                                                                          uint8_t* __restrict__ status, uint32_t flags) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = g2_unmarshal_wire(out + g2_out_size(flags) * idx, pts + g2_wire_size(flags) * idx, flags);
    if (status) status[idx] = (uint8_t)st;
}

// G1Elt.Hash / G2Elt.Hash (kilic/g1.go:161-170, g2.go; bls12381_h2c.hip's kernels) the same way: 2^18 messages
// 12.3 -> 10.5 ms (G1), 35.6 -> 30.2 ms (G2); at 2^16 the loose kernels (295 / 512 registers) are 3-5 % ahead.
__global__ __launch_bounds__(64, 2) void bls12381_hash_g1_w2_kernel(size_t n, const uint8_t* __restrict__ msgs, size_t msg_len, DstArg dst,
                                                                     uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = hash_g1_wire(out + 48 * idx, msgs + msg_len * idx, msg_len, dst);
    if (status) status[idx] = (uint8_t)st;
}
__global__ __launch_bounds__(64, 2) void bls12381_hash_g2_w2_kernel(size_t n, const uint8_t* __restrict__ msgs, size_t msg_len, DstArg dst,
                                                                     uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = hash_g2_wire(out + 96 * idx, msgs + msg_len * idx, msg_len, dst);
    if (status) status[idx] = (uint8_t)st;
}
void launch_hash_w2(bool g2, size_t n, const uint8_t* d_msgs, size_t msg_len, const DstArg& dst, uint8_t* d_out, uint8_t* d_status, hipStream_t st) {
    const unsigned grid = (unsigned)((n + 63) / 64);
    if (g2) hipLaunchKernelGGL(bls12381_hash_g2_w2_kernel, dim3(grid), dim3(64), 0, st, n, d_msgs, msg_len, dst, d_out, d_status);
    else hipLaunchKernelGGL(bls12381_hash_g1_w2_kernel, dim3(grid), dim3(64), 0, st, n, d_msgs, msg_len, dst, d_out, d_status);
}

void launch_unmarshal_w2(bool g2, size_t n, const uint8_t* d_points, uint8_t* d_out, uint8_t* d_status, uint32_t flags, hipStream_t st) {
    const unsigned grid = (unsigned)((n + 63) / 64);
    if (g2) hipLaunchKernelGGL(bls12381_g2_unmarshal_w2_kernel, dim3(grid), dim3(64), 0, st, n, d_points, d_out, d_status, flags);
    else hipLaunchKernelGGL(bls12381_g1_unmarshal_w2_kernel, dim3(grid), dim3(64), 0, st, n, d_points, d_out, d_status, flags);
}

}  // namespace bls
}  // namespace kyb
