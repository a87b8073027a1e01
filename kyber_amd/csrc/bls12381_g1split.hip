// BLS12-381 G1Elt.Mul (kilic/g1.go:110-116 behind UnmarshalBinary, g1.go:127-131) for a batch that leaves HALF the SIMDs
// idle: more elements than the cooperating-lane kernel takes (bls12381_g1coop.cuh), at most half a wave per SIMD.
//
// One lane that decodes, tests and multiplies walks ~1.1e6 dependent multiply-adds.  The r-torsion test (Scott's
// criterion: two multiplications by |z|, ~30 % of the chain) and the multiplication itself need nothing from each other
// but the decoded point, so they run in DIFFERENT workgroups -- even ones decode and multiply as if the point were
// vouched for, odd ones decode and test -- on SIMDs that would otherwise have nothing to do, and a merge kernel blanks
// the results the test refuses (UnmarshalBinary failed: status byte, zero bytes out).  One more decode per element
// (+12 % instructions); same box, 2^15 elements: 4.66 -> 3.82 ms.  NOT beyond half a wave per SIMD: at 2^16 the two roles
// share a SIMD and do not overlap at all (6.23 against 5.28 ms; profiles/r04_g1_split_roles.json).  (Round 4 blamed scratch
// traffic for that; round 6 halved the traffic -- the ladder's table in a global slab, DESIGN.md section 5 item 57 -- and the
// time did not move: two such waves on one SIMD simply take turns at an issue-bound chain.)
#include "bls12381.cuh"
#include "context.h"

namespace kyb {
namespace bls {

__global__ __launch_bounds__(64, 2) void bls12381_g1_mul_split_kernel(size_t n, const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ pts,
                                                                       uint8_t* __restrict__ out, uint8_t* __restrict__ st, uint32_t flags,
                                                                       uint32_t* __restrict__ tabs) {
    const size_t idx = (size_t)(blockIdx.x >> 1) * 64 + threadIdx.x;
    if (idx >= n) return;
    const uint8_t* pt = pts + g1_wire_size(flags) * idx;
    if (blockIdx.x & 1) {
        g1_aff a;
        st[idx] = (uint8_t)g1_decode_f(a, pt, flags, 0);
    } else {
        g1_mul_wire(out + g1_out_size(flags) * idx, scalars + 32 * idx, pt, flags | KYB_F_TRUSTED(0), tabs + G1_TAB_WORDS * idx);
    }
}
static __global__ __launch_bounds__(256) void bls12381_g1_split_merge_kernel(size_t n, const uint8_t* __restrict__ st, uint8_t* __restrict__ out,
                                                                              uint8_t* __restrict__ status, uint32_t flags) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const uint8_t s = st[idx];
    if (status) status[idx] = s;
    if (s) zero_bytes(out + g1_out_size(flags) * idx, (int)g1_out_size(flags));
}

void launch_g1_mul_split(size_t n, const uint8_t* d_scalars, const uint8_t* d_points, uint8_t* d_out, uint8_t* d_st, uint8_t* d_status,
                         uint32_t flags, hipStream_t st, uint32_t* d_tabs) {
    hipLaunchKernelGGL(bls12381_g1_mul_split_kernel, dim3(2 * (unsigned)((n + 63) / 64)), dim3(64), 0, st, n, d_scalars, d_points, d_out, d_st, flags,
                       d_tabs);
    hipLaunchKernelGGL(bls12381_g1_split_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, (const uint8_t*)d_st, d_out, d_status,
                       flags);
}

}  // namespace bls
}  // namespace kyb
