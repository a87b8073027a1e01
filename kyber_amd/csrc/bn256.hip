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
#include <stdlib.h>
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
// The same hash for large batches without the divergence of try-and-increment.  A lane needs two square-root attempts
// on average, but a wave of the kernel above waits for its unluckiest lane -- about seven attempts for 64 lanes, each a
// 254-bit power.  Here ONE wave owns HQ messages and a queue of the pending ones in LDS: every round all 64 lanes take the
// next 64 pending candidates, a failed candidate goes back into the queue as x + 1, and the wave's cost is the AVERAGE
// number of attempts (plus a short tail when the queue runs dry): 2 x HQ / 64 + ~3 rounds instead of 7.3 x HQ / 64.
template <int HQ>
__global__ __launch_bounds__(64, KYB_TU_WAVES) void bn256_hash_g1_queue_kernel(size_t n, const uint8_t* __restrict__ msgs, size_t msg_len,
                                                                 uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
    using namespace bn;
    __shared__ uint32_t xs[HQ][8];  // the current candidate x of every message (Montgomery words)
    __shared__ uint16_t q[2][HQ];   // the pending messages of this round / the next
    __shared__ uint32_t cnt[2];
    const size_t base = (size_t)blockIdx.x * HQ;
    const int m = (int)(n - base < (size_t)HQ ? n - base : (size_t)HQ), lane = (int)threadIdx.x;
    for (int j = lane; j < m; j += 64) {  // x = SHA-256(msg) mod p (point.go:286-288)
        uint32_t h[8], w[8];
        sha256(h, msgs + msg_len * (base + j), msg_len);
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = h[7 - k];
        fp x;
        fp_from_words<FC>(x, w);
#pragma unroll
        for (int k = 0; k < 8; k++) xs[j][k] = x.v[k];
        q[0][j] = (uint16_t)j;
    }
    if (lane == 0) {
        cnt[0] = (uint32_t)m;
        cnt[1] = 0;
    }
    __syncthreads();
    fp b, one;
    fp_const(b, CC::B1);
    fp_one(one);
    int cur = 0;
#pragma unroll 1
    for (int round = 0; round < 256; round++) {  // (the per-lane kernel gives up after 256 attempts too)
        const uint32_t c = cnt[cur];
        if (c == 0) break;
#pragma unroll 1
        for (uint32_t i0 = 0; i0 < c; i0 += 64) {
            const uint32_t i = i0 + (uint32_t)lane;
            if (i < c) {
                const int j = q[cur][i];
                fp x, t, y, y2;
#pragma unroll
                for (int k = 0; k < 8; k++) x.v[k] = xs[j][k];
                fp_sqr(t, x);
                fp_mul(t, t, x);
                fp_add(t, t, b);
                fp_pow_words<FC>(y, t, FC::SQRT_EXP, FC::SQRT_BITS);
                fp_sqr(y2, y);
                if (fp_eq(y2, t)) {
                    fp_encode(out + 64 * (base + j), x);
                    fp_encode(out + 64 * (base + j) + 32, y);
                    if (status) status[base + j] = (uint8_t)ST_OK;
                } else {
                    fp_add(x, x, one);
#pragma unroll
                    for (int k = 0; k < 8; k++) xs[j][k] = x.v[k];
                    q[cur ^ 1][atomicAdd(&cnt[cur ^ 1], 1u)] = (uint16_t)j;
                }
            }
        }
        __syncthreads();
        if (lane == 0) cnt[cur] = 0;
        cur ^= 1;
        __syncthreads();
    }
    const uint32_t left = cnt[cur];  // never, in practice: 256 failed rounds
    for (uint32_t i = (uint32_t)lane; i < left; i += 64) {
        const int j = q[cur][i];
        uint32_t* o = reinterpret_cast<uint32_t*>(out + 64 * (base + j));
        for (int k = 0; k < 16; k++) o[k] = 0;
        if (status) status[base + j] = (uint8_t)ST_BAD_POINT;
    }
}
// KYB_BN_HASH_QUEUE=0 keeps every batch on the per-lane kernel (A/B)
static bool hash_queue_on() {
    static const bool on = [] {
        const char* e = getenv("KYB_BN_HASH_QUEUE");
        return !(e && e[0] == '0');
    }();
    return on;
}
}  // namespace kyb
extern "C" {
int kyb_bn256_hash_g1_dev(size_t n, const void* d_msgs, size_t msg_len, void* d_out, void* d_status, void* stream) {
    if (n && ((!d_msgs && msg_len) || !d_out)) {
        kyb::set_error("kyb_bn256_hash_g1_dev: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    // large batches: a wave per 128 / 256 / 512 messages with the pending candidates queued in LDS (the pooling needs at least
    // two waves' worth of messages per wave, and the chip at least a wave per SIMD)
    if (kyb::hash_queue_on() && n >= (size_t(1) << 17)) {
        const hipStream_t st = (hipStream_t)stream;
        static const int forced = [] {  // KYB_BN_HASH_HQ = 128 / 256 / 512: messages per wave (experiments)
            const char* e = getenv("KYB_BN_HASH_HQ");
            return e ? atoi(e) : 0;
        }();
        const size_t pick = forced == 512 ? (size_t(1) << 19) : forced == 256 ? (size_t(1) << 18) : forced == 128 ? (size_t(1) << 17) : n;
        if (pick >= (size_t(1) << 19))
            hipLaunchKernelGGL(kyb::bn256_hash_g1_queue_kernel<512>, dim3((unsigned)((n + 511) / 512)), dim3(64), 0, st, n, (const uint8_t*)d_msgs,
                               msg_len, (uint8_t*)d_out, (uint8_t*)d_status);
        else if (pick >= (size_t(1) << 18))
            hipLaunchKernelGGL(kyb::bn256_hash_g1_queue_kernel<256>, dim3((unsigned)((n + 255) / 256)), dim3(64), 0, st, n, (const uint8_t*)d_msgs,
                               msg_len, (uint8_t*)d_out, (uint8_t*)d_status);
        else
            hipLaunchKernelGGL(kyb::bn256_hash_g1_queue_kernel<128>, dim3((unsigned)((n + 127) / 128)), dim3(64), 0, st, n, (const uint8_t*)d_msgs,
                               msg_len, (uint8_t*)d_out, (uint8_t*)d_status);
        KYB_HIP_CHECK(hipGetLastError());
        return KYB_OK;
    }
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
    KYB_TRY(kyb_bn256_hash_g1_dev(n, m.p, msg_len, o.p, st.p, sc_.stream()));
    KYB_TRY(o.download(out, n * 64));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
}

// ---- HashG1 (pairing/bn256/hash.go:10-110): HKDF-SHA-256 + Shallue-van de Woestijne, no divergent retry loop
namespace kyb {
__global__ __launch_bounds__(64, KYB_TU_WAVES) void bn256_hash_g1_svdw_kernel(size_t n, const uint8_t* __restrict__ msgs, size_t msg_len, DstArg dst,
                                                                uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = bn::hash_g1_svdw_wire(out + 64 * idx, msgs + msg_len * idx, msg_len, dst);
    if (status) status[idx] = (uint8_t)st;
}
}  // namespace kyb
extern "C" {
int kyb_bn256_hash_g1_svdw_dev(size_t n, const void* d_msgs, size_t msg_len, const uint8_t* dst, size_t dst_len, void* d_out,
                               void* d_status, void* stream) {
    if ((n && ((!d_msgs && msg_len) || !d_out)) || dst_len > 255 || (dst_len && !dst)) {
        kyb::set_error("kyb_bn256_hash_g1_svdw_dev: bad argument (the domain separation tag is at most 255 bytes)");
        return KYB_E_ARG;
    }
    kyb::DstArg d;
    memset(&d, 0, sizeof d);
    if (dst_len) memcpy(d.b, dst, dst_len);
    d.len = (uint32_t)dst_len;
    if (!n) return KYB_OK;
    hipLaunchKernelGGL(kyb::bn256_hash_g1_svdw_kernel, dim3(kyb::grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n,
                       (const uint8_t*)d_msgs, msg_len, d, (uint8_t*)d_out, (uint8_t*)d_status);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
int kyb_bn256_hash_g1_svdw(size_t n, const uint8_t* msgs, size_t msg_len, const uint8_t* dst, size_t dst_len, uint8_t* out,
                           uint8_t* status) {
    if ((n && ((!msgs && msg_len) || !out)) || dst_len > 255 || (dst_len && !dst)) {
        kyb::set_error("kyb_bn256_hash_g1_svdw: bad argument (the domain separation tag is at most 255 bytes)");
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
    KYB_TRY(kyb_bn256_hash_g1_svdw_dev(n, m.p, msg_len, dst, dst_len, o.p, st.p, sc_.stream()));
    KYB_TRY(o.download(out, n * 64));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
}
