// BLS12-381 hash-to-curve kernels + C-ABI entry points (bls12381_h2c.cuh): one message per lane.
// Replaces G1Elt.Hash / G2Elt.Hash (pairing/bls12381/kilic/g1.go:161-170, g2.go) -- HashablePoint.Hash, the step
// before the pairing check in sign/bls Verify (sign/bls/bls.go:87-88).
#include "bls12381_h2c.cuh"
#include "bls12381_tvm.h"
#include "pairing_abi.cuh"

#include <string.h>

namespace kyb {
__global__ __launch_bounds__(64) void bls12381_hash_g1_kernel(size_t n, const uint8_t* __restrict__ msgs, size_t msg_len,
                                                              bls::DstArg dst, uint8_t* __restrict__ out,
                                                              uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = bls::hash_g1_wire(out + 48 * idx, msgs + msg_len * idx, msg_len, dst);
    if (status) status[idx] = (uint8_t)st;
}
__global__ __launch_bounds__(64) void bls12381_hash_g2_kernel(size_t n, const uint8_t* __restrict__ msgs, size_t msg_len,
                                                              bls::DstArg dst, uint8_t* __restrict__ out,
                                                              uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = bls::hash_g2_wire(out + 96 * idx, msgs + msg_len * idx, msg_len, dst);
    if (status) status[idx] = (uint8_t)st;
}
// Operand kernels of the fused verification: one (key, message, signature) per lane -> the CHECK program's twelve
// field elements + flag byte in the tower machine's workspace (bls12381_tvm.h).
__device__ __forceinline__ void put_fp(uint32_t* in, size_t n, int idx, size_t i, const bls::fp& x) {
    uint32_t* d = in + ((size_t)idx * n + i) * blsvm::FP_WORDS;
#pragma unroll
    for (int k = 0; k < blsvm::FP_WORDS; k += 4) *reinterpret_cast<uint4*>(d + k) = make_uint4(x.v[k], x.v[k + 1], x.v[k + 2], x.v[k + 3]);
}
__device__ __forceinline__ void put_operands(uint32_t* in, size_t n, size_t i, const bls::g1_aff& a1, const bls::g2_aff& a2,
                                             const bls::g1_aff& b1, const bls::g2_aff& b2, uint8_t* fl, int st) {
    const bls::g1_aff* p[2] = {&a1, &b1};
    const bls::g2_aff* q[2] = {&a2, &b2};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        put_fp(in, n, 6 * k, i, p[k]->x);
        put_fp(in, n, 6 * k + 1, i, p[k]->y);
        put_fp(in, n, 6 * k + 2, i, q[k]->x.c0);
        put_fp(in, n, 6 * k + 3, i, q[k]->x.c1);
        put_fp(in, n, 6 * k + 4, i, q[k]->y.c0);
        put_fp(in, n, 6 * k + 5, i, q[k]->y.c1);
    }
    fl[i] = (uint8_t)(((a1.inf | a2.inf) ? blsvm::FL_DEAD_A : 0) | ((b1.inf | b2.inf) ? blsvm::FL_DEAD_B : 0) |
                      (st ? blsvm::FL_REJECTED : 0));
}
__global__ __launch_bounds__(64) void bls12381_verify_g1_prep_kernel(size_t n, const uint8_t* __restrict__ pks,
                                                                     const uint8_t* __restrict__ msgs, size_t msg_len,
                                                                     bls::DstArg dst, const uint8_t* __restrict__ sigs,
                                                                     uint32_t* __restrict__ in, uint8_t* __restrict__ fl,
                                                                     uint8_t* __restrict__ status, uint32_t flags) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    bls::g1_aff a1, b1;
    bls::g2_aff a2, b2;
    const int st = bls::verify_g1_operands(a1, a2, b1, b2, pks + bls::g2_wire_size(flags) * idx, msgs + msg_len * idx, msg_len,
                                           dst, sigs + bls::g1_wire_size(flags) * idx, flags);
    put_operands(in, n, idx, a1, a2, b1, b2, fl, st);
    if (status) status[idx] = (uint8_t)st;
}
__global__ __launch_bounds__(64) void bls12381_verify_g2_prep_kernel(size_t n, const uint8_t* __restrict__ pks,
                                                                     const uint8_t* __restrict__ msgs, size_t msg_len,
                                                                     bls::DstArg dst, const uint8_t* __restrict__ sigs,
                                                                     uint32_t* __restrict__ in, uint8_t* __restrict__ fl,
                                                                     uint8_t* __restrict__ status, uint32_t flags) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    bls::g1_aff a1, b1;
    bls::g2_aff a2, b2;
    const int st = bls::verify_g2_operands(a1, a2, b1, b2, pks + bls::g1_wire_size(flags) * idx, msgs + msg_len * idx, msg_len,
                                           dst, sigs + bls::g2_wire_size(flags) * idx, flags);
    put_operands(in, n, idx, a1, a2, b1, b2, fl, st);
    if (status) status[idx] = (uint8_t)st;
}
}  // namespace kyb

using namespace kyb;

static int make_dst(bls::DstArg& d, const uint8_t* dst, size_t dst_len) {
    if (dst_len > 255 || (dst_len && !dst)) {
        set_error("hash-to-curve: the domain separation tag must be at most 255 bytes");
        return KYB_E_ARG;
    }
    memset(&d, 0, sizeof d);
    if (dst_len) memcpy(d.b, dst, dst_len);
    d.len = (uint32_t)dst_len;
    return KYB_OK;
}

extern "C" {
int kyb_bls12381_hash_g1_dev(size_t n, const void* d_msgs, size_t msg_len, const uint8_t* dst, size_t dst_len, void* d_out,
                             void* d_status, void* stream) {
    if (n && ((!d_msgs && msg_len) || !d_out)) {
        set_error("kyb_bls12381_hash_g1_dev: bad argument");
        return KYB_E_ARG;
    }
    bls::DstArg d;
    KYB_TRY(make_dst(d, dst, dst_len));
    if (!n) return KYB_OK;
    hipLaunchKernelGGL(bls12381_hash_g1_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n,
                       (const uint8_t*)d_msgs, msg_len, d, (uint8_t*)d_out, (uint8_t*)d_status);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
int kyb_bls12381_hash_g2_dev(size_t n, const void* d_msgs, size_t msg_len, const uint8_t* dst, size_t dst_len, void* d_out,
                             void* d_status, void* stream) {
    if (n && ((!d_msgs && msg_len) || !d_out)) {
        set_error("kyb_bls12381_hash_g2_dev: bad argument");
        return KYB_E_ARG;
    }
    bls::DstArg d;
    KYB_TRY(make_dst(d, dst, dst_len));
    if (!n) return KYB_OK;
    hipLaunchKernelGGL(bls12381_hash_g2_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n,
                       (const uint8_t*)d_msgs, msg_len, d, (uint8_t*)d_out, (uint8_t*)d_status);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
static int hash_host(bool g2, size_t n, const uint8_t* msgs, size_t msg_len, const uint8_t* dst, size_t dst_len,
                     uint8_t* out, uint8_t* status) {
    if (n && ((!msgs && msg_len) || !out)) {
        set_error("kyb_bls12381_hash_g*: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    DeviceCtx* ctx;
    KYB_TRY(get_ctx(&ctx));
    const size_t osz = g2 ? 96 : 48;
    kyb::StageScope sc_(ctx);
    StageBuf m, o, st;
    KYB_TRY(m.upload(msgs, n * msg_len));
    KYB_TRY(o.alloc(n * osz));
    KYB_TRY(st.alloc(n));
    KYB_TRY(g2 ? kyb_bls12381_hash_g2_dev(n, m.p, msg_len, dst, dst_len, o.p, st.p, nullptr)
               : kyb_bls12381_hash_g1_dev(n, m.p, msg_len, dst, dst_len, o.p, st.p, nullptr));
    KYB_TRY(o.download(out, n * osz));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
int kyb_bls12381_hash_g1(size_t n, const uint8_t* msgs, size_t msg_len, const uint8_t* dst, size_t dst_len, uint8_t* out,
                         uint8_t* status) {
    return hash_host(false, n, msgs, msg_len, dst, dst_len, out, status);
}
int kyb_bls12381_hash_g2(size_t n, const uint8_t* msgs, size_t msg_len, const uint8_t* dst, size_t dst_len, uint8_t* out,
                         uint8_t* status) {
    return hash_host(true, n, msgs, msg_len, dst, dst_len, out, status);
}
int kyb_bls12381_verify_g1_dev(size_t n, const void* d_pks, const void* d_msgs, size_t msg_len, const uint8_t* dst,
                               size_t dst_len, const void* d_sigs, void* d_ok, void* d_status, uint32_t flags, void* stream) {
    if (n && (!d_pks || (!d_msgs && msg_len) || !d_sigs || !d_ok)) {
        set_error("kyb_bls12381_verify_g1_dev: bad argument");
        return KYB_E_ARG;
    }
    KYB_TRY(check_flags(flags, 2, false, "kyb_bls12381_verify_g1_dev"));
    bls::DstArg d;
    KYB_TRY(make_dst(d, dst, dst_len));
    if (!n) return KYB_OK;
    DeviceCtx* ctx;
    KYB_TRY(get_ctx(&ctx));
    std::lock_guard<std::recursive_mutex> enq_lock(ctx->enq_mu);
    blsvm::Work w;
    KYB_TRY(blsvm::workspace(ctx, (hipStream_t)stream, n, blsvm::CHECK_INPUTS, &w));
    hipLaunchKernelGGL(bls12381_verify_g1_prep_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n,
                       (const uint8_t*)d_pks, (const uint8_t*)d_msgs, msg_len, d, (const uint8_t*)d_sigs, w.in, w.flags,
                       (uint8_t*)d_status, flags);
    return blsvm::launch_check(w, n, (uint8_t*)d_ok, (hipStream_t)stream);
}
int kyb_bls12381_verify_g1(size_t n, const uint8_t* pks, const uint8_t* msgs, size_t msg_len, const uint8_t* dst,
                           size_t dst_len, const uint8_t* sigs, uint8_t* ok, uint8_t* status, uint32_t flags) {
    if (n && (!pks || (!msgs && msg_len) || !sigs || !ok)) {
        set_error("kyb_bls12381_verify_g1: bad argument");
        return KYB_E_ARG;
    }
    KYB_TRY(check_flags(flags, 2, false, "kyb_bls12381_verify_g1"));
    if (!n) return KYB_OK;
    if (md_active(n))
        return md_run(n, [&](int, size_t lo, size_t hi) {
            return kyb_bls12381_verify_g1(hi - lo, pks + bls::g2_wire_size(flags) * lo, msgs + msg_len * lo, msg_len, dst, dst_len,
                                         sigs + bls::g1_wire_size(flags) * lo, ok + lo, status ? status + lo : nullptr, flags);
        });
    DeviceCtx* ctx;
    KYB_TRY(get_ctx(&ctx));
    kyb::StageScope sc_(ctx);
    StageBuf p, m, s, o, st;
    KYB_TRY(p.upload(pks, n * bls::g2_wire_size(flags)));
    KYB_TRY(m.upload(msgs, n * msg_len));
    KYB_TRY(s.upload(sigs, n * bls::g1_wire_size(flags)));
    KYB_TRY(o.alloc(n));
    KYB_TRY(st.alloc(n));
    KYB_TRY(kyb_bls12381_verify_g1_dev(n, p.p, m.p, msg_len, dst, dst_len, s.p, o.p, st.p, flags, nullptr));
    KYB_TRY(o.download(ok, n));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
int kyb_bls12381_verify_g2_dev(size_t n, const void* d_pks, const void* d_msgs, size_t msg_len, const uint8_t* dst,
                               size_t dst_len, const void* d_sigs, void* d_ok, void* d_status, uint32_t flags, void* stream) {
    if (n && (!d_pks || (!d_msgs && msg_len) || !d_sigs || !d_ok)) {
        set_error("kyb_bls12381_verify_g2_dev: bad argument");
        return KYB_E_ARG;
    }
    KYB_TRY(check_flags(flags, 2, false, "kyb_bls12381_verify_g2_dev"));
    bls::DstArg d;
    KYB_TRY(make_dst(d, dst, dst_len));
    if (!n) return KYB_OK;
    DeviceCtx* ctx;
    KYB_TRY(get_ctx(&ctx));
    std::lock_guard<std::recursive_mutex> enq_lock(ctx->enq_mu);
    blsvm::Work w;
    KYB_TRY(blsvm::workspace(ctx, (hipStream_t)stream, n, blsvm::CHECK_INPUTS, &w));
    hipLaunchKernelGGL(bls12381_verify_g2_prep_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n,
                       (const uint8_t*)d_pks, (const uint8_t*)d_msgs, msg_len, d, (const uint8_t*)d_sigs, w.in, w.flags,
                       (uint8_t*)d_status, flags);
    return blsvm::launch_check(w, n, (uint8_t*)d_ok, (hipStream_t)stream);
}
int kyb_bls12381_verify_g2(size_t n, const uint8_t* pks, const uint8_t* msgs, size_t msg_len, const uint8_t* dst,
                           size_t dst_len, const uint8_t* sigs, uint8_t* ok, uint8_t* status, uint32_t flags) {
    if (n && (!pks || (!msgs && msg_len) || !sigs || !ok)) {
        set_error("kyb_bls12381_verify_g2: bad argument");
        return KYB_E_ARG;
    }
    KYB_TRY(check_flags(flags, 2, false, "kyb_bls12381_verify_g2"));
    if (!n) return KYB_OK;
    if (md_active(n))
        return md_run(n, [&](int, size_t lo, size_t hi) {
            return kyb_bls12381_verify_g2(hi - lo, pks + bls::g1_wire_size(flags) * lo, msgs + msg_len * lo, msg_len, dst, dst_len,
                                         sigs + bls::g2_wire_size(flags) * lo, ok + lo, status ? status + lo : nullptr, flags);
        });
    DeviceCtx* ctx;
    KYB_TRY(get_ctx(&ctx));
    kyb::StageScope sc_(ctx);
    StageBuf p, m, s, o, st;
    KYB_TRY(p.upload(pks, n * bls::g1_wire_size(flags)));
    KYB_TRY(m.upload(msgs, n * msg_len));
    KYB_TRY(s.upload(sigs, n * bls::g2_wire_size(flags)));
    KYB_TRY(o.alloc(n));
    KYB_TRY(st.alloc(n));
    KYB_TRY(kyb_bls12381_verify_g2_dev(n, p.p, m.p, msg_len, dst, dst_len, s.p, o.p, st.p, flags, nullptr));
    KYB_TRY(o.download(ok, n));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
}
