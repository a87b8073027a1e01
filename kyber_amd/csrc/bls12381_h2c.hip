// BLS12-381 hash-to-curve kernels + C-ABI entry points (bls12381_h2c.cuh): one message per lane.
// Replaces G1Elt.Hash / G2Elt.Hash (pairing/bls12381/kilic/g1.go:161-170, g2.go) -- HashablePoint.Hash, the step
// before the pairing check in sign/bls Verify (sign/bls/bls.go:87-88).
#include "bls12381_h2c.cuh"
#include "bls12381_tvm.h"
#include "pairing_abi.cuh"

#include <stdlib.h>
#include <string.h>

namespace kyb {
__global__ __launch_bounds__(64, KYB_TU_WAVES) void bls12381_hash_g1_kernel(size_t n, const uint8_t* __restrict__ msgs, size_t msg_len,
                                                              bls::DstArg dst, uint8_t* __restrict__ out,
                                                              uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = bls::hash_g1_wire(out + 48 * idx, msgs + msg_len * idx, msg_len, dst);
    if (status) status[idx] = (uint8_t)st;
}
__global__ __launch_bounds__(64, KYB_TU_WAVES) void bls12381_hash_g2_kernel(size_t n, const uint8_t* __restrict__ msgs, size_t msg_len,
                                                              bls::DstArg dst, uint8_t* __restrict__ out,
                                                              uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = bls::hash_g2_wire(out + 96 * idx, msgs + msg_len * idx, msg_len, dst);
    if (status) status[idx] = (uint8_t)st;
}
}  // namespace kyb

namespace kyb {
namespace bls {
// the same kernels on a two-wave register budget (bls12381_unm2.hip), for batches with two waves per SIMD in flight;
// KYB_UNM_W2=0 keeps every batch on the kernels above (A/B)
void launch_hash_w2(bool g2, size_t n, const uint8_t* d_msgs, size_t msg_len, const DstArg& dst, uint8_t* d_out, uint8_t* d_status, hipStream_t st);
}  // namespace bls
static bool hash_w2(size_t n) {
    static const bool on = [] {
        const char* e = getenv("KYB_UNM_W2");
        return !(e && e[0] == '0');
    }();
    DeviceCtx* ctx;
    return on && get_ctx(&ctx) == KYB_OK && n >= (size_t)ctx->num_cu * 4 * 64 * 2;
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
    if (hash_w2(n)) bls::launch_hash_w2(false, n, (const uint8_t*)d_msgs, msg_len, d, (uint8_t*)d_out, (uint8_t*)d_status, (hipStream_t)stream);
    else
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
    if (hash_w2(n)) bls::launch_hash_w2(true, n, (const uint8_t*)d_msgs, msg_len, d, (uint8_t*)d_out, (uint8_t*)d_status, (hipStream_t)stream);
    else
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
    KYB_TRY(g2 ? kyb_bls12381_hash_g2_dev(n, m.p, msg_len, dst, dst_len, o.p, st.p, sc_.stream())
               : kyb_bls12381_hash_g1_dev(n, m.p, msg_len, dst, dst_len, o.p, st.p, sc_.stream()));
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
    // signatures on G1, keys on G2:  e(H(m), X) e(-sig, G2.Base()) == 1   (status precedence: key, then signature)
    const blsvm::Operand ops[4] = {{(const uint8_t*)d_msgs, blsvm::OPND_G1_HASH, (uint32_t)msg_len, 0, 0, 0},
                                   {(const uint8_t*)d_pks, blsvm::OPND_G2, (uint32_t)bls::g2_wire_size(flags), 2, 0, 0},
                                   {(const uint8_t*)d_sigs, blsvm::OPND_G1, (uint32_t)bls::g1_wire_size(flags), 6, 1, 1},
                                   {nullptr, blsvm::OPND_G2_GEN, 0, 8, 0, 0}};
#ifdef KYB_BLS_VERIFY_GENERAL  // A/B: the general product check with the generator as an ordinary operand
    KYB_TRY(blsvm::workspace(ctx, (hipStream_t)stream, n, blsvm::CHECK_INPUTS, &w));
    KYB_TRY(blsvm::launch_prep(w, n, ops, 4, flags, dst, dst_len, (hipStream_t)stream));
    return blsvm::launch_check(w, n, (uint8_t*)d_ok, (uint8_t*)d_status, (hipStream_t)stream);
#else
    // the generator's Miller lines are constants of the VERIFY program: three operands, two thirds of the line work
    KYB_TRY(blsvm::workspace(ctx, (hipStream_t)stream, n, blsvm::VERIFY_INPUTS, &w));
    KYB_TRY(blsvm::launch_prep(w, n, ops, 3, flags, dst, dst_len, (hipStream_t)stream));
    return blsvm::launch_verify(w, n, (uint8_t*)d_ok, (uint8_t*)d_status, (hipStream_t)stream);
#endif
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
    KYB_TRY(kyb_bls12381_verify_g1_dev(n, p.p, m.p, msg_len, dst, dst_len, s.p, o.p, st.p, flags, sc_.stream()));
    KYB_TRY(o.download(ok, n));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
// sign/bls Verify (bls.go:82-96) for n (message, signature) pairs under ONE public key -- a drand chain, the partial
// signatures of one tbls participant (sign/tbls/tbls.go:100-107): both Miller loops from line tables (program VERIFYK)
int kyb_bls12381_verify_g1_same_key_dev(size_t n, const void* d_pk, const void* d_msgs, size_t msg_len, const uint8_t* dst,
                                        size_t dst_len, const void* d_sigs, void* d_ok, void* d_status, uint32_t flags, void* stream) {
    if (!d_pk || (n && ((!d_msgs && msg_len) || !d_sigs || !d_ok))) {
        set_error("kyb_bls12381_verify_g1_same_key_dev: bad argument");
        return KYB_E_ARG;
    }
    KYB_TRY(check_flags(flags, 2, false, "kyb_bls12381_verify_g1_same_key_dev"));
    bls::DstArg d;
    KYB_TRY(make_dst(d, dst, dst_len));
    if (!n) return KYB_OK;
    DeviceCtx* ctx;
    KYB_TRY(get_ctx(&ctx));
    std::lock_guard<std::recursive_mutex> enq_lock(ctx->enq_mu);
    const int32_t* table;
    const uint8_t* kst;
    KYB_TRY(blsvm::prepare_key(ctx, (hipStream_t)stream, (const uint8_t*)d_pk, flags, &table, &kst));
    blsvm::Work w;
    KYB_TRY(blsvm::workspace(ctx, (hipStream_t)stream, n, blsvm::VERIFYK_INPUTS, &w));
    // e(H(m), X) e(-sig, G2.Base()) == 1   (status precedence: key, then signature)
    const blsvm::Operand ops[3] = {{(const uint8_t*)d_msgs, blsvm::OPND_G1_HASH, (uint32_t)msg_len, 0, 0, 0},
                                   {kst, blsvm::OPND_STATUS, 0, 0, 0, 0},
                                   {(const uint8_t*)d_sigs, blsvm::OPND_G1, (uint32_t)bls::g1_wire_size(flags), 2, 1, 1}};
    KYB_TRY(blsvm::launch_prep(w, n, ops, 3, flags, dst, dst_len, (hipStream_t)stream));
    return blsvm::launch_verify_same_key(w, n, table, (uint8_t*)d_ok, (uint8_t*)d_status, (hipStream_t)stream);
}
int kyb_bls12381_verify_g1_same_key(size_t n, const uint8_t* pk, const uint8_t* msgs, size_t msg_len, const uint8_t* dst,
                                    size_t dst_len, const uint8_t* sigs, uint8_t* ok, uint8_t* status, uint32_t flags) {
    if (!pk || (n && ((!msgs && msg_len) || !sigs || !ok))) {
        set_error("kyb_bls12381_verify_g1_same_key: bad argument");
        return KYB_E_ARG;
    }
    KYB_TRY(check_flags(flags, 2, false, "kyb_bls12381_verify_g1_same_key"));
    if (!n) return KYB_OK;
    if (md_active(n))
        return md_run(n, [&](int, size_t lo, size_t hi) {
            return kyb_bls12381_verify_g1_same_key(hi - lo, pk, msgs + msg_len * lo, msg_len, dst, dst_len,
                                                   sigs + bls::g1_wire_size(flags) * lo, ok + lo, status ? status + lo : nullptr, flags);
        });
    DeviceCtx* ctx;
    KYB_TRY(get_ctx(&ctx));
    kyb::StageScope sc_(ctx);
    StageBuf p, m, s, o, st;
    KYB_TRY(p.upload(pk, bls::g2_wire_size(flags)));
    KYB_TRY(m.upload(msgs, n * msg_len));
    KYB_TRY(s.upload(sigs, n * bls::g1_wire_size(flags)));
    KYB_TRY(o.alloc(n));
    KYB_TRY(st.alloc(n));
    KYB_TRY(kyb_bls12381_verify_g1_same_key_dev(n, p.p, m.p, msg_len, dst, dst_len, s.p, o.p, st.p, flags, sc_.stream()));
    KYB_TRY(o.download(ok, n));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
// sign/bls Verify for n (public key, signature) pairs over ONE message -- tbls.Recover (sign/tbls/tbls.go:118-131: every
// partial signature of a round signs the same msg, each under its own public share public.Eval(idx).V): H(msg) is hashed
// once per call, next to the lanes that unmarshal keys and signatures, then dealt to every pairing; program VERIFY as in
// kyb_bls12381_verify_g1.  d_msg: the message on the device (msg_len bytes).
int kyb_bls12381_verify_g1_same_msg_dev(size_t n, const void* d_pks, const void* d_msg, size_t msg_len, const uint8_t* dst,
                                        size_t dst_len, const void* d_sigs, void* d_ok, void* d_status, uint32_t flags, void* stream) {
    if (n && (!d_pks || (!d_msg && msg_len) || !d_sigs || !d_ok)) {
        set_error("kyb_bls12381_verify_g1_same_msg_dev: bad argument");
        return KYB_E_ARG;
    }
    KYB_TRY(check_flags(flags, 2, false, "kyb_bls12381_verify_g1_same_msg_dev"));
    bls::DstArg d;
    KYB_TRY(make_dst(d, dst, dst_len));
    if (!n) return KYB_OK;
    DeviceCtx* ctx;
    KYB_TRY(get_ctx(&ctx));
    std::lock_guard<std::recursive_mutex> enq_lock(ctx->enq_mu);
    blsvm::Work w;
    // e(H(msg), X_i) e(-sig_i, G2.Base()) == 1   (status precedence: key, then signature)
    const blsvm::Operand ops[3] = {{(const uint8_t*)d_msg, blsvm::OPND_G1_SHARED_HASH, (uint32_t)msg_len, 0, 0, 0},
                                   {(const uint8_t*)d_pks, blsvm::OPND_G2, (uint32_t)bls::g2_wire_size(flags), 2, 0, 0},
                                   {(const uint8_t*)d_sigs, blsvm::OPND_G1, (uint32_t)bls::g1_wire_size(flags), 6, 1, 1}};
    KYB_TRY(blsvm::workspace(ctx, (hipStream_t)stream, n, blsvm::VERIFY_INPUTS, &w));
    KYB_TRY(blsvm::launch_prep(w, n, ops, 3, flags, dst, dst_len, (hipStream_t)stream));
    return blsvm::launch_verify(w, n, (uint8_t*)d_ok, (uint8_t*)d_status, (hipStream_t)stream);
}
int kyb_bls12381_verify_g1_same_msg(size_t n, const uint8_t* pks, const uint8_t* msg, size_t msg_len, const uint8_t* dst,
                                    size_t dst_len, const uint8_t* sigs, uint8_t* ok, uint8_t* status, uint32_t flags) {
    if (n && (!pks || (!msg && msg_len) || !sigs || !ok)) {
        set_error("kyb_bls12381_verify_g1_same_msg: bad argument");
        return KYB_E_ARG;
    }
    KYB_TRY(check_flags(flags, 2, false, "kyb_bls12381_verify_g1_same_msg"));
    if (!n) return KYB_OK;
    if (md_active(n))
        return md_run(n, [&](int, size_t lo, size_t hi) {
            return kyb_bls12381_verify_g1_same_msg(hi - lo, pks + bls::g2_wire_size(flags) * lo, msg, msg_len, dst, dst_len,
                                                   sigs + bls::g1_wire_size(flags) * lo, ok + lo, status ? status + lo : nullptr, flags);
        });
    DeviceCtx* ctx;
    KYB_TRY(get_ctx(&ctx));
    kyb::StageScope sc_(ctx);
    StageBuf p, m, s, o, st;
    KYB_TRY(p.upload(pks, n * bls::g2_wire_size(flags)));
    const uint8_t none = 0;
    KYB_TRY(m.upload(msg_len ? msg : &none, msg_len ? msg_len : 1));
    KYB_TRY(s.upload(sigs, n * bls::g1_wire_size(flags)));
    KYB_TRY(o.alloc(n));
    KYB_TRY(st.alloc(n));
    KYB_TRY(kyb_bls12381_verify_g1_same_msg_dev(n, p.p, m.p, msg_len, dst, dst_len, s.p, o.p, st.p, flags, sc_.stream()));
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
    // signatures on G2, keys on G1:  e(G1.Base(), sig) e(-X, H(m)) == 1
    const blsvm::Operand ops[4] = {{nullptr, blsvm::OPND_G1_GEN, 0, 0, 0, 0},
                                   {(const uint8_t*)d_sigs, blsvm::OPND_G2, (uint32_t)bls::g2_wire_size(flags), 2, 0, 1},
                                   {(const uint8_t*)d_pks, blsvm::OPND_G1, (uint32_t)bls::g1_wire_size(flags), 6, 1, 0},
                                   {(const uint8_t*)d_msgs, blsvm::OPND_G2_HASH, (uint32_t)msg_len, 8, 0, 0}};
    KYB_TRY(blsvm::launch_prep(w, n, ops, 4, flags, dst, dst_len, (hipStream_t)stream));
    return blsvm::launch_check(w, n, (uint8_t*)d_ok, (uint8_t*)d_status, (hipStream_t)stream);
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
    KYB_TRY(kyb_bls12381_verify_g2_dev(n, p.p, m.p, msg_len, dst, dst_len, s.p, o.p, st.p, flags, sc_.stream()));
    KYB_TRY(o.download(ok, n));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
}
