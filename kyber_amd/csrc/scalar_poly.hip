// kyb_<suite>_scalar_poly_eval: share.PriPoly.Eval / PriPoly.Shares (share/poly.go:85-102) for all participants in one
// launch -- scalar_field.cuh.  Two kernels: the t coefficients go to Montgomery form once (one lane each), then one lane
// per index runs the reference's Horner loop over them.
#include "context.h"
#include "scalar_field.cuh"

namespace kyb {
namespace sf {



static __global__ __launch_bounds__(256) void to_mont_kernel(size_t t, const uint8_t* __restrict__ coeffs, uint32_t* __restrict__ cm, Mod m) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= t) return;
    uint32_t r[8];
    to_mont(r, coeffs + 32 * j, m);
#pragma unroll
    for (int i = 0; i < 8; i++) cm[8 * j + i] = r[i];
}
static __global__ __launch_bounds__(256) void horner_kernel(size_t n, const uint32_t* __restrict__ idx, size_t t, const uint32_t* __restrict__ cm,
                                                            uint8_t* __restrict__ out, Mod m) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    horner(out + 32 * i, idx[i], t, cm, m);
}

static int run(const Mod& m, size_t n, const void* d_idx, size_t t, const void* d_coeffs, void* d_out, hipStream_t st) {
    if ((n && (!d_idx || !d_out)) || (t && !d_coeffs)) {
        set_error("scalar_poly_eval: bad argument");
        return KYB_E_ARG;
    }
    if (t >= (size_t(1) << 28)) {
        set_error("scalar_poly_eval: threshold too large");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    DeviceCtx* ctx;
    if (int rc = get_ctx(&ctx)) return rc;
    std::lock_guard<std::recursive_mutex> enq_lock(ctx->enq_mu);
    void* ws;
    if (int rc = ctx_workspace(ctx, WS_SCALAR, st, 32 * (t ? t : 1), &ws)) return rc;
    if (t) hipLaunchKernelGGL(to_mont_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, st, t, (const uint8_t*)d_coeffs, (uint32_t*)ws, m);
    hipLaunchKernelGGL(horner_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, (const uint32_t*)d_idx, t, (const uint32_t*)ws,
                       (uint8_t*)d_out, m);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
static int run_host(const Mod& m, size_t n, const uint32_t* idx, size_t t, const uint8_t* coeffs, uint8_t* out) {
    if ((n && (!idx || !out)) || (t && !coeffs)) {
        set_error("scalar_poly_eval: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    if (n >= (size_t(1) << 28) || t >= (size_t(1) << 28)) {  // before any staging allocation: n * 32 and t * 32 below are then exact
        set_error("scalar_poly_eval: n or t too large");
        return KYB_E_ARG;
    }
    DeviceCtx* ctx;
    if (int rc = get_ctx(&ctx)) return rc;
    StageScope sc_(ctx);  // (the enqueue mutex is run()'s: held for the enqueue, not for the copies)
    StageBuf d_i, d_c, d_o;
    int rc = d_i.upload(idx, n * 4);
    if (rc == KYB_OK) rc = d_c.upload(coeffs, t * 32);
    if (rc == KYB_OK) rc = d_o.alloc(n * 32);
    if (rc == KYB_OK) rc = run(m, n, d_i.p, t, d_c.p, d_o.p, sc_.stream());
    if (rc == KYB_OK) rc = d_o.download(out, n * 32);
    return rc;
}
static const Mod& mod_ed25519() {
    static const Mod m = make_mod(Q_ED25519, false);
    return m;
}
static const Mod& mod_bls12381() {
    static const Mod m = make_mod(Q_BLS12381, true);
    return m;
}
static const Mod& mod_bn256() {
    static const Mod m = make_mod(Q_BN256, true);
    return m;
}
static const Mod& mod_bn254() {
    static const Mod m = make_mod(Q_BN254, true);
    return m;
}
}  // namespace sf
}  // namespace kyb

#define KYB_DEFINE_SCALAR_POLY(SUITE)                                                                                          \
    int kyb_##SUITE##_scalar_poly_eval(size_t n, const uint32_t* idx, size_t t, const uint8_t* coeffs, uint8_t* out) {          \
        return kyb::sf::run_host(kyb::sf::mod_##SUITE(), n, idx, t, coeffs, out);                                              \
    }                                                                                                                          \
    int kyb_##SUITE##_scalar_poly_eval_dev(size_t n, const void* d_idx, size_t t, const void* d_coeffs, void* d_out,            \
                                           void* stream) {                                                                     \
        return kyb::sf::run(kyb::sf::mod_##SUITE(), n, d_idx, t, d_coeffs, d_out, (hipStream_t)stream);                        \
    }
extern "C" {
KYB_DEFINE_SCALAR_POLY(ed25519)
KYB_DEFINE_SCALAR_POLY(bls12381)
KYB_DEFINE_SCALAR_POLY(bn256)
KYB_DEFINE_SCALAR_POLY(bn254)
}
