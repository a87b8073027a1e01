// BLS12-381 batch kernels for gfx950 + their C-ABI entry points: one group operation / pairing
// per lane, integer VALU only, uniform control flow inside a wave except for rejected inputs.
//
// Replaces, behind pairing/bls12381/kilic (the adapter whose Point()/Pair() the suite exposes):
//   G1Elt.Mul / G2Elt.Mul            kilic/g1.go:110-116, g2.go  -> bls_g1_mul_kernel / bls_g2_mul_kernel
//   Suite.Pair                       kilic/suite.go:70-75        -> bls_pair_kernel
//   Suite.ValidatePairing            kilic/suite.go:57-68        -> bls_pair_check_kernel
//   Unmarshal/MarshalBinary          kilic/g1.go:119-131         -> fused into every kernel
#include "bls12381.cuh"
#include "context.h"

namespace kyb {

__global__ __launch_bounds__(64) void bls_g1_mul_kernel(size_t n, const uint8_t* __restrict__ scalars,
                                                        const uint8_t* __restrict__ pts, size_t pt_stride,
                                                        uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = bls::g1_mul_wire(out + 48 * idx, scalars + 32 * idx, pts + pt_stride * idx);
    if (status) status[idx] = (uint8_t)st;
}
__global__ __launch_bounds__(64) void bls_g2_mul_kernel(size_t n, const uint8_t* __restrict__ scalars,
                                                        const uint8_t* __restrict__ pts, size_t pt_stride,
                                                        uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = bls::g2_mul_wire(out + 96 * idx, scalars + 32 * idx, pts + pt_stride * idx);
    if (status) status[idx] = (uint8_t)st;
}
__global__ __launch_bounds__(64) void bls_pair_kernel(size_t n, const uint8_t* __restrict__ g1,
                                                      const uint8_t* __restrict__ g2, uint8_t* __restrict__ gt,
                                                      uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st = bls::pair_wire(gt + 576 * idx, g1 + 48 * idx, g2 + 96 * idx);
    if (status) status[idx] = (uint8_t)st;
}
__global__ __launch_bounds__(64) void bls_pair_check_kernel(size_t n, const uint8_t* __restrict__ p1,
                                                            const uint8_t* __restrict__ p2,
                                                            const uint8_t* __restrict__ i1,
                                                            const uint8_t* __restrict__ i2, uint8_t* __restrict__ ok,
                                                            uint8_t* __restrict__ status) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    uint8_t r = 0;
    const int st = bls::pair_check_wire(&r, p1 + 48 * idx, p2 + 96 * idx, i1 + 48 * idx, i2 + 96 * idx);
    ok[idx] = r;
    if (status) status[idx] = (uint8_t)st;
}

static inline unsigned grid_for(size_t n, int block) { return (unsigned)((n + block - 1) / block); }

}  // namespace kyb

using namespace kyb;

// Host entry points stage through temporary device buffers (StageBuf frees on scope exit).
namespace {
struct StageBuf {
    void* p = nullptr;
    ~StageBuf() {
        if (p) hipFree(p);
    }
    int alloc(size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
            set_error("hipMalloc failed");
            return KYB_E_ALLOC;
        }
        return KYB_OK;
    }
    int upload(const void* src, size_t bytes) {
        int rc = alloc(bytes);
        if (rc) return rc;
        KYB_HIP_CHECK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return KYB_OK;
    }
    int download(void* dst, size_t bytes) {
        KYB_HIP_CHECK(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
        return KYB_OK;
    }
};
#define KYB_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_) return rc_;     \
    } while (0)
}  // namespace

extern "C" {

int kyb_bls12381_g1_mul_dev(size_t n, const void* d_scalars, const void* d_points, size_t point_stride,
                            void* d_out, void* d_status, void* stream) {
    if (n && (!d_scalars || !d_points || !d_out) || (point_stride != 0 && point_stride != 48)) {
        set_error("kyb_bls12381_g1_mul_dev: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    hipLaunchKernelGGL(bls_g1_mul_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n,
                       (const uint8_t*)d_scalars, (const uint8_t*)d_points, point_stride, (uint8_t*)d_out,
                       (uint8_t*)d_status);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
int kyb_bls12381_g2_mul_dev(size_t n, const void* d_scalars, const void* d_points, size_t point_stride,
                            void* d_out, void* d_status, void* stream) {
    if (n && (!d_scalars || !d_points || !d_out) || (point_stride != 0 && point_stride != 96)) {
        set_error("kyb_bls12381_g2_mul_dev: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    hipLaunchKernelGGL(bls_g2_mul_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n,
                       (const uint8_t*)d_scalars, (const uint8_t*)d_points, point_stride, (uint8_t*)d_out,
                       (uint8_t*)d_status);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
int kyb_bls12381_pair_dev(size_t n, const void* d_g1, const void* d_g2, void* d_gt, void* d_status, void* stream) {
    if (n && (!d_g1 || !d_g2 || !d_gt)) {
        set_error("kyb_bls12381_pair_dev: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    hipLaunchKernelGGL(bls_pair_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n,
                       (const uint8_t*)d_g1, (const uint8_t*)d_g2, (uint8_t*)d_gt, (uint8_t*)d_status);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}
int kyb_bls12381_pair_check_dev(size_t n, const void* d_p1, const void* d_p2, const void* d_inv1, const void* d_inv2,
                                void* d_ok, void* d_status, void* stream) {
    if (n && (!d_p1 || !d_p2 || !d_inv1 || !d_inv2 || !d_ok)) {
        set_error("kyb_bls12381_pair_check_dev: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    hipLaunchKernelGGL(bls_pair_check_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n,
                       (const uint8_t*)d_p1, (const uint8_t*)d_p2, (const uint8_t*)d_inv1, (const uint8_t*)d_inv2,
                       (uint8_t*)d_ok, (uint8_t*)d_status);
    KYB_HIP_CHECK(hipGetLastError());
    return KYB_OK;
}

static int bls_mul_host(bool g2, size_t n, const uint8_t* scalars, const uint8_t* points, size_t stride,
                        uint8_t* out, uint8_t* status) {
    const size_t psz = g2 ? 96 : 48;
    if (n && (!scalars || !points || !out)) {
        set_error("kyb_bls12381_g*_mul: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    DeviceCtx* ctx;
    KYB_TRY(get_ctx(&ctx));
    StageBuf s, p, o, st;
    KYB_TRY(s.upload(scalars, n * 32));
    KYB_TRY(p.upload(points, (stride ? n : 1) * psz));
    KYB_TRY(o.alloc(n * psz));
    KYB_TRY(st.alloc(n));
    KYB_TRY(g2 ? kyb_bls12381_g2_mul_dev(n, s.p, p.p, stride, o.p, st.p, nullptr)
               : kyb_bls12381_g1_mul_dev(n, s.p, p.p, stride, o.p, st.p, nullptr));
    KYB_TRY(o.download(out, n * psz));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
int kyb_bls12381_g1_mul(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out, uint8_t* status) {
    return bls_mul_host(false, n, scalars, points, 48, out, status);
}
int kyb_bls12381_g2_mul(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out, uint8_t* status) {
    return bls_mul_host(true, n, scalars, points, 96, out, status);
}
int kyb_bls12381_g1_mul_same_base(size_t n, const uint8_t* scalars, const uint8_t point[48], uint8_t* out,
                                  uint8_t* status) {
    return bls_mul_host(false, n, scalars, point, 0, out, status);
}
int kyb_bls12381_g2_mul_same_base(size_t n, const uint8_t* scalars, const uint8_t point[96], uint8_t* out,
                                  uint8_t* status) {
    return bls_mul_host(true, n, scalars, point, 0, out, status);
}
int kyb_bls12381_pair(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, uint8_t* status) {
    if (n && (!g1 || !g2 || !gt)) {
        set_error("kyb_bls12381_pair: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    DeviceCtx* ctx;
    KYB_TRY(get_ctx(&ctx));
    StageBuf a, b, o, st;
    KYB_TRY(a.upload(g1, n * 48));
    KYB_TRY(b.upload(g2, n * 96));
    KYB_TRY(o.alloc(n * 576));
    KYB_TRY(st.alloc(n));
    KYB_TRY(kyb_bls12381_pair_dev(n, a.p, b.p, o.p, st.p, nullptr));
    KYB_TRY(o.download(gt, n * 576));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
int kyb_bls12381_pair_check(size_t n, const uint8_t* p1, const uint8_t* p2, const uint8_t* inv1, const uint8_t* inv2,
                            uint8_t* ok, uint8_t* status) {
    if (n && (!p1 || !p2 || !inv1 || !inv2 || !ok)) {
        set_error("kyb_bls12381_pair_check: bad argument");
        return KYB_E_ARG;
    }
    if (!n) return KYB_OK;
    DeviceCtx* ctx;
    KYB_TRY(get_ctx(&ctx));
    StageBuf a, b, c, d, o, st;
    KYB_TRY(a.upload(p1, n * 48));
    KYB_TRY(b.upload(p2, n * 96));
    KYB_TRY(c.upload(inv1, n * 48));
    KYB_TRY(d.upload(inv2, n * 96));
    KYB_TRY(o.alloc(n));
    KYB_TRY(st.alloc(n));
    KYB_TRY(kyb_bls12381_pair_check_dev(n, a.p, b.p, c.p, d.p, o.p, st.p, nullptr));
    KYB_TRY(o.download(ok, n));
    if (status) KYB_TRY(st.download(status, n));
    return KYB_OK;
}
}
