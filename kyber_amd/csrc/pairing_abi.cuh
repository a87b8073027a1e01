// Batch kernels + C-ABI entry points shared by the pairing suites (BLS12-381, bn256).
// KYB_DEFINE_MUL_ABI(PFX, NS, G1SZ, G2SZ) / KYB_DEFINE_GT_ABI(PFX, NS, GTSZ) stamp out, for the curve library in
// namespace kyb::NS (g1_mul_wire / g2_mul_wire / gt_mul_wire ...), one kernel per operation -- one group operation
// per lane, one wave per workgroup -- and the host / device-pointer entry points kyb_<PFX>_* declared in
// include/kyber_hip.h.  Pair / ValidatePairing are NOT per-lane code: each suite's *_pair.hip supplies the `_dev`
// entry points on the cooperative tower machine (tower_vm.cuh) and KYB_DEFINE_PAIR_HOST adds the host-buffer ones.
#pragma once
#include <string.h>

#include <map>
#include <mutex>
#include <string>

#include "context.h"
#include "fixed_base.cuh"

namespace kyb {
static inline unsigned grid_for(size_t n, int block) { return (unsigned)((n + block - 1) / block); }

// ---- same-base batches: when does the fixed-base table (fixed_base.cuh) take over?
constexpr int fb_suite_id(const char* pfx) { return (pfx[0] == 'b' && pfx[1] == 'l') ? 0 : (pfx[4] == '6' ? 1 : 2); }
// from this many scalars a table is worth building for an unknown base (KYB_FB_MIN overrides both; 0 disables the path)
static inline size_t fb_min_batch(bool g2) {
    static const long forced = [] {
        const char* e = getenv("KYB_FB_MIN");
        return e ? atol(e) : -1L;
    }();
    if (forced == 0) return ~size_t(0);
    if (forced > 0) return (size_t)forced;
    (void)g2;
    return size_t(1) << 17;  // measured break-even, table build included: ~1e5 (G1) / ~0.5e5-0.9e5 (G2) scalars
}
constexpr size_t FB_MIN_KNOWN = 64;  // ... and from this many when the table is (about to be) there anyway
// Host entry points stage through the per-device pool of context.h (StageScope / StageBuf).
}  // namespace kyb

#define KYB_TRY(expr)        \
    do {                     \
        int rc_ = (expr);    \
        if (rc_) return rc_; \
    } while (0)

// Register budget of the G1 / G2 multiplication kernels in waves per SIMD (a suite's translation unit may set it before
// including this file): one wave unless measured otherwise at the suite's configured batch size.
#ifndef KYB_G1_MUL_WAVES
#define KYB_G1_MUL_WAVES 1
#endif
#ifndef KYB_G2_MUL_WAVES
#define KYB_G2_MUL_WAVES 1
#endif
#ifndef KYB_FB_G1_WAVES
#define KYB_FB_G1_WAVES 2
#endif
#ifndef KYB_FB_G2_WAVES
#define KYB_FB_G2_WAVES 2
#endif
// The fixed-base traits of a suite's two groups (fixed_base.cuh) and the entry that runs a same-base batch through them.
// KYB_FB_EXTERN (set by a suite's scalar-multiplication unit before including this file): the entry is only declared and
// lives in a translation unit of its own (bls12381_fb.hip), whose kernels then share no out-of-line code -- and no
// register budget -- with this unit's.
#define KYB_DEFINE_FB_TRAITS(PFX, NS) \
namespace kyb { \
struct PFX##_FbG1 { \
    using F = NS::fp; \
    using P = NS::fb_g1_policy; \
    static constexpr int MUL_WAVES = KYB_FB_G1_WAVES; /* register budget of fb::mul_kernel in waves per SIMD */ \
    static constexpr int KIND = WS_FB + 2 * fb_suite_id(#PFX); \
    /* fixed_base.cuh chain_rows_kernel: the table's doubling chain on rowfp.cuh where the base field has its limb shape */ \
    static constexpr int ROW_CHAIN = (NS::FC::N == 13 && NS::FC::W == 30) ? 1 : 0; \
    using RowC = NS::FC; \
    static constexpr uint32_t KEY_FLAGS = KYB_F_UNCOMPRESSED | KYB_F_TRUSTED(0); \
    __host__ __device__ static int decode_on_curve(Aff<F>& a, const uint8_t* in, uint32_t flags) { return P::decode_on_curve(a, in, flags); } \
    __host__ __device__ static bool needs_member(uint32_t flags) { return P::needs_member(flags); } \
    __device__ static bool member(const Aff<F>& a, const fb::Entry<F, P::NI>* tab) { return P::member(a, tab); } \
    __host__ __device__ static void encode(uint8_t* out, const Aff<F>& a, uint32_t flags) { NS::g1_encode_f(out, a, flags); } \
    __host__ __device__ static size_t wire_size(uint32_t flags) { return NS::g1_wire_size(flags); } \
    __host__ __device__ static size_t out_size(uint32_t flags) { return NS::g1_out_size(flags); } \
    __host__ __device__ static void scalar(uint32_t (&k)[8], const uint8_t* in) { NS::scalar_from_be(k, in); } \
    static void generator(Aff<F>& a) { NS::fp_const(a.x, NS::CC::G1X); NS::fp_const(a.y, NS::CC::G1Y); a.inf = false; } \
}; \
struct PFX##_FbG2 { \
    using F = NS::fp2; \
    using P = NS::fb_g2_policy; \
    static constexpr int MUL_WAVES = KYB_FB_G2_WAVES; \
    static constexpr int KIND = WS_FB + 2 * fb_suite_id(#PFX) + 1; \
    static constexpr uint32_t KEY_FLAGS = KYB_F_UNCOMPRESSED | KYB_F_TRUSTED(0); \
    __host__ __device__ static int decode_on_curve(Aff<F>& a, const uint8_t* in, uint32_t flags) { return P::decode_on_curve(a, in, flags); } \
    __host__ __device__ static bool needs_member(uint32_t flags) { return P::needs_member(flags); } \
    __device__ static bool member(const Aff<F>& a, const fb::Entry<F, P::NI>* tab) { return P::member(a, tab); } \
    __host__ __device__ static void encode(uint8_t* out, const Aff<F>& a, uint32_t flags) { NS::g2_encode_f(out, a, flags); } \
    __host__ __device__ static size_t wire_size(uint32_t flags) { return NS::g2_wire_size(flags); } \
    __host__ __device__ static size_t out_size(uint32_t flags) { return NS::g2_out_size(flags); } \
    __host__ __device__ static void scalar(uint32_t (&k)[8], const uint8_t* in) { NS::scalar_from_be(k, in); } \
    static void generator(Aff<F>& a) { fp2_load_const<NS::TC>(a.x, NS::CC::G2X); fp2_load_const<NS::TC>(a.y, NS::CC::G2Y); a.inf = false; } \
}; \
/* the suite generator's wire form under `flags` (host side): a same-base batch over it is worth a table at any size */ \
template <class T> \
static std::string PFX##_fb_generator_key(uint32_t flags) { \
    Aff<typename T::F> g; \
    T::generator(g); \
    uint8_t buf[fb::WIRE_MAX]; \
    /* the INPUT form the flags select: compressed unless KYB_F_UNCOMPRESSED */ \
    T::encode(buf, g, (flags & KYB_F_UNCOMPRESSED) ? KYB_F_UNCOMPRESSED_OUT : 0u); \
    return std::string((const char*)buf, T::wire_size(flags)); \
} \
} \

#ifdef KYB_FB_EXTERN
#define KYB_DEFINE_FB_RUN(PFX) \
namespace kyb { \
int PFX##_fb_run(bool g2, size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status, uint32_t flags, \
                 hipStream_t st, const std::string* key); \
}
#else
#define KYB_DEFINE_FB_RUN(PFX) \
namespace kyb { \
inline int PFX##_fb_run(bool g2, size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status, uint32_t flags, \
                        hipStream_t st, const std::string* key) { \
    return g2 ? fb::run<PFX##_FbG2>(n, d_scalars, d_points, d_out, d_status, flags, st, key) \
              : fb::run<PFX##_FbG1>(n, d_scalars, d_points, d_out, d_status, flags, st, key); \
} \
}
#endif
#define KYB_DEFINE_MUL_ABI(PFX, NS, G1SZ, G2SZ) \
KYB_DEFINE_FB_TRAITS(PFX, NS) \
KYB_DEFINE_FB_RUN(PFX) \
namespace kyb { \
__global__ __launch_bounds__(64, KYB_G1_MUL_WAVES) void PFX##_g1_mul_kernel(size_t n, const uint8_t* __restrict__ scalars, \
                                                        const uint8_t* __restrict__ pts, size_t pt_stride, \
                                                        uint8_t* __restrict__ out, uint8_t* __restrict__ status, \
                                                        uint32_t flags, const uint8_t* __restrict__ only, \
                                                        uint32_t* __restrict__ tabs) { \
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; \
    if (idx >= n) return; \
    if (only && !only[idx]) return;  /* the lane machine did this element (bls12381_lvm.cuh step 4) */ \
    const int st = NS::g1_mul_wire(out + NS::g1_out_size(flags) * idx, scalars + 32 * idx, pts + pt_stride * idx, flags, \
                                   tabs + NS::G1_TAB_WORDS * idx); \
    if (status) status[idx] = (uint8_t)st; \
} \
__global__ __launch_bounds__(64, KYB_G2_MUL_WAVES) void PFX##_g2_mul_kernel(size_t n, const uint8_t* __restrict__ scalars, \
                                                        const uint8_t* __restrict__ pts, size_t pt_stride, \
                                                        uint8_t* __restrict__ out, uint8_t* __restrict__ status, \
                                                        uint32_t flags, const uint8_t* __restrict__ only, \
                                                        uint32_t* __restrict__ tabs) { \
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; \
    if (idx >= n) return; \
    if (only && !only[idx]) return;  /* the lane machine did this element (bls12381_lvm.cuh step 4) */ \
    /* tabs: a slab of NS::G2_TAB_WORDS words per lane for the ladder's window tables (suites that keep them in global \
       memory: bn_suite.inc), indexed by the lane's position in THIS launch */ \
    const int st = NS::g2_mul_wire(out + NS::g2_out_size(flags) * idx, scalars + 32 * idx, pts + pt_stride * idx, flags, \
                                   tabs + NS::G2_TAB_WORDS * idx); \
    if (status) status[idx] = (uint8_t)st; \
} \
__global__ __launch_bounds__(64, KYB_TU_WAVES) void PFX##_g1_unmarshal_kernel(size_t n, const uint8_t* __restrict__ pts, \
                                                              uint8_t* __restrict__ out, uint8_t* __restrict__ status, \
                                                              uint32_t flags) { \
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; \
    if (idx >= n) return; \
    const int st = NS::g1_unmarshal_wire(out + NS::g1_out_size(flags) * idx, pts + NS::g1_wire_size(flags) * idx, flags); \
    if (status) status[idx] = (uint8_t)st; \
} \
__global__ __launch_bounds__(64, KYB_TU_WAVES) void PFX##_g2_unmarshal_kernel(size_t n, const uint8_t* __restrict__ pts, \
                                                              uint8_t* __restrict__ out, uint8_t* __restrict__ status, \
                                                              uint32_t flags) { \
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; \
    if (idx >= n) return; \
    const int st = NS::g2_unmarshal_wire(out + NS::g2_out_size(flags) * idx, pts + NS::g2_wire_size(flags) * idx, flags); \
    if (status) status[idx] = (uint8_t)st; \
} \
__global__ __launch_bounds__(64, KYB_TU_WAVES) void PFX##_g1_add_kernel(size_t n, const uint8_t* __restrict__ a, \
                                                        const uint8_t* __restrict__ b, uint8_t* __restrict__ out, \
                                                        uint8_t* __restrict__ status) { \
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; \
    if (idx >= n) return; \
    const int st = NS::g1_add_wire(out + G1SZ * idx, a + G1SZ * idx, b + G1SZ * idx); \
    if (status) status[idx] = (uint8_t)st; \
} \
__global__ __launch_bounds__(64, KYB_TU_WAVES) void PFX##_g2_add_kernel(size_t n, const uint8_t* __restrict__ a, \
                                                        const uint8_t* __restrict__ b, uint8_t* __restrict__ out, \
                                                        uint8_t* __restrict__ status) { \
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; \
    if (idx >= n) return; \
    const int st = NS::g2_add_wire(out + G2SZ * idx, a + G2SZ * idx, b + G2SZ * idx); \
    if (status) status[idx] = (uint8_t)st; \
} \
} \
extern "C" { \
int kyb_##PFX##_g1_mul_dev(size_t n, const void* d_scalars, const void* d_points, size_t point_stride, \
                            void* d_out, void* d_status, uint32_t flags, void* stream) { \
    KYB_TRY(kyb::check_flags(flags, 1, true, "kyb_" #PFX "_g1_mul_dev")); \
    if ((n && (!d_scalars || !d_points || !d_out)) || (point_stride != 0 && point_stride != kyb::NS::g1_wire_size(flags))) { \
        kyb::set_error("kyb_" #PFX "_g1_mul_dev: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    if (point_stride == 0 && n >= kyb::fb_min_batch(false)) /* one base, many scalars: fixed_base.cuh */ \
        return kyb::PFX##_fb_run(false, n, d_scalars, d_points, d_out, d_status, flags, (hipStream_t)stream, nullptr); \
    const uint8_t* only = nullptr; \
    /* the machine's steps and the per-lane redo launch are ONE unit on the stream: `only` points into the (WS_LVM, \
       stream) workspace, which another thread's call on the same stream may rewrite or grow (enq_mu is recursive) */ \
    kyb::DeviceCtx* ctx_; \
    KYB_TRY(kyb::get_ctx(&ctx_)); \
    std::lock_guard<std::recursive_mutex> enq_(ctx_->enq_mu); \
    bool handled_ = false; /* the suite's hook did the whole batch (BLS12-381 G1: the small-batch kernel on cooperating lanes) */ \
    KYB_TRY(kyb::NS::lvm_mul(false, n, (const uint8_t*)d_scalars, (const uint8_t*)d_points, point_stride, (uint8_t*)d_out, \
                             (uint8_t*)d_status, flags, (hipStream_t)stream, &only, &handled_)); \
    if (handled_) return KYB_OK; \
    uint32_t* g1tabs_ = nullptr; \
    if (kyb::NS::G1_TAB_WORDS) { /* per-lane table slab of the ladder: (WS_G1TAB, stream), one launch's worth (this kernel \
                                    only ever runs batches below the lane machine's threshold) */ \
        void* tw_; \
        KYB_TRY(kyb::ctx_workspace(ctx_, kyb::WS_G1TAB, (hipStream_t)stream, n * kyb::NS::G1_TAB_WORDS * sizeof(uint32_t), &tw_)); \
        g1tabs_ = (uint32_t*)tw_; \
    } \
    hipLaunchKernelGGL(kyb::PFX##_g1_mul_kernel, dim3(kyb::grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n, \
                       (const uint8_t*)d_scalars, (const uint8_t*)d_points, point_stride, (uint8_t*)d_out, \
                       (uint8_t*)d_status, flags, only, g1tabs_); \
    KYB_HIP_CHECK(hipGetLastError()); \
    return KYB_OK; \
} \
int kyb_##PFX##_g2_mul_dev(size_t n, const void* d_scalars, const void* d_points, size_t point_stride, \
                            void* d_out, void* d_status, uint32_t flags, void* stream) { \
    KYB_TRY(kyb::check_flags(flags, 1, true, "kyb_" #PFX "_g2_mul_dev")); \
    if ((n && (!d_scalars || !d_points || !d_out)) || (point_stride != 0 && point_stride != kyb::NS::g2_wire_size(flags))) { \
        kyb::set_error("kyb_" #PFX "_g2_mul_dev: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    if (point_stride == 0 && n >= kyb::fb_min_batch(true)) \
        return kyb::PFX##_fb_run(true, n, d_scalars, d_points, d_out, d_status, flags, (hipStream_t)stream, nullptr); \
    const uint8_t* only = nullptr; \
    kyb::DeviceCtx* ctx_; \
    KYB_TRY(kyb::get_ctx(&ctx_)); \
    std::lock_guard<std::recursive_mutex> enq_(ctx_->enq_mu); \
    bool handled_ = false; /* the suite's hook did the whole batch (BLS12-381 G1: the small-batch kernel on cooperating lanes) */ \
    KYB_TRY(kyb::NS::lvm_mul(true, n, (const uint8_t*)d_scalars, (const uint8_t*)d_points, point_stride, (uint8_t*)d_out, \
                             (uint8_t*)d_status, flags, (hipStream_t)stream, &only, &handled_)); \
    if (handled_) return KYB_OK; \
    if (kyb::NS::G2_TAB_WORDS) { \
        /* the per-lane table slab: (WS_G2TAB, stream) workspace, at most 2^20 lanes of it -- larger batches go through it in \
           pieces (stream-ordered: a piece's kernel has read its tables before the next one writes them) */ \
        const size_t piece = n < (size_t(1) << 20) ? n : (size_t(1) << 20); \
        void* tw_; \
        KYB_TRY(kyb::ctx_workspace(ctx_, kyb::WS_G2TAB, (hipStream_t)stream, piece * kyb::NS::G2_TAB_WORDS * sizeof(uint32_t), &tw_)); \
        const size_t osz_ = kyb::NS::g2_out_size(flags); \
        for (size_t off = 0; off < n; off += piece) { \
            const size_t cnt = n - off < piece ? n - off : piece; \
            hipLaunchKernelGGL(kyb::PFX##_g2_mul_kernel, dim3(kyb::grid_for(cnt, 64)), dim3(64), 0, (hipStream_t)stream, cnt, \
                               (const uint8_t*)d_scalars + 32 * off, (const uint8_t*)d_points + point_stride * off, point_stride, \
                               (uint8_t*)d_out + osz_ * off, d_status ? (uint8_t*)d_status + off : nullptr, flags, \
                               (const uint8_t*)nullptr, (uint32_t*)tw_); \
        } \
        KYB_HIP_CHECK(hipGetLastError()); \
        return KYB_OK; \
    } \
    hipLaunchKernelGGL(kyb::PFX##_g2_mul_kernel, dim3(kyb::grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n, \
                       (const uint8_t*)d_scalars, (const uint8_t*)d_points, point_stride, (uint8_t*)d_out, \
                       (uint8_t*)d_status, flags, only, (uint32_t*)nullptr); \
    KYB_HIP_CHECK(hipGetLastError()); \
    return KYB_OK; \
} \
static int PFX##_mul_host(bool g2, size_t n, const uint8_t* scalars, const uint8_t* points, size_t stride, \
                          uint8_t* out, uint8_t* status, uint32_t flags) { \
    KYB_TRY(kyb::check_flags(flags, 1, true, "kyb_" #PFX "_g*_mul")); \
    const size_t psz = g2 ? kyb::NS::g2_out_size(flags) : kyb::NS::g1_out_size(flags); \
    const size_t isz = g2 ? kyb::NS::g2_wire_size(flags) : kyb::NS::g1_wire_size(flags); \
    if (stride) stride = isz; \
    if (n && (!scalars || !points || !out)) { \
        kyb::set_error("kyb_" #PFX "_g*_mul: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    if (kyb::md_active(n)) \
        return kyb::md_run(n, [&](int, size_t lo, size_t hi) { \
            return PFX##_mul_host(g2, hi - lo, scalars + 32 * lo, points + (stride ? isz * lo : 0), stride, out + psz * lo, \
                                  status ? status + lo : nullptr, flags); \
        }); \
    kyb::DeviceCtx* ctx; \
    KYB_TRY(kyb::get_ctx(&ctx)); \
    kyb::StageScope sc_(ctx); \
    kyb::StageBuf s, p, o, st; \
    KYB_TRY(s.upload(scalars, n * 32)); \
    KYB_TRY(p.upload(points, (stride ? n : 1) * isz)); \
    KYB_TRY(o.alloc(n * psz)); \
    KYB_TRY(st.alloc(n)); \
    if (!stride && kyb::fb_min_batch(g2) != ~size_t(0)) { \
        /* same base: the fixed-base table takes the batch when it is large, or from FB_MIN_KNOWN scalars when the \
           table is there already (the base of the previous such call on this device) or is the suite's generator */ \
        const int kind = g2 ? kyb::PFX##_FbG2::KIND : kyb::PFX##_FbG1::KIND; \
        std::string key((const char*)points, isz); \
        key.push_back((char)(flags & 0xff)); \
        key.push_back((char)((flags >> 8) & 0xff)); \
        bool use = n >= kyb::fb_min_batch(g2); \
        if (!use && n >= kyb::FB_MIN_KNOWN) { \
            use = kyb::fb_hint_is(ctx, kind, sc_.stream(), key); \
            if (!use) { \
                std::string gk = g2 ? kyb::PFX##_fb_generator_key<kyb::PFX##_FbG2>(flags) \
                                    : kyb::PFX##_fb_generator_key<kyb::PFX##_FbG1>(flags); \
                use = gk.size() == isz && memcmp(gk.data(), points, isz) == 0; \
            } \
        } \
        if (use) { \
            KYB_TRY(kyb::PFX##_fb_run(g2, n, s.p, p.p, o.p, st.p, flags, sc_.stream(), &key)); \
            KYB_TRY(o.download(out, n * psz)); \
            if (status) KYB_TRY(st.download(status, n)); \
            return KYB_OK; \
        } \
    } \
    if (!stride && n >= (size_t(1) << 18) && !(flags & KYB_F_TRUSTED(0)) && (!g2 || kyb::NS::g2_decode_proves_subgroup())) { \
        /* one shared base and MANY coefficients (PriPoly.Commit): UnmarshalBinary's checks run once, in one lane, and \
           the lanes take the point as validated.  A lone lane needs as long for them (~2 ms on BLS12-381 G1) as a full \
           chip of lanes does side by side, so this pays only once every SIMD has several waves to run one after the \
           other (measured: 2^16 coefficients 6.0 ms per-lane against 6.9 ms this way; from 2^18 on it wins).  Not for a G2 \
           base of a suite whose UnmarshalBinary does not prove subgroup membership (bn256): there TRUSTED selects the GLS \
           walk, which differs from the reference's double-and-add on off-subgroup points whatever n is */ \
        KYB_TRY(g2 ? kyb_##PFX##_g2_unmarshal_dev(1, p.p, o.p, st.p, flags & ~KYB_F_UNCOMPRESSED_OUT, sc_.stream()) \
                   : kyb_##PFX##_g1_unmarshal_dev(1, p.p, o.p, st.p, flags & ~KYB_F_UNCOMPRESSED_OUT, sc_.stream())); \
        uint8_t st0 = 0; \
        KYB_HIP_CHECK(hipMemcpy(&st0, st.p, 1, hipMemcpyDeviceToHost)); \
        if (st0) { \
            memset(out, 0, n * psz); \
            if (status) memset(status, st0, n); \
            return KYB_OK; \
        } \
        flags |= KYB_F_TRUSTED(0); \
    } \
    KYB_TRY(g2 ? kyb_##PFX##_g2_mul_dev(n, s.p, p.p, stride, o.p, st.p, flags, sc_.stream()) \
               : kyb_##PFX##_g1_mul_dev(n, s.p, p.p, stride, o.p, st.p, flags, sc_.stream())); \
    KYB_TRY(o.download(out, n * psz)); \
    if (status) KYB_TRY(st.download(status, n)); \
    return KYB_OK; \
} \
int kyb_##PFX##_g1_mul(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out, uint8_t* status, \
                        uint32_t flags) { \
    return PFX##_mul_host(false, n, scalars, points, G1SZ, out, status, flags); \
} \
int kyb_##PFX##_g2_mul(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out, uint8_t* status, \
                        uint32_t flags) { \
    return PFX##_mul_host(true, n, scalars, points, G2SZ, out, status, flags); \
} \
int kyb_##PFX##_g1_mul_same_base(size_t n, const uint8_t* scalars, const uint8_t* point, uint8_t* out, \
                                  uint8_t* status, uint32_t flags) { \
    return PFX##_mul_host(false, n, scalars, point, 0, out, status, flags); \
} \
int kyb_##PFX##_g2_mul_same_base(size_t n, const uint8_t* scalars, const uint8_t* point, uint8_t* out, \
                                  uint8_t* status, uint32_t flags) { \
    return PFX##_mul_host(true, n, scalars, point, 0, out, status, flags); \
} \
int kyb_##PFX##_g1_unmarshal_dev(size_t n, const void* d_points, void* d_out, void* d_status, uint32_t flags, \
                                  void* stream) { \
    KYB_TRY(kyb::check_flags(flags, 1, true, "kyb_" #PFX "_g1_unmarshal_dev")); \
    if (n && (!d_points || !d_out)) { \
        kyb::set_error("kyb_" #PFX "_g1_unmarshal_dev: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    bool small_ = false; /* the suite's small-batch kernel took it (BLS12-381: cooperating lanes) */ \
    KYB_TRY(kyb::NS::unmarshal_small(false, n, (const uint8_t*)d_points, (uint8_t*)d_out, (uint8_t*)d_status, flags, (hipStream_t)stream, &small_)); \
    if (small_) return KYB_OK; \
    hipLaunchKernelGGL(kyb::PFX##_g1_unmarshal_kernel, dim3(kyb::grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n, \
                       (const uint8_t*)d_points, (uint8_t*)d_out, (uint8_t*)d_status, flags); \
    KYB_HIP_CHECK(hipGetLastError()); \
    return KYB_OK; \
} \
int kyb_##PFX##_g2_unmarshal_dev(size_t n, const void* d_points, void* d_out, void* d_status, uint32_t flags, \
                                  void* stream) { \
    KYB_TRY(kyb::check_flags(flags, 1, true, "kyb_" #PFX "_g2_unmarshal_dev")); \
    if (n && (!d_points || !d_out)) { \
        kyb::set_error("kyb_" #PFX "_g2_unmarshal_dev: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    bool small_ = false; /* the suite's small-batch kernel took it (BLS12-381: cooperating lanes) */ \
    KYB_TRY(kyb::NS::unmarshal_small(true, n, (const uint8_t*)d_points, (uint8_t*)d_out, (uint8_t*)d_status, flags, (hipStream_t)stream, &small_)); \
    if (small_) return KYB_OK; \
    hipLaunchKernelGGL(kyb::PFX##_g2_unmarshal_kernel, dim3(kyb::grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n, \
                       (const uint8_t*)d_points, (uint8_t*)d_out, (uint8_t*)d_status, flags); \
    KYB_HIP_CHECK(hipGetLastError()); \
    return KYB_OK; \
} \
static int PFX##_unmarshal_host(bool g2, size_t n, const uint8_t* points, uint8_t* out, uint8_t* status, \
                                uint32_t flags) { \
    KYB_TRY(kyb::check_flags(flags, 1, true, "kyb_" #PFX "_g*_unmarshal")); \
    const size_t psz = g2 ? kyb::NS::g2_out_size(flags) : kyb::NS::g1_out_size(flags); \
    const size_t isz = g2 ? kyb::NS::g2_wire_size(flags) : kyb::NS::g1_wire_size(flags); \
    if (n && (!points || !out)) { \
        kyb::set_error("kyb_" #PFX "_g*_unmarshal: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    kyb::DeviceCtx* ctx; \
    KYB_TRY(kyb::get_ctx(&ctx)); \
    kyb::StageScope sc_(ctx); \
    kyb::StageBuf p, o, st; \
    KYB_TRY(p.upload(points, n * isz)); \
    KYB_TRY(o.alloc(n * psz)); \
    KYB_TRY(st.alloc(n)); \
    KYB_TRY(g2 ? kyb_##PFX##_g2_unmarshal_dev(n, p.p, o.p, st.p, flags, sc_.stream()) \
               : kyb_##PFX##_g1_unmarshal_dev(n, p.p, o.p, st.p, flags, sc_.stream())); \
    KYB_TRY(o.download(out, n * psz)); \
    if (status) KYB_TRY(st.download(status, n)); \
    return KYB_OK; \
} \
int kyb_##PFX##_g1_unmarshal(size_t n, const uint8_t* points, uint8_t* out, uint8_t* status, uint32_t flags) { \
    return PFX##_unmarshal_host(false, n, points, out, status, flags); \
} \
int kyb_##PFX##_g2_unmarshal(size_t n, const uint8_t* points, uint8_t* out, uint8_t* status, uint32_t flags) { \
    return PFX##_unmarshal_host(true, n, points, out, status, flags); \
} \
static int PFX##_add_host(bool g2, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* status) { \
    const size_t psz = g2 ? G2SZ : G1SZ; \
    if (n && (!a || !b || !out)) { \
        kyb::set_error("kyb_" #PFX "_g*_add: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    kyb::DeviceCtx* ctx; \
    KYB_TRY(kyb::get_ctx(&ctx)); \
    kyb::StageScope sc_(ctx); \
    kyb::StageBuf x, y, o, st; \
    KYB_TRY(x.upload(a, n * psz)); \
    KYB_TRY(y.upload(b, n * psz)); \
    KYB_TRY(o.alloc(n * psz)); \
    KYB_TRY(st.alloc(n)); \
    if (g2) \
        hipLaunchKernelGGL(kyb::PFX##_g2_add_kernel, dim3(kyb::grid_for(n, 64)), dim3(64), 0, sc_.stream(), n, (const uint8_t*)x.p, \
                           (const uint8_t*)y.p, (uint8_t*)o.p, (uint8_t*)st.p); \
    else \
        hipLaunchKernelGGL(kyb::PFX##_g1_add_kernel, dim3(kyb::grid_for(n, 64)), dim3(64), 0, sc_.stream(), n, (const uint8_t*)x.p, \
                           (const uint8_t*)y.p, (uint8_t*)o.p, (uint8_t*)st.p); \
    KYB_HIP_CHECK(hipGetLastError()); \
    KYB_TRY(o.download(out, n * psz)); \
    if (status) KYB_TRY(st.download(status, n)); \
    return KYB_OK; \
} \
int kyb_##PFX##_g1_add_dev(size_t n, const void* d_a, const void* d_b, void* d_out, void* d_status, void* stream) { \
    if (n && (!d_a || !d_b || !d_out)) { \
        kyb::set_error("kyb_" #PFX "_g1_add_dev: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    hipLaunchKernelGGL(kyb::PFX##_g1_add_kernel, dim3(kyb::grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n, (const uint8_t*)d_a, \
                       (const uint8_t*)d_b, (uint8_t*)d_out, (uint8_t*)d_status); \
    KYB_HIP_CHECK(hipGetLastError()); \
    return KYB_OK; \
} \
int kyb_##PFX##_g2_add_dev(size_t n, const void* d_a, const void* d_b, void* d_out, void* d_status, void* stream) { \
    if (n && (!d_a || !d_b || !d_out)) { \
        kyb::set_error("kyb_" #PFX "_g2_add_dev: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    hipLaunchKernelGGL(kyb::PFX##_g2_add_kernel, dim3(kyb::grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, n, (const uint8_t*)d_a, \
                       (const uint8_t*)d_b, (uint8_t*)d_out, (uint8_t*)d_status); \
    KYB_HIP_CHECK(hipGetLastError()); \
    return KYB_OK; \
} \
int kyb_##PFX##_g1_add(size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* status) { \
    return PFX##_add_host(false, n, a, b, out, status); \
} \
int kyb_##PFX##_g2_add(size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* status) { \
    return PFX##_add_host(true, n, a, b, out, status); \
} \
}

// ---- pairing entry points:
//   KYB_DEFINE_GT_ABI         GT exponentiation: the _dev + host entry points over the suite's <pfx>_gt_mul_enqueue
//                             (tower machine, program GTMUL)
//   KYB_DEFINE_PAIR_HOST      the host-buffer Pair / ValidatePairing entry points, which stage and call the suite's
//                             own `_dev` ones (the tower machine)
#define KYB_DEFINE_GT_ABI(PFX, NS, GTSZ) \
extern "C" { \
int kyb_##PFX##_gt_mul_dev(size_t n, const void* d_scalars, const void* d_gt, void* d_out, void* d_status, void* stream) { \
    if (n && (!d_scalars || !d_gt || !d_out)) { \
        kyb::set_error("kyb_" #PFX "_gt_mul_dev: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    /* the suite's GTMUL program on the tower machine (its *_pair translation unit defines the enqueue function) */ \
    return kyb::PFX##_gt_mul_enqueue(n, (const uint8_t*)d_scalars, (const uint8_t*)d_gt, (uint8_t*)d_out, (uint8_t*)d_status, \
                                     (hipStream_t)stream); \
} \
int kyb_##PFX##_gt_mul(size_t n, const uint8_t* scalars, const uint8_t* gt, uint8_t* out, uint8_t* status) { \
    if (n && (!scalars || !gt || !out)) { \
        kyb::set_error("kyb_" #PFX "_gt_mul: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    kyb::DeviceCtx* ctx; \
    KYB_TRY(kyb::get_ctx(&ctx)); \
    kyb::StageScope sc_(ctx); \
    kyb::StageBuf a, b, o, st; \
    KYB_TRY(a.upload(scalars, n * 32)); \
    KYB_TRY(b.upload(gt, n * GTSZ)); \
    KYB_TRY(o.alloc(n * GTSZ)); \
    KYB_TRY(st.alloc(n)); \
    KYB_TRY(kyb_##PFX##_gt_mul_dev(n, a.p, b.p, o.p, st.p, sc_.stream())); \
    KYB_TRY(o.download(out, n * GTSZ)); \
    if (status) KYB_TRY(st.download(status, n)); \
    return KYB_OK; \
} \
}

#define KYB_DEFINE_PAIR_HOST(PFX, NS, GTSZ) \
extern "C" { \
int kyb_##PFX##_pair(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, uint8_t* status, uint32_t flags) { \
    KYB_TRY(kyb::check_flags(flags, 2, false, "kyb_" #PFX "_pair")); \
    if (n && (!g1 || !g2 || !gt)) { \
        kyb::set_error("kyb_" #PFX "_pair: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    if (kyb::md_active(n)) \
        return kyb::md_run(n, [&](int, size_t lo, size_t hi) { \
            return kyb_##PFX##_pair(hi - lo, g1 + kyb::NS::g1_wire_size(flags) * lo, g2 + kyb::NS::g2_wire_size(flags) * lo, \
                                    gt + (size_t)GTSZ * lo, status ? status + lo : nullptr, flags); \
        }); \
    kyb::DeviceCtx* ctx; \
    KYB_TRY(kyb::get_ctx(&ctx)); \
    kyb::StageScope sc_(ctx); \
    kyb::StageBuf a, b, o, st; \
    KYB_TRY(a.upload(g1, n * kyb::NS::g1_wire_size(flags))); \
    KYB_TRY(b.upload(g2, n * kyb::NS::g2_wire_size(flags))); \
    KYB_TRY(o.alloc(n * GTSZ)); \
    KYB_TRY(st.alloc(n)); \
    KYB_TRY(kyb_##PFX##_pair_dev(n, a.p, b.p, o.p, st.p, flags, sc_.stream())); \
    KYB_TRY(o.download(gt, n * GTSZ)); \
    if (status) KYB_TRY(st.download(status, n)); \
    return KYB_OK; \
} \
int kyb_##PFX##_pair_check(size_t n, const uint8_t* p1, const uint8_t* p2, const uint8_t* inv1, const uint8_t* inv2, \
                            uint8_t* ok, uint8_t* status, uint32_t flags) { \
    KYB_TRY(kyb::check_flags(flags, 4, false, "kyb_" #PFX "_pair_check")); \
    if (n && (!p1 || !p2 || !inv1 || !inv2 || !ok)) { \
        kyb::set_error("kyb_" #PFX "_pair_check: bad argument"); \
        return KYB_E_ARG; \
    } \
    if (!n) return KYB_OK; \
    if (kyb::md_active(n)) \
        return kyb::md_run(n, [&](int, size_t lo, size_t hi) { \
            const size_t s1 = kyb::NS::g1_wire_size(flags), s2 = kyb::NS::g2_wire_size(flags); \
            return kyb_##PFX##_pair_check(hi - lo, p1 + s1 * lo, p2 + s2 * lo, inv1 + s1 * lo, inv2 + s2 * lo, ok + lo, \
                                          status ? status + lo : nullptr, flags); \
        }); \
    kyb::DeviceCtx* ctx; \
    KYB_TRY(kyb::get_ctx(&ctx)); \
    kyb::StageScope sc_(ctx); \
    kyb::StageBuf a, b, c, d, o, st; \
    KYB_TRY(a.upload(p1, n * kyb::NS::g1_wire_size(flags))); \
    KYB_TRY(b.upload(p2, n * kyb::NS::g2_wire_size(flags))); \
    KYB_TRY(c.upload(inv1, n * kyb::NS::g1_wire_size(flags))); \
    KYB_TRY(d.upload(inv2, n * kyb::NS::g2_wire_size(flags))); \
    KYB_TRY(o.alloc(n)); \
    KYB_TRY(st.alloc(n)); \
    KYB_TRY(kyb_##PFX##_pair_check_dev(n, a.p, b.p, c.p, d.p, o.p, st.p, flags, sc_.stream())); \
    KYB_TRY(o.download(ok, n)); \
    if (status) KYB_TRY(st.download(status, n)); \
    return KYB_OK; \
} \
}
