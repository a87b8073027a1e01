// bls12381: multi-scalar multiplication entry points (msm.cuh pipeline over the Weierstrass adapter).
#include "bls12381.cuh"
#include "pairing_abi.cuh"
#include "rowfp.cuh"
#include "msm_ws.cuh"
#include "bls12381_msm_codec.cuh"
#include "msm_adapters.h"
namespace kyb {
// G1 with a balanced GLV split: every (k, P) becomes (|k0|, +-P) and (|k1|, +-z^2 P), z^2 P = (beta x, -y), with
// k = k1 z^2 + k0 (mod r) and |k0|, |k1| <= z^2 / 2 + 2 < 2^126.5.  Start from the long division k = q z^2 + rem, move
// rem into (-z^2/2, z^2/2] (q += 1), then fold q with z^4 = z^2 - 1 (mod r): (q, rem) -> (q - z^2 + 1, rem - 1), at
// most twice for k < 2^256.  Balanced halves matter: their top 16-bit window stays below 2^15, so the signed recoding
// never carries into a ninth window -- which would put a quarter of all points into one bucket.  8 windows of 16 bits
// replace 17 and the serial doubling chain of the tail shrinks from 256 to 112.  Valid because every accepted P is in
// G1 (checked, or vouched for by KYB_F_TRUSTED).
struct BlsG1Msm : msm::Weierstrass<bls::fp, BlsG1Codec> {
    using Base = msm::Weierstrass<bls::fp, BlsG1Codec>;
    static constexpr int SPLIT = 2, SPLIT_BITS = 127;
    static constexpr int ROW_FINAL = 1;  // msm.cuh final_rows_kernel: the window sums' doubling chains on rowfp.cuh
    using RowC = bls::FC;
#ifndef KYB_BLS_G1_DECODE_WAVES
#define KYB_BLS_G1_DECODE_WAVES 2
#endif
    static constexpr int DECODE_WAVES = KYB_BLS_G1_DECODE_WAVES;  // register budget of decode_kernel (the square root and the subgroup test)
    // 160-bit two's-complement helpers (five words)
    __device__ static void add5(uint32_t (&x)[5], const uint32_t (&y)[5]) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) x[i] = adc32(x[i], y[i], c);
    }
    __device__ static void sub5(uint32_t (&x)[5], const uint32_t (&y)[5]) {
        uint32_t b = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) x[i] = sbb32(x[i], y[i], b);
    }
    __device__ static bool gt5(const uint32_t (&x)[5], const uint32_t (&y)[5]) {  // signed x > y
        uint32_t t[5];
        uint32_t b = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) t[i] = sbb32(y[i], x[i], b);  // y - x < 0 ?
        return (t[4] >> 31) != 0;  // |x|, |y| < 2^130: no overflow
    }
    __device__ static bool abs5(uint32_t (&x)[5]) {  // x = |x|, returns the sign
        const bool neg = (x[4] >> 31) != 0;
        uint32_t z[5] = {0, 0, 0, 0, 0};
        sub5(z, x);
#pragma unroll
        for (int i = 0; i < 5; i++) x[i] = neg ? z[i] : x[i];
        return neg;
    }
    __device__ static int decode_split(Aff (&a)[2], uint32_t (&k)[2][8], const uint8_t* pt, const uint8_t* scalar,
                                       uint32_t flags) {
        const int st = Base::decode(a[0], pt, flags);
        split_halves(a, k, scalar);
        return st;
    }
    // KYB_F_UNCOMPRESSED | KYB_F_TRUSTED(0), the form a resident pipeline keeps its points in: a decode kernel of its own
    // (msm.cuh decode_kernel<A, true>) without the square root, the curve equation and the subgroup rule in its code --
    // a third of the registers, twice the waves to hide the loads behind
    static constexpr int LIGHT_DECODE_WAVES = 4;
    __device__ static int decode_split_light(Aff (&a)[2], uint32_t (&k)[2][8], const uint8_t* pt, const uint8_t* scalar) {
        bls::g1_aff t;
        const int st = bls::g1_decode_unc_trusted(t, pt);
        a[0].x = t.x;
        a[0].y = t.y;
        a[0].inf = t.inf ? 1u : 0u;
        split_halves(a, k, scalar);
        return st;
    }
    __device__ __forceinline__ static void split_halves(Aff (&a)[2], uint32_t (&k)[2][8], const uint8_t* scalar) {
        uint32_t kk[8], q8[8], rem[4];
        Base::scalar_words(kk, scalar);
        bls::divmod_z<4>(q8, rem, kk);
        const uint32_t Z2[5] = {bls::ZDiv<4>::D[0], bls::ZDiv<4>::D[1], bls::ZDiv<4>::D[2], bls::ZDiv<4>::D[3], 0u};  // z^2
        const uint32_t HALF[5] = {0x80000000u, 0x00000000u, 0x8000d201u, 0x5622d200u, 0u};  // z^2 / 2
        const uint32_t ONE[5] = {1u, 0u, 0u, 0u, 0u};
        uint32_t q[5] = {q8[0], q8[1], q8[2], q8[3], q8[4]};  // q < 2^129
        uint32_t r[5] = {rem[0], rem[1], rem[2], rem[3], 0u};
        if (gt5(r, HALF)) {
            sub5(r, Z2);
            add5(q, ONE);
        }
#pragma unroll 1
        for (int it = 0; it < 3; it++) {
            if (gt5(q, HALF)) {  // q z^2 = (q - z^2) z^2 + z^4 and z^4 = z^2 - 1 (mod r)
                sub5(q, Z2);
                add5(q, ONE);
                sub5(r, ONE);
            }
        }
        const bool n0 = abs5(r), n1 = abs5(q);
        bls::fp beta, ny;
        bls::fp_const(beta, bls::CC::BETA);
        fp_mul(a[1].x, a[0].x, beta);
        a[1].y = a[0].y;
        a[1].inf = a[0].inf;
        fp_neg(ny, a[0].y);
        fp_cmov(a[1].y, ny, !n1);  // z^2 P = (beta x, -y); a negative k1 flips it back
        fp_cmov(a[0].y, ny, n0);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            k[0][i] = i < 5 ? r[i] : 0u;
            k[1][i] = i < 5 ? q[i] : 0u;
        }
    }
};
// short scalars (KYB_F_SCALAR_BITS(b), b <= 160) take the plain G1 adapter of bls12381_msm_plain.hip
inline bool bls_g1_msm_plain(uint32_t flags) {
    const uint32_t want = (flags >> 16) & 0x1ffu;
    return want != 0 && want <= 160;
}
using BlsG2Msm = msm::Weierstrass<bls::fp2, BlsG2Codec>;
// which adapter a call takes: the quarters for scalars of full length, the plain windows for scalars cut short
inline bool bls_g2_msm_gls(uint32_t flags, size_t n) {
    static const int mode = [] {  // KYB_BLS_G2_MSM_GLS=0: never; =1: scalars of full length only; =2: always (A/B)
        const char* e = getenv("KYB_BLS_G2_MSM_GLS");
        return e ? atoi(e) : -1;
    }();
    const bool full = ((flags >> 16) & 0x1ffu) == 0;
    if (mode >= 0) return mode == 2 || (mode == 1 && full);
    // scalars cut short by KYB_F_SCALAR_BITS (bdn's 128-bit coefficients: two quarters and a bit): the quarters up to 2^15
    // points, where the tail is the call (3 000 keys: 5.6 -> 4.7 ms); the plain windows above (2^20: 11.9 against 16.5 ms)
    return full || n <= (size_t(1) << 15);
}
}  // namespace kyb

extern "C" {
int kyb_bls12381_g1_msm(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out[48], uint8_t* status,
                         uint32_t flags) {
    if (kyb::bls_g1_msm_plain(flags)) return kyb::bls12381_g1_msm_plain_host(n, scalars, points, out, status, flags);
    return kyb::msm::run_host<kyb::BlsG1Msm>(n, scalars, points, out, status, flags);
}
int kyb_bls12381_g2_msm(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out[96], uint8_t* status,
                         uint32_t flags) {
    if (kyb::bls_g2_msm_gls(flags, n)) return kyb::bls12381_g2_msm_gls_host(n, scalars, points, out, status, flags);
    return kyb::msm::run_host<kyb::BlsG2Msm>(n, scalars, points, out, status, flags);
}
int kyb_bls12381_g1_msm_dev(size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
                            uint32_t flags, void* stream) {
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    if (kyb::bls_g1_msm_plain(flags))
        return kyb::bls12381_g1_msm_plain_dev(ctx, n, d_scalars, d_points, d_out, d_status, (hipStream_t)stream, flags);
    return kyb::msm::run<kyb::BlsG1Msm>(ctx, n, d_scalars, d_points, d_out, d_status, (hipStream_t)stream, flags);
}
int kyb_bls12381_g2_msm_dev(size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
                            uint32_t flags, void* stream) {
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    if (kyb::bls_g2_msm_gls(flags, n))
        return kyb::bls12381_g2_msm_gls_dev(ctx, n, d_scalars, d_points, d_out, d_status, (hipStream_t)stream, flags);
    return kyb::msm::run<kyb::BlsG2Msm>(ctx, n, d_scalars, d_points, d_out, d_status, (hipStream_t)stream, flags);
}
int kyb_bls12381_g1_poly_eval(size_t n, const uint32_t* idx, size_t t, const uint8_t* commits, uint8_t* out, uint8_t* status,
                       uint32_t flags) {
    return kyb::msm::poly_eval_host<kyb::BlsG1Msm>(n, idx, t, commits, out, status, flags);
}
int kyb_bls12381_g1_poly_eval_dev(size_t n, const void* d_idx, size_t t, const void* d_commits, void* d_out, void* d_status,
                           uint32_t flags, void* stream) {
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    return kyb::msm::poly_eval_run<kyb::BlsG1Msm>(ctx, n, d_idx, t, d_commits, d_out, d_status, flags, (hipStream_t)stream);
}
int kyb_bls12381_g2_poly_eval(size_t n, const uint32_t* idx, size_t t, const uint8_t* commits, uint8_t* out, uint8_t* status,
                       uint32_t flags) {
    return kyb::msm::poly_eval_host<kyb::BlsG2Msm>(n, idx, t, commits, out, status, flags);
}
int kyb_bls12381_g2_poly_eval_dev(size_t n, const void* d_idx, size_t t, const void* d_commits, void* d_out, void* d_status,
                           uint32_t flags, void* stream) {
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    return kyb::msm::poly_eval_run<kyb::BlsG2Msm>(ctx, n, d_idx, t, d_commits, d_out, d_status, flags, (hipStream_t)stream);
}
}
