// bls12381: multi-scalar multiplication entry points (msm.cuh pipeline over the Weierstrass adapter).
#include "bls12381.cuh"
#include "pairing_abi.cuh"
#include "rowfp.cuh"
#include "msm_ws.cuh"
namespace kyb {
struct BlsG1Codec {
    static constexpr int WIRE = 48;
    __host__ __device__ static size_t wire_size(uint32_t flags) { return bls::g1_wire_size(flags); }
    __device__ static int decode(bls::g1_aff& a, const uint8_t* in, uint32_t flags) {
        return bls::g1_decode_f(a, in, flags, 0);
    }
    __device__ static void encode(uint8_t* out, const bls::g1_aff& a) { bls::g1_encode(out, a); }
};
struct BlsG2Codec {
    static constexpr int WIRE = 96;
    __host__ __device__ static size_t wire_size(uint32_t flags) { return bls::g2_wire_size(flags); }
    __device__ static int decode(bls::g2_aff& a, const uint8_t* in, uint32_t flags) {
        return bls::g2_decode_f(a, in, flags, 0);
    }
    __device__ static void encode(uint8_t* out, const bls::g2_aff& a) { bls::g2_encode(out, a); }
};
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
// G2 on balanced GLS quarters (round 6): psi(Q) = [z] Q on every accepted point (UnmarshalBinary proves the subgroup, or the
// caller vouches for it), z = -|z|, so with k = a0 + a1 |z| + a2 |z|^2 + a3 |z|^3 (three long divisions, as g2_mul_gls)
//     k Q = a0 Q - a1 psi(Q) + a2 psi^2(Q) - a3 psi^3(Q).
// The quarters are moved into (-|z| / 2, |z| / 2] with a carry into the next one; what a3 cannot hold (k < 2^256 leaves it
// up to 2.3 |z|) goes into an a4, and |z|^4 = z^2 - 1 (mod r) folds that back: a2 += a4, a0 -= a4.  All |a_i| < 2^63 then:
// four windows of 16 bits whose top one stays below 2^15 -- no carry into a fifth -- instead of 18 windows of 15: the
// reduce runs over 4 x 2^15 buckets instead of 18 x 2^14, the doubling chains are 62 long instead of 269, and a point is
// 16 window visits instead of 18.  Calls with KYB_F_SCALAR_BITS keep the plain adapter (kyb_bls12381_g2_msm below): a
// 128-bit coefficient is 9 window visits there, and here its third quarter is 0 or 1 -- a third of all points in ONE
// bucket (measured: 9.1 against 7.5 ms for 2^18 points).
// G1 WITHOUT the split, for calls that say their scalars are short (KYB_F_SCALAR_BITS(b), b <= 160: bdn's 128-bit
// coefficients, sign/bdn/bdn.go:126-161 on a G1 signature scheme).  On halves a 128-bit k is k0 + k1 z^2 with k1 in {0, +-1}:
// two fifths of all points in ONE bucket of the second half, and the call took LONGER than with full scalars (4.85
// against 3.83 ms for 2^20 points).  Plain windows: ceil(129 / 16) = 9 visits per point instead of 16, the same tail
// kernels (limb-per-lane chains, light decode).
struct BlsG1MsmPlain : msm::Weierstrass<bls::fp, BlsG1Codec> {
    using Base = msm::Weierstrass<bls::fp, BlsG1Codec>;
    static constexpr int ROW_FINAL = 1;
    using RowC = bls::FC;
    static constexpr int DECODE_WAVES = KYB_BLS_G1_DECODE_WAVES;
    static constexpr int LIGHT_DECODE_WAVES = 4;
    __device__ static int decode_split_light(Aff (&a)[1], uint32_t (&k)[1][8], const uint8_t* pt, const uint8_t* scalar) {
        bls::g1_aff t;
        const int st = bls::g1_decode_unc_trusted(t, pt);
        a[0].x = t.x;
        a[0].y = t.y;
        a[0].inf = t.inf ? 1u : 0u;
        Base::scalar_words(k[0], scalar);
        return st;
    }
};
inline bool bls_g1_msm_plain(uint32_t flags) {
    const uint32_t want = (flags >> 16) & 0x1ffu;
    return want != 0 && want <= 160;
}
using BlsG2Msm = msm::Weierstrass<bls::fp2, BlsG2Codec>;
struct BlsG2MsmGls : msm::Weierstrass<bls::fp2, BlsG2Codec> {
    using Base = msm::Weierstrass<bls::fp2, BlsG2Codec>;
    static constexpr int SPLIT = 4, SPLIT_BITS = 63;
    __device__ static int decode_split(Aff (&a)[4], uint32_t (&k)[4][8], const uint8_t* pt, const uint8_t* scalar, uint32_t flags) {
        const int st = Base::decode(a[0], pt, flags);
        uint32_t kk[8];
        Base::scalar_words(kk, scalar);
        uint32_t q1[8], q2[8], q3[8], r0[2], r1[2], r2[2];
        bls::divmod_z<2>(q1, r0, kk);
        bls::divmod_z<2>(q2, r1, q1);
        bls::divmod_z<2>(q3, r2, q2);  // q3 = a3 < 2^65
        using i128 = __int128;
        const i128 Z = (i128)0xd201000000010000ull, H = Z >> 1;
        i128 A[4] = {(i128)(((uint64_t)r0[1] << 32) | r0[0]), (i128)(((uint64_t)r1[1] << 32) | r1[0]),
                     (i128)(((uint64_t)r2[1] << 32) | r2[0]),
                     (i128)(((unsigned __int128)q3[2] << 64) | ((uint64_t)q3[1] << 32) | q3[0])};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (A[i] > H) {
                A[i] -= Z;
                A[i + 1] += 1;
            }
        }
        i128 a4 = 0;
#pragma unroll
        for (int it = 0; it < 4; it++) {
            if (A[3] > H) {
                A[3] -= Z;
                a4 += 1;
            }
        }
        A[2] += a4;  // |z|^4 = z^2 - 1 (mod r)
        A[0] -= a4;
        // the images: psi(x, y) = (cx conj x, cy conj y), psi^2 = (N(cx) x, N(cy) y), psi^3 = (cx N(cx) conj x, cy N(cy) conj y)
        bls::fp2 cx, cy, t;
        bls::fp nx, ny, u;
        fp2_load_const<bls::TC>(cx, bls::CC::PSI_CX);
        fp2_load_const<bls::TC>(cy, bls::CC::PSI_CY);
        fp_sqr(nx, cx.c0);
        fp_sqr(u, cx.c1);
        fp_add(nx, nx, u);
        fp_sqr(ny, cy.c0);
        fp_sqr(u, cy.c1);
        fp_add(ny, ny, u);
        fp2_conj(t, a[0].x);
        fp2_mul_c(a[1].x, t, cx);
        fp2_conj(t, a[0].y);
        fp2_mul_c(a[1].y, t, cy);
        fp2_mul_fp(a[2].x, a[0].x, nx);
        fp2_mul_fp(a[2].y, a[0].y, ny);
        fp2_mul_fp(a[3].x, a[1].x, nx);
        fp2_mul_fp(a[3].y, a[1].y, ny);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool neg = A[i] < 0;
            const unsigned __int128 m = (unsigned __int128)(neg ? -A[i] : A[i]);
            a[i].inf = a[0].inf;
            bls::fp2 ny2;
            fp2_neg(ny2, a[i].y);
            fp2_cmov(a[i].y, ny2, neg != ((i & 1) != 0));  // |z|^i Q = (-1)^i psi^i(Q)
#pragma unroll
            for (int j = 0; j < 8; j++) k[i][j] = j < 2 ? (uint32_t)(m >> (32 * j)) : 0u;
        }
        return st;
    }
};
// which adapter a call takes: the quarters for scalars of full length, the plain windows for scalars cut short
inline bool bls_g2_msm_gls(uint32_t flags) {
    static const bool off = [] {  // KYB_BLS_G2_MSM_GLS=0: never (A/B)
        const char* e = getenv("KYB_BLS_G2_MSM_GLS");
        return e && e[0] == '0';
    }();
    return !off && ((flags >> 16) & 0x1ffu) == 0;
}
}  // namespace kyb

extern "C" {
int kyb_bls12381_g1_msm(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out[48], uint8_t* status,
                         uint32_t flags) {
    if (kyb::bls_g1_msm_plain(flags)) return kyb::msm::run_host<kyb::BlsG1MsmPlain>(n, scalars, points, out, status, flags);
    return kyb::msm::run_host<kyb::BlsG1Msm>(n, scalars, points, out, status, flags);
}
int kyb_bls12381_g2_msm(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out[96], uint8_t* status,
                         uint32_t flags) {
    if (kyb::bls_g2_msm_gls(flags)) return kyb::msm::run_host<kyb::BlsG2MsmGls>(n, scalars, points, out, status, flags);
    return kyb::msm::run_host<kyb::BlsG2Msm>(n, scalars, points, out, status, flags);
}
int kyb_bls12381_g1_msm_dev(size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
                            uint32_t flags, void* stream) {
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    if (kyb::bls_g1_msm_plain(flags))
        return kyb::msm::run<kyb::BlsG1MsmPlain>(ctx, n, d_scalars, d_points, d_out, d_status, (hipStream_t)stream, flags);
    return kyb::msm::run<kyb::BlsG1Msm>(ctx, n, d_scalars, d_points, d_out, d_status, (hipStream_t)stream, flags);
}
int kyb_bls12381_g2_msm_dev(size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
                            uint32_t flags, void* stream) {
    kyb::DeviceCtx* ctx;
    KYB_TRY(kyb::get_ctx(&ctx));
    if (kyb::bls_g2_msm_gls(flags))
        return kyb::msm::run<kyb::BlsG2MsmGls>(ctx, n, d_scalars, d_points, d_out, d_status, (hipStream_t)stream, flags);
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
