// bls12381: the G2 MSM adapter on balanced GLS quarters (kyb_bls12381_g2_msm of full-length scalars).
// A translation unit of its own: two adapters over one field in one unit share their inlined helpers, the compiler stops
// inlining them, and BOTH accumulate kernels pay (accumulate_kernel<BlsG1Msm> went from 209 registers and no scratch to
// 248 and 320 B -- 2.63 -> 3.45 ms per 2^20 points -- when BlsG1MsmPlain sat beside it: DESIGN.md section 5 item 58 j).
#include "bls12381.cuh"
#include "pairing_abi.cuh"
#include "msm_ws.cuh"
#include "bls12381_msm_codec.cuh"
#include "msm_adapters.h"
namespace kyb {
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
struct BlsG2MsmGls : msm::Weierstrass<bls::fp2, BlsG2Codec> {
    using Base = msm::Weierstrass<bls::fp2, BlsG2Codec>;
    static constexpr int SPLIT = 4, SPLIT_BITS = 63;
    __device__ static int decode_split(Aff (&a)[4], uint32_t (&k)[4][8], const uint8_t* pt, const uint8_t* scalar, uint32_t flags) {
        const int st = Base::decode(a[0], pt, flags);
        uint32_t kk[8];
        Base::scalar_words(kk, scalar);
        const int want = (int)((flags >> 16) & 0x1ffu);  // KYB_F_SCALAR_BITS(b): the bits from b up are ignored
        if (want && want < 256) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int lo = 32 * i;
                kk[i] &= want >= lo + 32 ? 0xffffffffu : (want > lo ? (1u << (want - lo)) - 1u : 0u);
            }
        }
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
int bls12381_g2_msm_gls_host(size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out, uint8_t* status, uint32_t flags) {
    return msm::run_host<BlsG2MsmGls>(n, scalars, points, out, status, flags);
}
int bls12381_g2_msm_gls_dev(DeviceCtx* ctx, size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status,
                            hipStream_t st, uint32_t flags) {
    return msm::run<BlsG2MsmGls>(ctx, n, d_scalars, d_points, d_out, d_status, st, flags);
}
}  // namespace kyb
