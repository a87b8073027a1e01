// Same-base batches of the BLS12-381 groups (share.PriPoly.Commit, share/poly.go:143-149; key generation; Mul(s, nil))
// through the fixed-base tables of fixed_base.cuh under the endomorphism policies of bls12381_fb.cuh -- in a
// translation unit of their own.  Out-of-line device code is compiled for the loosest register budget of the kernels
// that reach it: next to the ladders and cooperating-lane kernels of bls12381.hip, which want 512 registers,
// fb::mul_kernel<FbG2> (two-wave launch bounds) came out at 288 registers -- one wave per SIMD.  Here, where only the
// fixed-base kernels share that code, it is 256: 2^18 G2 commits 5.61 -> 5.0 ms.  (The table, chain and encode kernels
// keep their own looser budgets: all five kernels on two waves measured 2.5 % slower, profiles/r04_tu_wave_budgets.json.)
#include "bls12381.cuh"
#include "rowfp.cuh"
#include "bls12381_fb.cuh"
#include "pairing_abi.cuh"

KYB_DEFINE_FB_TRAITS(bls12381, bls)

namespace kyb {
int bls12381_fb_run(bool g2, size_t n, const void* d_scalars, const void* d_points, void* d_out, void* d_status, uint32_t flags, hipStream_t st,
                    const std::string* key) {
    return g2 ? fb::run<bls12381_FbG2>(n, d_scalars, d_points, d_out, d_status, flags, st, key)
              : fb::run<bls12381_FbG1>(n, d_scalars, d_points, d_out, d_status, flags, st, key);
}
}  // namespace kyb
