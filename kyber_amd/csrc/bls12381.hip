// BLS12-381 batch kernels for gfx950 + their C-ABI entry points (stamped out by pairing_abi.cuh):
// one group operation / pairing per lane, integer VALU only, uniform control flow inside a wave
// except for rejected inputs.
//
// Replaces, behind pairing/bls12381/kilic (the adapter whose Point()/Pair() the suite exposes):
//   G1Elt.Mul / G2Elt.Mul            kilic/g1.go:110-116, g2.go  -> bls12381_g1_mul_kernel / _g2_mul_kernel
//   Suite.Pair                       kilic/suite.go:70-75        -> bls12381_pair_kernel
//   Suite.ValidatePairing            kilic/suite.go:57-68        -> bls12381_pair_check_kernel
//   Unmarshal/MarshalBinary          kilic/g1.go:119-131         -> fused into every kernel
// (this translation unit: G1 / G2 scalar multiplication; pairing kernels are in bls12381_pair.hip, MSM in
//  bls12381_msm.hip -- split only so that the three compile in parallel)
// (same-base batches: bls12381_fb.hip -- a unit of its own for its register budget)
#define KYB_FB_EXTERN
#include "bls12381.cuh"
#include "bls12381_lvm.cuh"
#include "bls12381_fb.cuh"
#include "pairing_abi.cuh"

KYB_DEFINE_MUL_ABI(bls12381, bls, 48, 96)

// Debugging aid of the lane machine (tests/test_gpu_lane_vm.py): the multiplication of bls12381_lvm.cuh with the
// interpreter's trace switched on -- lanes 0 and 1 of the first wave store the result of every record that writes a
// slot, [record][lane][16] int32 -- so that a divergence from gen_lane_vm.py's simulator is found at the first record
// that differs.  Device pointers; the batch must be large enough for the machine (KYB_LVM_MIN).
extern "C" int kyb_debug_bls12381_lvm_trace(int g2, size_t n, const void* d_scalars, const void* d_points, size_t point_stride, void* d_out,
                                            void* d_status, uint32_t flags, void* d_trace, void* stream) {
    const uint8_t* only = nullptr;
    KYB_TRY(kyb::bls::lvm_mul(g2 != 0, n, (const uint8_t*)d_scalars, (const uint8_t*)d_points, point_stride, (uint8_t*)d_out,
                              (uint8_t*)d_status, flags, (hipStream_t)stream, &only, nullptr, (int32_t*)d_trace));
    return only ? KYB_OK : KYB_E_ARG;
}
// The machine's batch threshold for both groups (< 0: back to the built-in rule); tests and A/B runs.
extern "C" int kyb_debug_bls12381_lvm_min(long long n) {
    kyb::bls::lvm_min_override() = n;
    return KYB_OK;
}
// Runs an arbitrary lane-machine program (device pointers: records, schedule entries of four words, constants) on
// `nlanes` lanes whose inputs, digits, table and outputs live in one scratch allocation made here: the micro-benchmark
// of tools/lvm_microbench.py (cost of a record by kind).  Returns the kernel's time in microseconds through *usec.
extern "C" int kyb_debug_bls12381_lvm_run(int pair, size_t nlanes, const void* d_prog, const void* d_sched, uint32_t nsched,
                                          const void* d_consts, int reps, float* usec) {
    using namespace kyb;
    const uint32_t ncoord = pair ? LVM_BLS12381_G2_MUL_NCOORD : LVM_BLS12381_G1_MUL_NCOORD, nentry = LVM_BLS12381_G1_MUL_NENTRY;
    const size_t b_io = 2 * nlanes * 48, b_dig = nlanes * 72, b_tab = nlanes * (size_t)nentry * ncoord * lvm::TAB_WORDS * 4;
    // (every early return releases what was acquired before it)
    struct Res {
        uint8_t* base = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Res() {
            if (e0) hipEventDestroy(e0);
            if (e1) hipEventDestroy(e1);
            if (base) hipFree(base);
        }
    } res;
    KYB_HIP_CHECK(hipMalloc(&res.base, 2 * b_io + b_dig + nlanes + b_tab + 1024));
    uint8_t* base = res.base;
    KYB_HIP_CHECK(hipMemset(base, 1, 2 * b_io + b_dig + nlanes + b_tab + 1024));
    lvm::Args a{};
    a.prog = (const uint32_t*)d_prog;
    a.sched = (const lvm::Sched*)d_sched;
    a.nsched = nsched;
    a.consts = (const int32_t*)d_consts;
    a.in = (const uint32_t*)base;
    a.out = (uint32_t*)(base + b_io);
    a.digits = base + 2 * b_io;
    a.dstride = 72;
    a.zflag = base + 2 * b_io + b_dig;
    a.table = (int32_t*)(base + ((2 * b_io + b_dig + nlanes + 255) & ~size_t(255)));
    a.nentry = nentry;
    a.ncoord = ncoord;
    a.nlanes = nlanes;
    KYB_HIP_CHECK(hipEventCreate(&res.e0));
    KYB_HIP_CHECK(hipEventCreate(&res.e1));
    const unsigned gw = (unsigned)((nlanes + 63) / 64);
    for (int r = 0; r <= reps; r++) {
        if (r == 1) KYB_HIP_CHECK(hipEventRecord(res.e0, nullptr));
        if (pair) hipLaunchKernelGGL((kyb::bls::bls12381_lvm_mul_kernel<true, true>), dim3(gw), dim3(64), 0, nullptr, a);
        else hipLaunchKernelGGL((kyb::bls::bls12381_lvm_mul_kernel<false, true>), dim3(gw), dim3(64), 0, nullptr, a);
    }
    KYB_HIP_CHECK(hipEventRecord(res.e1, nullptr));
    KYB_HIP_CHECK(hipEventSynchronize(res.e1));
    float ms = 0;
    KYB_HIP_CHECK(hipEventElapsedTime(&ms, res.e0, res.e1));
    *usec = ms * 1e3f / (reps > 0 ? reps : 1);
    return KYB_OK;
}
