// Where does a lone wave lose time in the pairing kernels?  Each kernel below runs ONE wave per SIMD (1024 waves) and
// repeats one building block of the BLS12-381 pairing exactly as the product kernels call it (same out-of-line
// functions, operands behind references).  cycles_per_call / (static VALU count of the callee, tools/isa_mem_profile.py)
// = cycles per VALU instruction: 3.6 is a saturated SIMD (tools/fpmul_probe.hip), 4.4 a lone wave with dependency
// stalls only; anything above that is the operand traffic of the call boundary.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ikyber_amd/csrc -Iinclude tools/tower_probe.hip -o tools/tower_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "bls12381.cuh"

using namespace kyb;
using namespace kyb::bls;

__device__ __forceinline__ void seed_fp(fp& x, uint32_t s) {
#pragma unroll
    for (int l = 0; l < FC::NWORDS; l++) x.v[l] = (threadIdx.x * 2654435761u + l * 40503u + s * 977u) & 0x0FFFFFFFu;
}
__device__ __forceinline__ void seed_fp2(fp2& x, uint32_t s) { seed_fp(x.c0, s); seed_fp(x.c1, s + 1); }
__device__ __forceinline__ void seed_fp6(fp6& x, uint32_t s) { seed_fp2(x.c0, s); seed_fp2(x.c1, s + 2); seed_fp2(x.c2, s + 4); }
__device__ __forceinline__ void seed_fp12(fp12& x, uint32_t s) { seed_fp6(x.c0, s); seed_fp6(x.c1, s + 6); }
__device__ __forceinline__ uint32_t fold(const fp12& f) {
    uint32_t s = 0;
    const uint32_t* w = (const uint32_t*)&f;
    for (int i = 0; i < (int)(sizeof(fp12) / 4); i++) s ^= w[i];
    return s;
}

template <int OP>
__global__ __launch_bounds__(64) void probe(uint32_t* out, int iters, uint32_t salt, int desync) {
    if (desync) {  // break the lockstep of the waves: each workgroup starts up to 1023 * desync cycles late
        const long long t0 = clock64(), d = (long long)((blockIdx.x * 7919u) & 1023u) * desync;
        while (clock64() - t0 < d) {}
    }
    fp12 f, g;
    seed_fp12(f, salt);
    seed_fp12(g, salt + 20);
    fp2 o0, o1, o4;
    seed_fp2(o0, salt + 40);
    seed_fp2(o1, salt + 42);
    seed_fp2(o4, salt + 44);
    g2_jac t;
    t.X = o0; t.Y = o1; t.Z = o4;
    fp xp, yp;
    seed_fp(xp, salt + 50);
    seed_fp(yp, salt + 51);
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (OP == 0) fp2_mul(f.c0.c0, f.c0.c0, g.c0.c0);          // inline, registers only
        if (OP == 1) fp2_sqr(f.c0.c0, f.c0.c0);
        if (OP == 2) fp2_mul_c<TC>(f.c0.c0, f.c0.c0, g.c0.c0);    // the same through a call
        if (OP == 3) fp6_mul_c<TC>(f.c0, f.c0, g.c0);
        if (OP == 4) fp12_sqr(f, f);
        if (OP == 5) fp12_mul(f, f, g);
        if (OP == 6) fp12_mul_by_014(f, o0, o1, o4);
        if (OP == 7) miller_dbl_line(o0, o1, o4, t, xp, yp);
        if (OP == 8) fp12_cyclo_sqr_n(f, f, 16);
        if (OP == 9) { miller_dbl_line(o0, o1, o4, t, xp, yp); fp12_sqr(f, f); fp12_mul_by_014(f, o0, o1, o4); }  // one Miller doubling step
    }
    out[blockIdx.x * 64 + threadIdx.x] = fold(f) ^ o0.c0.v[0] ^ o4.c1.v[3] ^ t.X.c0.v[1] ^ t.Y.c1.v[2] ^ t.Z.c0.v[5];
}

static int g_waves = 1024, g_desync = 0;
template <int OP>
static void report(const char* name, int iters) {
    uint32_t* d;
    hipMalloc(&d, (size_t)g_waves * 64 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(probe<OP>, dim3(g_waves), dim3(64), 0, 0, d, iters, 1u, g_desync);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<OP>, dim3(g_waves), dim3(64), 0, 0, d, iters, 2u, g_desync);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    printf("{\"waves\": %d, \"desync\": %d, \"op\": \"%s\", \"iters\": %d, \"ms\": %.3f, \"cycles_per_call\": %.0f}\n", g_waves, g_desync, name, iters, ms, ms * 1e-3 * 2.4e9 / iters);
}

int main(int argc, char** argv) {
    if (argc > 1) g_waves = atoi(argv[1]);
    if (argc > 2) g_desync = atoi(argv[2]);
    if (argc > 3) {  // short list
        report<3>("fp6_mul_c call", 600);
        report<4>("fp12_sqr call", 200);
        report<4>("fp12_sqr call", 800);
        report<9>("miller doubling step (3 calls)", 100);
        return 0;
    }
    report<0>("fp2_mul inline", 2000);
    report<1>("fp2_sqr inline", 2000);
    report<2>("fp2_mul_c call", 2000);
    report<3>("fp6_mul_c call", 600);
    report<4>("fp12_sqr call", 200);
    report<5>("fp12_mul call", 200);
    report<6>("fp12_mul_by_014 call", 200);
    report<7>("miller_dbl_line call", 200);
    report<8>("fp12_cyclo_sqr_n(16) call", 40);
    report<9>("miller doubling step (3 calls)", 100);
    return 0;
}
